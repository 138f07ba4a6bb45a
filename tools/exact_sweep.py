"""Exact encoder (pg_vit_forward_precise) time per image against the batch size: where the 256-row tiles of the N = 1024 GEMMs fill
whole rounds of the 256 CUs (64 row panels = 16 384 rows = 28.4 images per round) and where they do not.
   python tools/exact_sweep.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pigeon_amd import hip_ops, synthetic

ns = [int(a) for a in sys.argv[1:]] or [20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 72, 84, 88, 112, 116, 128]
sd = synthetic.make_vit_weights(seed=0, layers=24)
enc = hip_ops.VitEncoder(sd, precise=True)
px_all = torch.randn((max(ns), 3, 336, 336), device="cuda")
for n in ns:
    px = px_all[:n]
    enc.forward_precise(px)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        enc.forward_precise(px)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    t = sorted(ts)[1]
    rows = n * 577
    panels = (rows + 255) // 256
    print(f"{n:4d} images ({n / 4:5.1f} panoramas) {rows:6d} rows {panels:4d} row panels = {panels * 4 / 256:5.2f} rounds at N = 1024: "
          f"{t * 1e3:7.2f} ms = {t / n * 1e3:.4f} ms/image", flush=True)
