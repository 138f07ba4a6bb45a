"""Latency of the encoder at small batches (the serving path: one panorama = 4 images), default GEMM routing vs everything
through the small-tile kernel (PIGEON_GEMM_VARIANT=70).   python tools/latency_probe.py [n_images ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pigeon_amd import hip_ops, synthetic

ns = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32, 64]
sd = synthetic.make_vit_weights(seed=0, layers=24)
enc = hip_ops.VitEncoder(sd)
for n in ns:
    px = torch.randn((n, 3, 336, 336), device="cuda")
    for _ in range(3):
        e = enc(px)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        enc(px)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / 10
    print(f"variant {os.environ.get('PIGEON_GEMM_VARIANT', 'default')}: {n:3d} images {t * 1e3:7.2f} ms  ({n / t:7.0f} images/s)  sum {float(e.sum()):.6f}", flush=True)
