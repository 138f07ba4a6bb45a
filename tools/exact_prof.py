"""Time the encoder's exact mode (pg_vit_forward_precise) on N images; run under rocprofv3 by tools/prof_exact.sh.
   python tools/exact_prof.py [n_images] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pigeon_amd import hip_ops, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 52
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sd = synthetic.make_vit_weights(seed=0, layers=24)
enc = hip_ops.VitEncoder(sd, precise=True)
px = torch.randn((n, 3, 336, 336), device="cuda")
enc.forward_precise(px)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(iters):
    enc.forward_precise(px)
torch.cuda.synchronize()
t = (time.perf_counter() - t) / iters
fast = enc(px); torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(iters):
    enc(px)
torch.cuda.synchronize()
t2 = (time.perf_counter() - t2) / iters
print(f"exact mode: {n} images in {t * 1e3:.1f} ms = {t / n * 1e3:.3f} ms/image ({n / t:.0f} images/s, {n * 381.918e9 / t / 1e12:.0f} TFLOP/s fp32-equivalent); "
      f"fast path on the same batch: {t2 * 1e3:.1f} ms = {t2 / n * 1e3:.3f} ms/image; ratio {t / t2:.2f}")
