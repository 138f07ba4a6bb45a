"""Time the GPU CLIP preprocessing (pg_prep_forward): uint8 (N,H,W,3) -> fp16 (N,3,336,336).
   python tools/prep_bench.py --h 640 --w 640 --n 512"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import hip_ops

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=640); ap.add_argument("--w", type=int, default=640); ap.add_argument("--n", type=int, default=512)
a = ap.parse_args()
img = torch.randint(0, 256, (a.n, a.h, a.w, 3), dtype=torch.uint8, device="cuda")
prep = hip_ops.Preprocessor(a.h, a.w)
for dt in (torch.float16, torch.float32):
    for _ in range(3):
        prep(img, dt)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        prep(img, dt)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    inb, outb = a.n * a.h * a.w * 3, a.n * 3 * 336 * 336 * (2 if dt == torch.float16 else 4)
    print(f"PREP {a.h}x{a.w} n={a.n} -> {dt}: {ms:.3f} ms, {a.n / ms * 1e3:.0f} images/s, {(inb + outb) / ms / 1e6:.0f} GB/s algorithmic "
          f"(rows used {prep.nrows} of {a.h})", flush=True)
