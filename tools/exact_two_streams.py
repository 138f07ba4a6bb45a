"""Do two independent exact passes overlap when issued on two HIP streams?  (round 5 experiment)
   8 images as ONE pass  vs  2 x 4 images back to back on one stream  vs  2 x 4 images on two streams (two encoder handles)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pigeon_amd import hip_ops, synthetic

sd = synthetic.make_vit_weights(seed=0, layers=24)
e1 = hip_ops.VitEncoder(sd, precise=True)
e2 = hip_ops.VitEncoder(sd, precise=True)
px = torch.randn((16, 3, 336, 336), device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, iters=4):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def two_streams(n):
    with torch.cuda.stream(s1):
        e1.forward_precise(px[:n])
    with torch.cuda.stream(s2):
        e2.forward_precise(px[n:2 * n])


for n in (4, 8):
    one = timed(lambda: e1.forward_precise(px[:2 * n]))
    seq = timed(lambda: (e1.forward_precise(px[:n]), e1.forward_precise(px[n:2 * n])))
    par = timed(lambda: two_streams(n))
    print(f"{2 * n} images: one pass {one:.1f} ms; two passes of {n} back to back {seq:.1f} ms; two passes of {n} on two streams {par:.1f} ms")
