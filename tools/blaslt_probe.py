"""Which kernels does stock PyTorch-ROCm (hipBLASLt / Tensile) pick for the four GEMM shapes of a 512-image chunk, and how
fast are they?  Run under rocprofv3 --kernel-trace --stats: the Tensile kernel name spells out macro tile, MFMA shape, depth,
LDS / prefetch options -- the calibration of our own GEMM (secondary baseline, never part of the product path).
   python tools/blaslt_probe.py [--iters 10]"""
import argparse
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = "cuda"
M = 512 * 577
for name, (k, n) in {"qkv": (1024, 3072), "out": (1024, 1024), "fc1": (1024, 4096), "fc2": (4096, 1024)}.items():
    a = torch.randn((M, k), device=dev, dtype=torch.float16)
    w = torch.randn((n, k), device=dev, dtype=torch.float16)
    b = torch.randn((n,), device=dev, dtype=torch.float16)
    for _ in range(3):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / args.iters
    print(f"{name}: M={M} N={n} K={k}  {t * 1e3:.3f} ms  {2.0 * M * n * k / t / 1e12:.1f} TF/s", flush=True)
    # no-bias variant too (pure GEMM)
    for _ in range(2):
        a @ w.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        a @ w.t()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / args.iters
    print(f"{name} (no bias): {t * 1e3:.3f} ms  {2.0 * M * n * k / t / 1e12:.1f} TF/s", flush=True)
    del a, w
