"""Which kernels run at the 1400 W package power cap, and at what clock?

Loops ONE kernel for ~1.5 s at a time while a thread samples `rocm-smi --showpower --showclocks`, and prints the
median power / sclk of the samples taken inside the loop next to the kernel's time per launch.  Kernels: the four
GEMM shapes of a layer (product variant 36), the timing-only GEMM ablations (40 no DMA, 41 every DMA hits operand
panel 0 = L2 resident, 43 MFMA + barriers only), attention, LayerNorm.

    python tools/power_probe.py [--images 512] [--seconds 1.5]
"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import _lib as L, hip_ops  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            t = time.time()
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                continue
            p = re.search(r"Package Power \(W\): ([0-9.]+)", o)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
            if p and c:
                self.samples.append((0.5 * (t + time.time()), float(p.group(1)), int(c.group(1))))


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def probe(name, fn, seconds, sampler, flops=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    t1 = time.time()
    ms = a.elapsed_time(b) / n
    # the first 0.4 s are the firmware ramping down from the idle clock: keep the samples after that
    s = [x for x in sampler.samples if t0 + 0.4 <= x[0] <= t1]
    tf = f"{flops / (ms * 1e-3) / 1e12:7.1f} TF/s" if flops else " " * 12
    print(f"{name:34s} {ms:7.3f} ms {tf}   power {med([x[1] for x in s]):6.0f} W   sclk {med([x[2] for x in s]):5.0f} MHz   ({len(s)} samples)",
          flush=True)
    time.sleep(0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--seconds", type=float, default=1.5)
    args = ap.parse_args()
    dev, dt = "cuda", torch.float16
    M = args.images * 577
    sm = Sampler(); sm.start()
    time.sleep(1.0)
    idle = [x for x in sm.samples]
    print(f"idle: power {med([x[1] for x in idle]):.0f} W  sclk {med([x[2] for x in idle]):.0f} MHz", flush=True)
    g = torch.Generator().manual_seed(1)
    shapes = {"qkv": (3072, 1024, L.EPI_QKV), "out": (1024, 1024, L.EPI_RESID), "fc1": (4096, 1024, L.EPI_GELU),
              "fc2": (1024, 4096, L.EPI_RESID)}
    for name, (N, K, epi) in shapes.items():
        A = torch.randn((M, K), generator=g).to(dt).to(dev)
        W = (torch.randn((N, K), generator=g) * 0.03).to(dt).to(dev)
        bias = torch.zeros(N, device=dev)
        out = torch.zeros((M, N), dtype=dt if epi in (L.EPI_QKV, L.EPI_GELU) else torch.float32, device=dev)
        variants = (36, 41, 40, 43) if name in ("fc1", "fc2") else (36,)
        for v in variants:
            probe(f"gemm {name} variant {v}", lambda: hip_ops.gemm16(A, W, bias, out, epi, qscale=0.18, qcols=1024, variant=v),
                  args.seconds, sm, 2.0 * M * N * K)
        del A, W, out
    qkv = torch.randn((M, 3072), generator=g).to(dt).to(dev)
    qkv[:, :1024] *= 0.18
    probe("attention", lambda: hip_ops.attention(qkv, args.images), args.seconds, sm, 4.0 * 577 * 577 * 64 * 16 * args.images)
    del qkv
    x = torch.randn((M, 1024), generator=g).to(dev)
    gamma, beta = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    xn = torch.empty((M, 1024), dtype=dt, device=dev)
    probe("layernorm", lambda: hip_ops.layernorm(x, gamma, beta, 1e-5, out=xn), args.seconds, sm)
    sm.stop = True


if __name__ == "__main__":
    main()
