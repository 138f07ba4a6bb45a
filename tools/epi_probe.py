"""What does each fused epilogue of the persistent 256x256 GEMM cost?  Same mainloop (variant 36), the out-projection and fc2
shapes (295 424 rows), epilogues: 16-bit store (+bias), fp32 store, fp32 residual read-modify-write, the same + 16-bit copy +
row statistics (LayerNorm fold).  Timing only.   python tools/epi_probe.py [variants, e.g. 36,64]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import _lib, hip_ops

_lib.require_gpu()
VARIANTS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "36").split(",")]
dev, dt = "cuda", torch.float16
M = 512 * 577
g = torch.Generator(device=dev).manual_seed(1)
for name, K in (("out  (N=1024, K=1024)", 1024), ("fc2  (N=1024, K=4096)", 4096)):
    N = 1024
    A = torch.randn((M, K), generator=g, device=dev).to(dt)
    W = (torch.randn((N, K), generator=g, device=dev) * 0.03).to(dt)
    bias = torch.zeros(N, device=dev)
    o16 = torch.zeros((M, N), dtype=dt, device=dev)
    o32 = torch.zeros((M, N), dtype=torch.float32, device=dev)
    forms = {}
    for v in VARIANTS:
        forms.update({
            f"v{v} 16-bit store": lambda v=v: hip_ops.gemm16(A, W, bias, o16, _lib.EPI_QKV, qscale=1.0, qcols=0, variant=v),
            f"v{v} fp32 store": lambda v=v: hip_ops.gemm16(A, W, bias, o32, _lib.EPI_F32, variant=v),
            f"v{v} fp32 residual RMW": lambda v=v: hip_ops.gemm16(A, W, bias, o32, _lib.EPI_RESID, variant=v),
            f"v{v} residual RMW + 16-bit copy + row stats": lambda v=v: hip_ops.gemm16_resid_stat(A, W, bias, o32, variant=v),
        })
    res = {}
    for k, fn in forms.items():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for rnd in range(5):
        for k, fn in forms.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(6):
                fn()
            b.record(); torch.cuda.synchronize()
            res.setdefault(k, []).append(a.elapsed_time(b) / 6)
    fl = 2.0 * M * N * K
    tiles_per_cu = (M + 255) // 256 * 4 / 256
    for k, v in res.items():
        med = sorted(v)[len(v) // 2]
        print(f"{name}  {k:42s} {med:.3f} ms  {fl / (med * 1e-3) / 1e12:6.0f} TF/s   {med * 1e3 / tiles_per_cu:5.1f} us per tile", flush=True)
    del A, W, o16, o32
