"""Round 6: where does the 128 x 128 small-batch kernel (gemm_mid.hip, variant 71) beat the persistent kernels (variant 56 = the product
default: 384 x 256 where it exists, 256 x 256 elsewhere)?  The model's four GEMM shapes with their fused epilogues, M = 577 x n images;
microseconds per launch (HIP events over 50 launches), outputs compared bit for bit; the letter is what pg_gemm_launch's cost model
picks when left alone (p = the variant's persistent kernel, m = the 128 x 128 kernel; with --three a third arm and letter: q = the
256 x 256 persistent kernel where variant 56 means the 384 x 256 one).
   python tools/gemm_mid_sweep.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigeon_amd import _lib, hip_ops as ops

L = _lib
dev = "cuda"
g = torch.Generator().manual_seed(0)


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


shapes = [("qkv  1024->3072 LN fold", 3072, 1024, "ln", L.EPI_QKV_LN), ("out  1024->1024 resid+stat", 1024, 1024, "rs", None),
          ("fc1  1024->4096 LN fold + GELU", 4096, 1024, "ln", L.EPI_GELU_LN), ("fc2  4096->1024 resid+stat", 1024, 4096, "rs", None)]
THREE = "--three" in sys.argv      # third arm: the 256 x 256 persistent kernel (variant 36) where variant 56 means the 384 x 256 one
print(f"{'shape':34s} " + " ".join(f"n={n:<3d} pers/{'p256/' if THREE else ''}mid us" for n in (1, 2, 4, 8, 12, 16, 24, 32, 64)))
for name, N, K, kind, epi in shapes:
    W = (torch.randn((N, K), generator=g) * 0.05).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    cs = torch.randn(N, generator=g).to(dev)
    cells = []
    for n in (1, 2, 4, 8, 12, 16, 24, 32, 64):
        M = 577 * n
        A = torch.randn((M, K), generator=g).half().to(dev)
        rs = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)], dim=1).contiguous().to(dev)
        X = torch.randn((M + 400, N), generator=g).to(dev)
        t, outs = {}, {}
        X0 = X.clone()
        for v in ((56, 36, 71) if THREE else (56, 71)):
            ops.tune_gemm_mid(False)                                   # variant 56 = the persistent kernels themselves, no small-batch routing
            if kind == "ln":
                fn = lambda: ops.gemm16_ln(A, W, bias, cs, rs, epi, qscale=0.25, qcols=1024, variant=v)      # noqa: E731
                outs[v] = [fn()]
            else:
                Xc = X0.clone()
                x16, part = ops.gemm16_resid_stat(A, W, bias, Xc[:M], variant=v)
                outs[v] = [Xc[:M].clone(), x16, part]
                Xt = torch.zeros_like(X0)                              # timing target: zeros stay finite however often the residual is added
                fn = lambda: ops.gemm16_resid_stat(A, W, bias * 0, Xt[:M], variant=v)                        # noqa: E731
            t[v] = timeit(fn)
        ops.tune_gemm_mid(True)
        tp = timeit((lambda: ops.gemm16_ln(A, W, bias, cs, rs, epi, qscale=0.25, qcols=1024, variant=56)) if kind == "ln" else
                    (lambda: ops.gemm16_resid_stat(A, W, bias * 0, Xt[:M], variant=56)))
        same = all(torch.equal(a, b) for a, b in zip(outs[56], outs[71]))
        kind = C.c_int(-1)                                             # what pg_gemm_launch's routing takes for this shape (pg_gemm_route)
        L.load().pg_gemm_route(56, epi if epi is not None else L.EPI_RESID_STAT, M, N, K, C.byref(kind))
        own_is_256 = epi is None and K < 2048                          # out-projection: the variant's own kernel is the 256 x 256 one
        pick = "m" if kind.value == 2 else ("p" if (kind.value == 0 or own_is_256) else "q")
        if THREE:
            same = same and all(torch.equal(a, b) for a, b in zip(outs[56], outs[36]))
            cells.append(f"{t[56]:6.1f}/{t[36]:6.1f}/{t[71]:6.1f}{pick}{'' if same else ' DIFF'}")
        else:
            cells.append(f"{t[56]:6.1f}/{t[71]:6.1f}{pick}{'' if same else ' DIFF'}")
    print(f"{name:34s} " + "  ".join(cells))
