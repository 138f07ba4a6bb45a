"""Sweep the XCD start stagger of the persistent GEMMs (pg_tune_gemm_stagger) on the model's four GEMM forms (the
LayerNorm-folded epilogues, 512 images = 295 424 rows) and on the whole 24-layer encoder.  Timing only.
   python tools/stagger_sweep.py [--fractions 0,0.5,0.875] [--layers 24]"""
import argparse, ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import _lib, hip_ops, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fractions", default="0,0.25,0.5,0.75,0.875,1.0,1.5")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--images", type=int, default=512)
    args = ap.parse_args()
    fr = [float(x) for x in args.fractions.split(",")]
    lib = _lib.load()
    _lib.require_gpu()
    dev = "cuda"
    M = args.images * 577
    g = torch.Generator(device=dev).manual_seed(1)
    dt = torch.float16
    A1 = torch.randn((M, 1024), generator=g, device=dev).to(dt)
    A4 = torch.randn((M, 4096), generator=g, device=dev).to(dt)
    X = torch.randn((M, 1024), generator=g, device=dev)
    rs = torch.stack([torch.ones(M, device=dev), torch.zeros(M, device=dev)], dim=1).contiguous()

    def mk(n, k):
        return (torch.randn((n, k), generator=g, device=dev) * 0.03).to(dt), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    Wq, bq, cq = mk(3072, 1024); Wo, bo, _ = mk(1024, 1024); W1, b1, c1 = mk(4096, 1024); W2, b2, _ = mk(1024, 4096)
    forms = {
        "qkv": (lambda: hip_ops.gemm16_ln(A1, Wq, bq, cq, rs, _lib.EPI_QKV_LN, qscale=0.18, qcols=1024), 2.0 * M * 3072 * 1024),
        "out": (lambda: hip_ops.gemm16_resid_stat(A1, Wo, bo, X), 2.0 * M * 1024 * 1024),
        "fc1": (lambda: hip_ops.gemm16_ln(A1, W1, b1, c1, rs, _lib.EPI_GELU_LN), 2.0 * M * 4096 * 1024),
        "fc2": (lambda: hip_ops.gemm16_resid_stat(A4, W2, b2, X), 2.0 * M * 1024 * 4096),
    }
    for name, (fn, fl) in forms.items():
        times = {f: [] for f in fr}
        for f in fr:
            lib.pg_tune_gemm_stagger(C.c_float(f))
            for _ in range(2):
                fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for f in fr:
                lib.pg_tune_gemm_stagger(C.c_float(f))
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    fn()
                b.record(); torch.cuda.synchronize()
                times[f].append(a.elapsed_time(b) / args.iters)
        line = "  ".join(f"f={f:g}: {sorted(times[f])[len(times[f]) // 2]:.3f} ms ({fl / (sorted(times[f])[len(times[f]) // 2] * 1e-3) / 1e12:.0f} TF)" for f in fr)
        print(f"{name}: {line}", flush=True)
    del A1, A4, X
    torch.cuda.empty_cache()
    if args.layers > 0:
        enc = hip_ops.VitEncoder(synthetic.make_vit_weights(seed=0, layers=args.layers), device=0)
        px = torch.randn((args.images, 3, 336, 336), generator=g, device=dev)
        for f in fr:
            lib.pg_tune_gemm_stagger(C.c_float(f)); enc.forward(px)
        torch.cuda.synchronize()
        times = {f: [] for f in fr}
        ref = None
        for _ in range(3):
            for f in fr:
                lib.pg_tune_gemm_stagger(C.c_float(f))
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); e = enc.forward(px); b.record(); torch.cuda.synchronize()
                times[f].append(a.elapsed_time(b))
                if ref is None:
                    ref = e.clone()
                assert torch.equal(e, ref), "stagger changed results"
        print("encoder (%d layers, %d images): " % (args.layers, args.images) +
              "  ".join(f"f={f:g}: {sorted(times[f])[1]:.2f} ms ({args.images / sorted(times[f])[1] * 1e3:.0f} img/s)" for f in fr), flush=True)


if __name__ == "__main__":
    main()
