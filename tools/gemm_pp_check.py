"""Bit-compare GEMM variants against variant 8 (the one-tile-per-block kernel) and time them on the model's shapes.

   python tools/gemm_pp_check.py --variants 56,36,30 [--time] [--images 512]

Every variant runs in its own subprocess under a timeout (a wrong barrier count would hang the GPU), the parent only
collects the printed lines.  Variants built on the same MFMA instruction accumulate every output element in the same order
and must be BIT-identical; since round 2 the persistent kernels (33, 36, 56, 64) use v_mfma_f32_16x16x32 while variant 8 keeps
32x32x16, so against variant 8 they are held to fp32 rounding (TOL_VARIANTS).
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


TOL_VARIANTS = (64, 56, 36, 33, 70, 72)       # v_mfma_f32_16x16x32: same math, different association inside the instruction


def child(args):
    import torch
    from pigeon_amd import _lib, hip_ops
    L = _lib
    dev = "cuda"
    v = args.child
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    ok = True
    if not args.skip_check:
        # (M, N, K): ragged M tail with one tile per block; > 256 tiles so persistent blocks walk several tiles;
        # a long-K shape; the patch-embedding shape (K = 640, 10 K tiles)
        for (M, N, K) in [(1191, 512, 384), (70 * 256 + 19, 1024, 256), (3 * 577, 1024, 4096), (2 * 576, 1024, 640)]:
            g = torch.Generator().manual_seed(M + N + K)
            A = torch.randn((M, K), generator=g).to(dt).to(dev)
            W = (torch.randn((N, K), generator=g) * 0.05).to(dt).to(dev)
            bias = torch.randn(N, generator=g).to(dev)
            X0 = torch.randn((M, N), generator=g).to(dev)
            for epi, name in [(L.EPI_F32, "f32"), (L.EPI_QKV, "qkv"), (L.EPI_GELU, "gelu"), (L.EPI_RESID, "resid")]:
                outs = []
                for var in (8, v):
                    if epi in (L.EPI_QKV, L.EPI_GELU):
                        o = torch.full((M + 3, N), 7.0, dtype=dt, device=dev)      # 3 guard rows past M
                    elif epi == L.EPI_RESID:
                        o = torch.cat([X0, torch.full((3, N), 7.0, device=dev)]).contiguous()
                    else:
                        o = torch.full((M + 3, N), 7.0, device=dev)
                    hip_ops.gemm16(A, W, bias, o, epi, qscale=0.25, qcols=256, variant=var, M=M)
                    torch.cuda.synchronize()
                    outs.append(o)
                same = torch.equal(outs[0], outs[1])
                if not same and v in TOL_VARIANTS:       # a different MFMA shape sums the k products in another order
                    a_, b_ = outs[0][:M].float(), outs[1][:M].float()
                    same = bool(((a_ - b_).abs() <= 1e-5 * a_.abs().max() + 2e-3 * a_.abs()).all())
                guard = bool((outs[1][M:].float() == 7.0).all())
                if not (same and guard):
                    ok = False
                    d = (outs[0].float() - outs[1].float()).abs()
                    bad = (d > 0).nonzero()
                    print(f"CHECK v{v} {M}x{N}x{K} {name}: MISMATCH same={same} guard={guard} n_bad={bad.shape[0]} "
                          f"first={bad[:3].tolist()} maxdiff={d.max().item():.3e}", flush=True)
            # LayerNorm-fold epilogues (persistent kernels only): against variant 36
            if N % 256 == 0 and K % 128 == 0 and v != 36:
                rs = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)], dim=1).contiguous().to(dev)
                cs = torch.randn(N, generator=g).to(dev)
                for epi, name in [(L.EPI_QKV_LN, "qkv_ln"), (L.EPI_GELU_LN, "gelu_ln")]:
                    o = [hip_ops.gemm16_ln(A, W, bias, cs, rs, epi, qscale=0.25, qcols=256, variant=var) for var in (36, v)]
                    torch.cuda.synchronize()
                    eq = torch.equal(o[0], o[1])
                    if not eq and v in TOL_VARIANTS:
                        eq = bool(((o[0].float() - o[1].float()).abs() <= 1e-5 * o[0].float().abs().max() + 2e-3 * o[0].float().abs()).all())
                    if not eq:
                        ok = False
                        print(f"CHECK v{v} {M}x{N}x{K} {name}: MISMATCH maxdiff={(o[0].float() - o[1].float()).abs().max().item():.3e}", flush=True)
                outs = []
                for var in (36, v):
                    X = X0.clone()
                    x16, part = hip_ops.gemm16_resid_stat(A, W, bias, X, variant=var)
                    torch.cuda.synchronize()
                    outs.append((X, x16, part))
                for nm, a_, b_ in zip(("X", "x16", "statpart"), outs[0], outs[1]):
                    eq = torch.equal(a_, b_)
                    if not eq and v in TOL_VARIANTS:
                        eq = bool(((a_.float() - b_.float()).abs() <= 1e-5 * a_.float().abs().max() + 2e-3 * a_.float().abs()).all())
                    if not eq:
                        ok = False
                        print(f"CHECK v{v} {M}x{N}x{K} resid_stat {nm}: MISMATCH maxdiff={(a_.float() - b_.float()).abs().max().item():.3e}", flush=True)
            print(f"CHECK v{v} {M}x{N}x{K}: done ok={ok}", flush=True)
    if args.time:
        shapes = {"qkv": (3072, 1024, L.EPI_QKV), "out": (1024, 1024, L.EPI_RESID),
                  "fc1": (4096, 1024, L.EPI_GELU), "fc2": (1024, 4096, L.EPI_RESID)}
        M = args.images * 577
        for name, (N, K, epi) in shapes.items():
            g = torch.Generator().manual_seed(1)
            pa, pw, po = args.pad_a, args.pad_w, args.pad_o          # leading-dimension padding in elements
            A = torch.empty((M, K + pa), dtype=dt, device=dev)[:, :K]
            A.copy_(torch.randn((M, K), generator=g).to(dt))
            W = torch.empty((N, K + pw), dtype=dt, device=dev)[:, :K]
            W.copy_((torch.randn((N, K), generator=g) * 0.03).to(dt))
            bias = torch.zeros(N, device=dev)
            out = torch.zeros((M, N + po), dtype=dt if epi in (L.EPI_QKV, L.EPI_GELU) else torch.float32, device=dev)[:, :N]
            for _ in range(3):
                hip_ops.gemm16(A, W, bias, out, epi, qscale=0.18, qcols=1024, variant=v)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.rounds):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    hip_ops.gemm16(A, W, bias, out, epi, qscale=0.18, qcols=1024, variant=v)
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / args.iters)
            ts.sort()
            fl = 2.0 * M * N * K
            print(f"TIME v{v} {name} n={args.images} pad={args.pad_a}/{args.pad_w}/{args.pad_o}: median {ts[len(ts)//2]:.3f} ms {fl/(ts[len(ts)//2]*1e-3)/1e12:7.1f} TF/s"
                  f"   best {ts[0]:.3f} ms {fl/(ts[0]*1e-3)/1e12:7.1f} TF/s", flush=True)
            del A, W, out
    print(f"RESULT v{v} ok={ok}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="56,36")
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--timeout", type=int, default=150)
    ap.add_argument("--pad-a", type=int, default=0)
    ap.add_argument("--pad-w", type=int, default=0)
    ap.add_argument("--pad-o", type=int, default=0)
    args = ap.parse_args()
    if args.child >= 0:
        child(args)
        return
    for v in [int(x) for x in args.variants.split(",")]:
        cmd = ["timeout", "-k", "10", str(args.timeout), sys.executable, os.path.abspath(__file__), "--child", str(v),
               "--images", str(args.images), "--iters", str(args.iters), "--rounds", str(args.rounds), "--dtype", args.dtype,
               "--pad-a", str(args.pad_a), "--pad-w", str(args.pad_w), "--pad-o", str(args.pad_o)]
        if args.time:
            cmd.append("--time")
        if args.skip_check:
            cmd.append("--skip-check")
        r = subprocess.run(cmd, capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode != 0:
            print(f"RESULT v{v} FAILED rc={r.returncode} stderr tail: {r.stderr[-600:]}", flush=True)
            if r.returncode in (124, 137):
                print("a variant hung: stopping here (the GPU may need a reset)", flush=True)
                break


if __name__ == "__main__":
    main()
