"""Run one GEMM shape repeatedly (for rocprofv3 / quick A/B timing).
   python tools/gemm_prof.py --shape qkv --variant 1 --iters 10 [--dtype f16] [--time]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import _lib, hip_ops

SHAPES = {"qkv": (3072, 1024, _lib.EPI_QKV), "out": (1024, 1024, _lib.EPI_RESID),
          "fc1": (4096, 1024, _lib.EPI_GELU), "fc2": (1024, 4096, _lib.EPI_RESID)}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="qkv")
    ap.add_argument("--variants", default="1")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--real", action="store_true", help="LayerNorm-like operand statistics instead of N(0,1)")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    dev = "cuda"
    M = args.images * 577
    for shape in args.shape.split(","):
        N, K, epi = SHAPES[shape]
        g = torch.Generator().manual_seed(1)
        A = (torch.randn((M, K), generator=g)).to(dt).to(dev)
        W = (torch.randn((N, K), generator=g) * 0.03).to(dt).to(dev)
        bias = torch.zeros(N, device=dev)
        out = torch.zeros((M, N), dtype=dt if epi in (_lib.EPI_QKV, _lib.EPI_GELU) else torch.float32, device=dev)
        variants = [int(x) for x in args.variants.split(",")]
        times = {v: [] for v in variants}
        for v in variants:                      # warm up every variant (and the clocks) first
            for _ in range(3):
                hip_ops.gemm16(A, W, bias, out, epi, qscale=0.18, qcols=1024, variant=v)
        torch.cuda.synchronize()
        for rnd in range(args.rounds):          # interleaved rounds: order effects cancel
            for v in variants:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    hip_ops.gemm16(A, W, bias, out, epi, qscale=0.18, qcols=1024, variant=v)
                b.record(); torch.cuda.synchronize()
                times[v].append(a.elapsed_time(b) / args.iters)
        for v in variants:
            ts = sorted(times[v]); med = ts[len(ts)//2]; mn = ts[0]
            fl = 2.0*M*N*K
            print(f"{shape} variant {v:2d} {args.dtype}: median {med:.3f} ms {fl/(med*1e-3)/1e12:7.1f} TF/s   best {mn:.3f} ms {fl/(mn*1e-3)/1e12:7.1f} TF/s", flush=True)

if __name__ == "__main__":
    main()
