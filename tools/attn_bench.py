"""Time the attention kernel (variant from env PIGEON_ATTN_VARIANT) and check it against an fp32 reference.
   PIGEON_ATTN_VARIANT=2 python tools/attn_bench.py --images 512"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import hip_ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    var = os.environ.get("PIGEON_ATTN_VARIANT", "default")
    g = torch.Generator().manual_seed(21)
    n = 2
    qkv = torch.randn((n * 577, 3072), generator=g)
    qkv[:, :1024] *= 0.125 * 1.4426950408889634 * 2.0
    qkv[300, 1024:2048] *= 30
    q16 = qkv.to(torch.float16)
    out = hip_ops.attention(q16.cuda(), n).float().cpu()
    x = q16.float().view(n, 577, 3, 16, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    p = torch.softmax((q @ k.transpose(-1, -2)) * 0.6931471805599453, dim=-1)       # Q carries log2(e): 2^s == e^(s ln2)
    ref = (p @ v).transpose(1, 2).reshape(n * 577, 1024)
    err = ((out - ref).norm() / ref.norm()).item()
    last = torch.tensor([576, 2 * 577 - 1])
    err_last = ((out[last] - ref[last]).norm() / ref[last].norm()).item()
    print(f"ATTN variant {var}: rel err vs fp32 {err:.2e} (token 576's rows: {err_last:.2e})", flush=True)
    n = args.images
    big = torch.randn((n * 577, 3072), generator=g).to(torch.float16).cuda()
    big[:, :1024] *= 0.18
    for _ in range(3):
        hip_ops.attention(big, n)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            hip_ops.attention(big, n)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / args.iters)
    ts.sort()
    fl = 4.0 * 577 * 577 * 64 * 16 * n
    print(f"ATTN variant {var} n={n}: median {ts[len(ts)//2]:.3f} ms {fl/(ts[len(ts)//2]*1e-3)/1e12:6.1f} TF/s  best {ts[0]:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
