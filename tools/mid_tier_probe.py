"""Round 6: what an ACTIVATION-SPLIT-ONLY tier (pg_tune_exact_products(2): hi.Wh + lo.Wh, the weights at their fp16 value) would be worth
between the fast path and the exact tier.  Per tower: the relative error of the fast and of the 2-product encoder against the exact
(3-product) one on panel-mean embeddings of N panoramas -- total RMS, systematic part (mean vector), residual after it (out of sample:
fitted on the even samples, measured on the odd ones, as pigeon_amd/certainty.py does) -- and the time per image of both at 44 images.
usage: python tools/mid_tier_probe.py [panoramas] [default|spread]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pigeon_amd import _lib, hip_ops, synthetic  # noqa: E402


def stats(a, ref, what):
    rel = (a - ref) / ref.norm(dim=1, keepdim=True)
    total = float(rel.norm(dim=1).pow(2).mean().sqrt())
    beta = rel[0::2].mean(dim=0)
    resid = float((rel[1::2] - beta).norm(dim=1).pow(2).mean().sqrt())
    scale = ((rel @ beta) / beta.norm().pow(2))
    print(f"  {what}: total {total:.3e}, systematic |beta| {float(rel.mean(dim=0).norm()):.3e}, residual (out of sample) {resid:.3e}; "
          f"per-sample scale of beta: mean {float(scale.mean()):.3f} std {float(scale.std()):.3f}")
    return resid


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    tower = sys.argv[2] if len(sys.argv) > 2 else "default"
    dev = "cuda"
    sd = synthetic.make_vit_weights_spread(seed=31, layers=24) if tower == "spread" else synthetic.make_vit_weights(seed=0, layers=24)
    enc = hip_ops.VitEncoder(sd, layers=24, precise=True)
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(4321)
    px = torch.randn((n * 4, 3, 336, 336), generator=g, device=dev)
    pm = lambda e: e.reshape(n, 4, 1024).mean(dim=1).double()                       # noqa: E731
    fast = pm(enc(px))
    lib.pg_tune_exact_products(3)
    exact = pm(enc.forward_precise(px))
    lib.pg_tune_exact_products(2)
    mid = pm(enc.forward_precise(px))
    print(f"{tower} tower, {n} panoramas:")
    r_fast = stats(fast, exact, "fast (16-bit) vs exact")
    r_mid = stats(mid, exact, "2-product     vs exact")
    print(f"  residual ratio fast / 2-product: {r_fast / max(r_mid, 1e-30):.1f}")
    for prod in (3, 2):
        lib.pg_tune_exact_products(prod)
        for m in (44, 64, 128):
            x = px[:m]
            enc.forward_precise(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                enc.forward_precise(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print(f"  products {prod}, {m} images: {dt * 1e3:.1f} ms = {dt * 1e3 / m:.3f} ms/image")
    lib.pg_tune_exact_products(3)


if __name__ == "__main__":
    main()
