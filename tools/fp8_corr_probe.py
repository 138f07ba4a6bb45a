"""Round 6 probe (numerics only, no kernel): what the exact tier's residual would be if its two CORRECTION products ran on the fp8
matrix pipe (block-scaled e4m3, `v_mfma_scale_f32_*_f8f6f4`: twice the fp16 rate on gfx950) instead of fp16.

The exact tier computes x.W ~= hi.Wh + lo.Wh + hi.Wl (pigeon_amd/csrc/precise.hip).  The last two terms are 2^-11 of the first, so they
need 2^-11 less relative accuracy.  This script runs transformers.CLIPVisionModel (the module the reference calls,
models/clip_embedder.py:63-65) in float64 with every nn.Linear replaced by an emulation of one operand scheme (all sums in float64:
what is measured is the OPERAND rounding, nothing else) and compares token-mean embeddings with the plain float64 model:

  x3      hi.Wh + lo.Wh + hi.Wl, all operands fp16 (the exact tier as built; its floor)
  fp8     hi.Wh (fp16) + q8(lo).q8(Wh) + q8(hi).q8(Wl)        -- 1 + 1/2 + 1/2 = 2 units of MFMA time instead of 3
  lo8     hi.Wh + q8(lo).q8(Wh) + hi.Wl (fp16)                 -- 2.5 units; no systematic part expected (lo is rounding noise)
  wl8     hi.Wh + lo.Wh (fp16) + q8(hi).q8(Wl)                 -- 2.5 units; the weight-side error is the same for every image
  fast    hi.Wh only (the 16-bit path's GEMM operands; sanity: the known 2.6e-4 / 4.7e-5)

q8 = OCP e4m3 with one power-of-two scale per 32 consecutive k (the MX block format of the scaled MFMA).
Per scheme: total relative error, systematic part |beta| (mean error vector), residual after removing beta (fitted on the even images,
measured on the odd ones -- what pigeon_amd/certainty.py would calibrate away).
usage: python tools/fp8_corr_probe.py [images] [default|spread]"""
import contextlib, io, os, sys, time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import synthetic  # noqa: E402


def q_mx(t):
    """float32 [..., K] -> e4m3 values with a shared power-of-two scale per 32 consecutive k (largest magnitude lands in [128, 256))."""
    s = t.shape
    b = t.reshape(-1, 32)
    amax = b.abs().amax(dim=1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 7.0)
    return ((b / scale).to(torch.float8_e4m3fn).float() * scale).reshape(s)


class SplitLinear(torch.nn.Module):
    def __init__(self, lin, mode):
        super().__init__()
        W = lin.weight.detach().float()
        self.mode = mode
        self.bias = lin.bias.detach().double()
        Wh = W.half().float()
        Wl = W - Wh
        self.Wh = Wh.double().t().contiguous()
        self.Wl16 = ((Wl * 256.0).half().float() / 256.0).double().t().contiguous()
        self.Wh8 = q_mx(Wh).double().t().contiguous()
        self.Wl8 = q_mx(Wl).double().t().contiguous()

    def forward(self, x):
        x32 = x.float()
        hi = x32.half().float()
        lo = (x32 - hi)
        lo16 = lo.half().float()
        his = ((hi * (1 / 256.0)).half().float() * 256.0)              # hi * 2^-8 as fp16 (subnormals of tiny values), scaled back
        m = self.mode
        out = hi.double() @ self.Wh
        if m == "fast":
            return out + self.bias
        out = out + ((q_mx(lo).double() @ self.Wh8) if m in ("fp8", "lo8") else (lo16.double() @ self.Wh))
        out = out + ((q_mx(hi).double() @ self.Wl8) if m in ("fp8", "wl8") else (his.double() @ self.Wl16))
        return out + self.bias


def patch(model, mode):
    for name, mod in list(model.named_modules()):
        for cname, child in list(mod.named_children()):
            if isinstance(child, torch.nn.Linear):
                setattr(mod, cname, SplitLinear(child, mode))
    return model


def stats(a, ref, what):
    rel = (a - ref) / ref.norm(dim=1, keepdim=True)
    total = float(rel.norm(dim=1).pow(2).mean().sqrt())
    beta = rel[0::2].mean(dim=0)
    resid = float((rel[1::2] - beta).norm(dim=1).pow(2).mean().sqrt())
    worst = float((rel[1::2] - beta).norm(dim=1).max())
    print(f"  {what:5s}: total {total:.3e}   systematic |beta| {float(rel.mean(dim=0).norm()):.3e}   residual after beta (out of sample) "
          f"{resid:.3e} (worst image {worst:.3e})   worst image total {float(rel.norm(dim=1).max()):.3e}", flush=True)


def main():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    tower = sys.argv[2] if len(sys.argv) > 2 else "default"
    dev = "cuda"
    sd = synthetic.make_vit_weights_spread(seed=31, layers=24) if tower == "spread" else synthetic.make_vit_weights(seed=0, layers=24)
    cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                           patch_size=14, projection_dim=768)
    g = torch.Generator(device=dev).manual_seed(4321)
    px = torch.randn((n, 3, 336, 336), generator=g, device=dev)

    def run(mode):
        with contextlib.redirect_stdout(io.StringIO()):
            hf = CLIPVisionModel._from_config(cfg, attn_implementation="eager")
        hf.load_state_dict(sd, strict=True)
        hf = hf.to(dev).double().eval()
        if mode != "f64":
            patch(hf, mode)
        out = []
        with torch.no_grad():
            for j in range(0, n, 8):
                out.append(hf(pixel_values=px[j:j + 8].double()).last_hidden_state.mean(dim=1))
        return torch.cat(out)

    t0 = time.perf_counter()
    ref = run("f64")
    print(f"{tower} tower, {n} images (per-image embeddings, float64 reference {time.perf_counter() - t0:.0f} s):", flush=True)
    for mode in ("fast", "x3", "fp8", "lo8", "wl8"):
        stats(run(mode), ref, mode)
    print(f"  ({time.perf_counter() - t0:.0f} s)")


if __name__ == "__main__":
    main()
