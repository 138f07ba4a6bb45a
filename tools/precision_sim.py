"""CPU emulation of the reduced-precision data path (which tensors are rounded to 16 bit and how) to predict the
embedding error of a precision scheme before building it.  Usage: python tools/precision_sim.py"""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import synthetic
from oracle import pigeon_oracle as orc


def rnd(x, dt):
    return x if dt is None else x.to(dt).float()


@torch.no_grad()
def vit_emul(sd, px, act=torch.bfloat16, wdt=torch.bfloat16, patch=torch.bfloat16, pdt=torch.bfloat16):
    N = px.shape[0]
    W = {k: v for k, v in sd.items()}
    def lin(x, w, b, a=act, wd=wdt):
        return F.linear(rnd(x, a), rnd(w, wd), b)
    x = F.unfold(px, 14, stride=14).transpose(1, 2)              # (N,576,588)
    pe = lin(x, W["embeddings.patch_embedding.weight"].reshape(1024, -1), None, a=patch, wd=patch)
    h = torch.cat([W["embeddings.class_embedding"].expand(N, 1, -1), pe], 1) + W["embeddings.position_embedding.weight"][None]
    h = F.layer_norm(h, (1024,), W["pre_layrnorm.weight"], W["pre_layrnorm.bias"], 1e-5)
    L = orc.num_layers(sd)
    for i in range(L):
        p = f"encoder.layers.{i}."
        y = F.layer_norm(h, (1024,), W[p+"layer_norm1.weight"], W[p+"layer_norm1.bias"], 1e-5)
        q = rnd(lin(y, W[p+"self_attn.q_proj.weight"], W[p+"self_attn.q_proj.bias"]) * 0.125, act)
        k = rnd(lin(y, W[p+"self_attn.k_proj.weight"], W[p+"self_attn.k_proj.bias"]), act)
        v = rnd(lin(y, W[p+"self_attn.v_proj.weight"], W[p+"self_attn.v_proj.bias"]), act)
        q = q.view(N, 577, 16, 64).transpose(1, 2); k = k.view(N, 577, 16, 64).transpose(1, 2); v = v.view(N, 577, 16, 64).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        m = s.max(-1, keepdim=True).values
        e = torch.exp(s - m)
        o = (rnd(e, pdt) @ v) / e.sum(-1, keepdim=True)
        o = rnd(o.transpose(1, 2).reshape(N, 577, 1024), act)
        h = h + lin(o, W[p+"self_attn.out_proj.weight"], W[p+"self_attn.out_proj.bias"])
        y = F.layer_norm(h, (1024,), W[p+"layer_norm2.weight"], W[p+"layer_norm2.bias"], 1e-5)
        y = lin(y, W[p+"mlp.fc1.weight"], W[p+"mlp.fc1.bias"])
        y = rnd(y * torch.sigmoid(1.702 * y), act)
        h = h + lin(y, W[p+"mlp.fc2.weight"], W[p+"mlp.fc2.bias"])
    return h

if __name__ == "__main__":
    torch.set_num_threads(8)
    for (name, kw, n) in (("L2 jitter", dict(seed=11, layers=2, affine_jitter=True), 2),
                          ("L24 default", dict(seed=0, layers=24), 1),
                          ("L24 stress", dict(seed=5, layers=24, affine_jitter=True, scale=2.0), 1)):
        sd = synthetic.make_vit_weights(**kw)
        px = synthetic.make_pixels(4, seed=77)[:n]
        ref = orc.vit_last_hidden_state(sd, px)
        bf, hf = torch.bfloat16, torch.float16
        schemes = {
            "all bf16": dict(act=bf, wdt=bf, patch=bf, pdt=bf),
            "bf16, exact patch": dict(act=bf, wdt=bf, patch=None, pdt=bf),
            "act fp16, w bf16": dict(act=hf, wdt=bf, patch=hf, pdt=hf),
            "all fp16": dict(act=hf, wdt=hf, patch=hf, pdt=hf),
            "fp16, patch bf16": dict(act=hf, wdt=hf, patch=bf, pdt=hf),
        }
        for sname, s in schemes.items():
            t = time.time()
            h = vit_emul(sd, px, **s)
            print(f"{name:12s} {sname:20s} hidden rel {orc.rel_err(h, ref):.2e}  emb rel {orc.rel_err(h.mean(1), ref.mean(1)):.2e}  ({time.time()-t:.0f}s)", flush=True)
