"""Dump the fast and the exact encoder's panel-mean embeddings of the bench's pixel stream (for offline work on the error model):
   python tools/dump_fast_exact.py [n_batches] [weights: default|spread] -> gpurun_out/r05/fast_exact_<weights>.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pigeon_amd import hip_ops, synthetic

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
weights = sys.argv[2] if len(sys.argv) > 2 else "default"
sd = synthetic.make_vit_weights_spread(seed=31, layers=24) if weights == "spread" else synthetic.make_vit_weights(seed=0, layers=24)
enc = hip_ops.VitEncoder(sd, precise=True)
g = torch.Generator(device="cuda").manual_seed(1234)
fast, exact = [], []
for i in range(nb):
    px = torch.randn((128, 12, 336, 336), generator=g, device="cuda").reshape(512, 3, 336, 336)
    fast.append(enc(px).reshape(128, 4, 1024).mean(dim=1).cpu())
    exact.append(enc.forward_precise(px).reshape(128, 4, 1024).mean(dim=1).cpu())
out = os.path.join(ROOT, "gpurun_out", "r05", f"fast_exact_{weights}.npz")
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez(out, fast=torch.cat(fast).numpy(), exact=torch.cat(exact).numpy())
f, e = torch.cat(fast), torch.cat(exact)
print(out, f.shape, "rel err", float((f - e).norm() / e.norm()))
