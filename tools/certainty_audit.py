"""Audit of the certainty signal on many panoramas (round 5): the bench's model (24 layers, seed 0, head centred on the mean embedding,
sigma(logit) = 4, 1M-row bank, top-5 refinement), N batches of 128 fresh panoramas.  Ground truth = the same chain on the EXACT
encoder's embeddings (pg_vit_forward_precise, ~2e-7 of the fp32 reference).  Reported: how many panoramas the fast mode flags, how
many of its outputs differ from the truth, and whether any of those was flagged certain (must be none).
   python tools/certainty_audit.py [n_batches] [default|spread]      (spread: synthetic.make_vit_weights_spread(seed 31), the tower whose
                                                                      embeddings spread like a trained one's and whose 16-bit error is 7e-4)"""
import contextlib, io, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pigeon_amd import synthetic
from pigeon_amd.clip_embedder import HipCLIPVisionModel
from pigeon_amd.evaluate import certain_forward
from pigeon_amd.proto_refiner import ProtoRefiner
from pigeon_amd.super_guessr import SuperGuessr

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda"
C = 10000
weights = sys.argv[2] if len(sys.argv) > 2 else "default"
base = HipCLIPVisionModel(synthetic.make_vit_weights_spread(seed=31, layers=24) if weights == "spread"
                          else synthetic.make_vit_weights(seed=0, layers=24), layers=24)
print(f"tower: {weights}")
geo = os.path.join(tempfile.mkdtemp(prefix="pigeon_audit_"), "g.csv")
synthetic.write_geocell_csv(geo, synthetic.make_geocells(C, seed=0))
with contextlib.redirect_stdout(io.StringIO()):
    model = SuperGuessr(base, panorama=True, freeze_base=True, num_candidates=5, geocell_path=geo, exact_top1=False, margin_autocalibrate=False)
W, b = synthetic.make_head_weights(C, seed=0)
with torch.no_grad():
    model.cell_layer.weight.copy_(W); model.cell_layer.bias.copy_(b)
model.to(dev).eval()
refiner = ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, bank=synthetic.make_bank_device(C, 100, seed=2, device=dev), device=dev).eval()
g = torch.Generator(device=dev).manual_seed(4242)
px0 = torch.randn((128, 12, 336, 336), generator=g, device=dev)
with torch.no_grad():                                           # centre / scale the head as bench.py does
    pe = model(pixel_values=px0, labels_clf=None).embedding.mean(dim=1)
    center = pe.mean(dim=0)
    sig = float(((pe - center) @ model.cell_layer.weight.data.t()).std())
    model.cell_layer.weight.mul_(float(2.0 ** np.round(np.log2(4.0 / sig))))
    model.cell_layer.bias.copy_(b.to(dev) - model.cell_layer.weight.data @ center)
model.calibrate_certainty(px0, max_samples=128)
print(model.certainty.describe())
enc = base._encoder(torch.device(dev))
tot = dict(n=0, flagged=0, top1_bad=0, ref_bad=0, bad_certain=0, head=0, refine=0)
worst_z = 0.0
z_wrong, z_all = [], []
for i in range(nb):
    px = torch.randn((128, 12, 336, 336), generator=g, device=dev)
    out, info = certain_forward(model, refiner, pixel_values=px)          # fast mode: nothing re-encoded, certainty reported
    _, llh, cell = refiner(out.embedding, initial_preds=out.preds_LLH, candidate_cells=out.top5_geocells.indices,
                           candidate_probs=out.top5_geocells.values, quiet=True)
    emb_x = enc.forward_precise(px.reshape(-1, 3, 336, 336)).reshape(128, 4, 1024)
    ox = model.package(model.encode_head(embedding=emb_x))               # the same head / refiner on the exact embeddings
    _, llh_x, cell_x = refiner(ox.embedding, initial_preds=ox.preds_LLH, candidate_cells=ox.top5_geocells.indices,
                               candidate_probs=ox.top5_geocells.values, quiet=True)
    top1_bad = out.preds_geocell != ox.preds_geocell
    ref_bad = (cell != cell_x) | (llh != llh_x).any(dim=1)
    bad = top1_bad | ref_bad
    cert = info["certain"]
    tot["n"] += 128; tot["flagged"] += int((~cert).sum()); tot["top1_bad"] += int(top1_bad.sum()); tot["ref_bad"] += int(ref_bad.sum())
    tot["bad_certain"] += int((bad & cert).sum())
    tot["head"] += int((info["cause"] == 1).sum()); tot["refine"] += int((info["cause"] > 1).sum() + (info["cause"] < 0).sum())
    # how close the wrong ones came to being called certain: their tolerance in units of the calibrated residual error (a z-score;
    # certain means > kappa)
    z = torch.minimum(info["head_tol"], info["refine_tol"]) / model.certainty.rel_tol
    z_wrong += z[bad].clamp_min(0).tolist()
    z_all += z.clamp(0, 1e6).tolist()
print(tot)
zw, za = np.asarray(z_wrong), np.asarray(z_all)
edges = [0, 0.5, 1, 1.5, 2, 2.5, 3, 3.6, 1e9]
print("tolerance / rel_tol of the panoramas the fast mode gets WRONG (kappa = %.1f): max %.2f; histogram over %s: %s" %
      (model.certainty.kappa, zw.max() if zw.size else float("nan"), edges[:-1], np.histogram(zw, bins=edges)[0].tolist()))
print("the same of ALL panoramas: %s; so the share that is wrong per bin: %s" %
      (np.histogram(za, bins=edges)[0].tolist(),
       [round(float(a) / max(1, int(b)), 3) for a, b in zip(np.histogram(zw, bins=edges)[0], np.histogram(za, bins=edges)[0])]))
print(f"{tot['n']} panoramas: {tot['flagged']} flagged uncertain ({100.0 * tot['flagged'] / tot['n']:.2f} %: {tot['head']} for the head's top-1, "
      f"{tot['refine']} for the refiner); fast-mode outputs differing from the exact chain: top-1 {tot['top1_bad']}, refined {tot['ref_bad']}; "
      f"of those, flagged CERTAIN: {tot['bad_certain']}")
