"""Audit of the certainty signal against the REFERENCE MODULE (round 6).  tools/certainty_audit.py compares the fast
mode with this repo's own exact encoder; here the ground truth is computed from the pixels by the reference's chain itself:
transformers.CLIPVisionModel in fp32 with eager attention under stock PyTorch-ROCm on this GPU (the module the reference calls,
models/super_guessr.py:395; never part of the product) -> token mean -> the oracle's head and refinement on the CPU (oracle/
pigeon_oracle.py: the restatement pinned bit for bit to the real reference's outputs, tests/test_oracle_golden.py).
Against that truth, over N batches of 128 fresh panoramas of the bench's model (24 layers, seed 0, head centred, 1M-row bank, top-5):
  * the FAST mode: how many outputs differ, and whether any of those was called certain (must be none);
  * the PRODUCT (deferred exact tier): how many outputs differ at all (must be none), how many rows stay uncertain at the exact floor.
   python tools/certainty_audit_ref.py [n_batches] [default|spread]"""
import contextlib, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from transformers import CLIPVisionConfig, CLIPVisionModel
import bench
from oracle import pigeon_oracle as orc
from pigeon_amd import synthetic
from pigeon_amd.clip_embedder import HipCLIPVisionModel
from pigeon_amd.deferred import DeferredExact
from pigeon_amd.evaluate import certain_forward
from pigeon_amd.proto_refiner import ProtoRefiner
from pigeon_amd.super_guessr import SuperGuessr

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
weights = sys.argv[2] if len(sys.argv) > 2 else "default"
dev, C, B = "cuda", 10000, 128
sd = synthetic.make_vit_weights_spread(seed=31, layers=24) if weights == "spread" else synthetic.make_vit_weights(seed=0, layers=24)
base = HipCLIPVisionModel(sd, layers=24)
geo = os.path.join(tempfile.mkdtemp(prefix="pigeon_audit_"), "g.csv")
synthetic.write_geocell_csv(geo, synthetic.make_geocells(C, seed=0))
with contextlib.redirect_stdout(io.StringIO()):
    model = SuperGuessr(base, panorama=True, freeze_base=True, num_candidates=5, geocell_path=geo, exact_top1=True, margin_autocalibrate=False)
W, b = synthetic.make_head_weights(C, seed=0)
with torch.no_grad():
    model.cell_layer.weight.copy_(W); model.cell_layer.bias.copy_(b)
model.to(dev).eval()
bank_t = synthetic.make_bank_device(C, 100, seed=2, device=dev)
refiner = ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, bank=bank_t, device=dev).eval()
g = torch.Generator(device=dev).manual_seed(777)
px0 = torch.randn((B, 12, 336, 336), generator=g, device=dev)
model.exact_top1 = False
with torch.no_grad():                                           # centre / scale the head as bench.py does
    pe = model(pixel_values=px0, labels_clf=None).embedding.mean(dim=1)
    center = pe.mean(dim=0)
    if weights == "default":
        sig = float(((pe - center) @ model.cell_layer.weight.data.t()).std())
        model.cell_layer.weight.mul_(float(2.0 ** np.round(np.log2(4.0 / sig))))
    model.cell_layer.bias.copy_(b.to(dev) - model.cell_layer.weight.data @ center)
model.exact_top1 = True
model.calibrate_certainty(px0, max_samples=B)
print(f"tower: {weights}; {model.certainty.describe()}")
cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14,
                       projection_dim=768)
with contextlib.redirect_stdout(io.StringIO()):
    hf = CLIPVisionModel._from_config(cfg, attn_implementation="eager")
hf.load_state_dict(sd, strict=True)
hf = hf.to(dev).eval()
torch.backends.cuda.matmul.allow_tf32 = False
engine = DeferredExact(model, refiner)
ref_emb, fast, prod = [], [], {}
t0 = time.perf_counter()
with torch.no_grad():
    for i in range(nb):
        px = torch.randn((B, 12, 336, 336), generator=g, device=dev)
        flat = px.reshape(-1, 3, 336, 336)
        ref_emb.append(torch.cat([hf(pixel_values=flat[j:j + 32]).last_hidden_state.mean(dim=1) for j in range(0, flat.shape[0], 32)]).reshape(B, 4, 1024).cpu())
        model.exact_top1 = False
        out, info = certain_forward(model, refiner, pixel_values=px)        # the fast mode: flags only
        fast.append(dict(cell=out.preds_geocell.cpu(), rcell=info["refined_geocell"].cpu(), rllh=info["refined_LLH"].cpu(), certain=info["certain"].cpu(),
                         z=(torch.minimum(info["head_tol"], info["refine_tol"]) / model.certainty.rel_tol).cpu(), emb=out.embedding.cpu()))
        model.exact_top1 = True
        for r in engine.submit(px, meta=i):
            prod[r["meta"]] = {k: r[k].cpu() for k in ("preds_geocell", "refined_geocell", "refined_LLH", "certain", "exact")}
    for r in engine.flush():
        prod[r["meta"]] = {k: r[k].cpu() for k in ("preds_geocell", "refined_geocell", "refined_LLH", "certain", "exact")}
torch.cuda.synchronize()
print(f"{nb * B} panoramas through the reference module (fp32), the fast mode and the product in {time.perf_counter() - t0:.0f} s; "
      f"exact passes: {len(engine.flush_log)}, rows that did not fit the queue: {engine.check_nothing_dropped()}")
del hf
ref_emb = torch.cat(ref_emb)
Wc, bc, cen = model.cell_layer.weight.data.cpu(), model.cell_layer.bias.data.cpu(), model.lla_geocells.data.cpu()
o = orc.super_guessr_forward(Wc, bc, cen, 5, embedding=ref_emb)


class _B:
    pass


hb = _B()
hb.proto_emb, hb.train_emb, hb.train_lnglat = bench._LazyRows(bank_t["proto_emb"]), bench._LazyRows(bank_t["train_emb"]), bench._LazyRows(bank_t["train_lnglat"])
for kk in ("cell_off", "proto_count", "member_off", "member_idx"):
    setattr(hb, kk, bank_t[kk].cpu().numpy())
hb.proto_lnglat = bank_t["proto_lnglat"].cpu().numpy()
_, t_llh, t_cell = orc.proto_refiner_forward(hb, o["embedding"], o["preds_LLH"], o["topk"].indices, o["topk"].values, 5, 1.6, 1000)
t_top1 = o["preds_geocell"]
cat = lambda k: torch.cat([f[k] for f in fast])                           # noqa: E731
f_bad_top1 = cat("cell") != t_top1
f_bad_ref = (cat("rcell") != t_cell) | (cat("rllh") != t_llh).any(dim=1)
f_bad = f_bad_top1 | f_bad_ref
cert = cat("certain")
z = cat("z").clamp(0, 1e6)
emb_err = float((cat("emb") - ref_emb).norm() / ref_emb.norm())
pcat = lambda k: torch.cat([prod[i][k] for i in range(nb)])                # noqa: E731
p_bad_top1 = pcat("preds_geocell") != t_top1
p_bad_ref = (pcat("refined_geocell") != t_cell) | (pcat("refined_LLH") != t_llh).any(dim=1)
n = nb * B
edges = [0, 0.5, 1, 1.5, 2, 2.5, 3, 3.6, 1e9]
hz_w, hz_a = np.histogram(z[f_bad].numpy(), bins=edges)[0], np.histogram(z.numpy(), bins=edges)[0]
print(f"truth = reference module fp32 (GPU) + oracle head / refinement (CPU); fast-mode embedding error vs the reference module {emb_err:.2e}")
print(f"FAST mode, {n} panoramas: {int((~cert).sum())} flagged uncertain ({100.0 * float((~cert).float().mean()):.2f} %); outputs differing from the "
      f"reference chain: top-1 {int(f_bad_top1.sum())}, refined {int(f_bad_ref.sum())}; of those, called CERTAIN: {int((f_bad & cert).sum())}")
print(f"   tolerance / rel_tol of the wrong ones: max {float(z[f_bad].max()) if bool(f_bad.any()) else float('nan'):.2f} (kappa {model.certainty.kappa}); "
      f"histogram over {edges[:-1]}: wrong {hz_w.tolist()}, all {hz_a.tolist()}")
print(f"PRODUCT (deferred exact tier), {n} panoramas: re-encoded {int(pcat('exact').sum())} ({100.0 * float(pcat('exact').float().mean()):.2f} %); outputs "
      f"differing from the reference chain: top-1 {int(p_bad_top1.sum())}, refined {int(p_bad_ref.sum())}; still uncertain at the exact floor: "
      f"{int((~pcat('certain')).sum())} (of which wrong: {int(((p_bad_top1 | p_bad_ref) & ~pcat('certain')).sum())})")

# every panorama on which the PRODUCT differs from this execution of the reference: how close the reference's own decision was.  (The
# reference module's fp32 result is itself not a function of the inputs alone: GPU vs CPU execution of the same module differ by ~2e-7
# in the embedding, bench.py `vs_cpu_oracle_embedding_rel_err`.)
wrong = torch.nonzero(p_bad_top1 | p_bad_ref).flatten().tolist()
lg = o["logits"]
for i in wrong:
    top2 = torch.topk(lg[i], 2)
    print(f"   product != reference on panorama {i}: reference top-1 / top-2 logits {float(top2.values[0]):.6f} / {float(top2.values[1]):.6f} "
          f"(margin {float(top2.values[0] - top2.values[1]):.2e} at sigma(logit) {float(lg.std()):.2f}; cells {top2.indices.tolist()}), product cell "
          f"{int(pcat('preds_geocell')[i])}, called certain: {bool(pcat('certain')[i])}, re-encoded: {bool(pcat('exact')[i])}")
