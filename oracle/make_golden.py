"""Generate tests/golden/*.npz by running the REAL reference (from /root/reference) on seeded synthetic
inputs.  Authoring-container only (the GPU box has no /root/reference); the outputs are committed.

    python oracle/make_golden.py [--only vit2|vit24|vit24_trained|head|pipeline|pipeline24|pipeline24_wide|pipeline24_spread|refine|geo|refiner_cache]

Every fixture stores only small tensors (inputs are regenerated from their seeds by
``pigeon_amd.synthetic`` on both sides).  What runs for each fixture:

  vit2      reference CLIPEmbedding.forward (models/clip_embedder.py:79-89 -> :42-66) around a HF
            CLIPVisionModel (ViT-L/14-336 geometry, 2 layers, jittered bias/LN params), 4 images.
  vit24     same, full 24-layer default-init model (seed 0), 4 images of the bench pixel stream.
  head      reference SuperGuessr(None, panorama=True, num_candidates=50).forward(embedding=...)
            (models/super_guessr.py:350-483) with C = 10000 geocells.
  pipeline  reference SuperGuessr(vit2, panorama=True, freeze_base=True, num_candidates=50) on 2 panoramas
            -> reference ProtoRefiner.forward (models/proto_refiner.py:121-231) over a 200-cell bank.
  refine    reference ProtoRefiner.forward on 48 purpose-built queries (near-prototype embeddings, empty
            candidate cells, singleton and multi-member clusters) at both parameter settings used by
            the reference: class defaults (topk 5, T 1.6, 1000 km; proto_refiner.py:20-21) and
            evaluate()'s (T 0.6, 100000 km; evaluation/evaluate.py:79-80; topk 20 of 50 candidates).
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import reference_loader  # noqa: E402
from pigeon_amd import synthetic  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_vit(sd, layers):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=layers,
                           num_attention_heads=16, image_size=336, patch_size=14, projection_dim=768)
    m = CLIPVisionModel(cfg)
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    m.eval()
    return m


def refine_queries(bank, B, k, seed):
    """Queries that make refinement meaningful + exercise every branch."""
    rng = np.random.default_rng(seed)
    C = bank.num_cells
    nonempty = np.nonzero(np.diff(bank.cell_off) > 0)[0]
    empty = np.nonzero(np.diff(bank.cell_off) == 0)[0]
    emb = np.empty((B, 1024), dtype=np.float32)
    cands = np.empty((B, k), dtype=np.int64)
    for i in range(B):
        cells = rng.permutation(C)[:k]
        if len(empty) and i % 3 == 0:
            cells[rng.integers(0, min(k, 5))] = empty[rng.integers(0, len(empty))]   # an empty candidate up front
        if i % 7 == 0 and len(empty):
            cells[0] = empty[0]
        cands[i] = cells
        # query close to a random prototype of one of the first 5 candidate cells
        pick = [c for c in cells[:5] if bank.cell_off[c + 1] > bank.cell_off[c]]
        c = pick[rng.integers(0, len(pick))] if pick else nonempty[0]
        p = rng.integers(bank.cell_off[c], bank.cell_off[c + 1])
        emb[i] = bank.proto_emb[p] + 0.5 * rng.standard_normal(1024).astype(np.float32)
    probs = np.sort(rng.dirichlet(np.ones(k) * 0.3, size=B).astype(np.float32), axis=1)[:, ::-1].copy()
    init = np.stack([rng.uniform(-180, 180, B), rng.uniform(-90, 90, B)], axis=1)   # float64
    # make some initial predictions sit right next to a candidate cell's data so the km veto is mixed
    for i in range(0, B, 2):
        c = cands[i, 0]
        if bank.cell_off[c + 1] > bank.cell_off[c]:
            init[i] = bank.proto_lnglat[bank.cell_off[c]].astype(np.float64) + rng.normal(0, 2.0, 2)
    init[:, 1] = np.clip(init[:, 1], -89.5, 89.5)
    return torch.from_numpy(emb), torch.from_numpy(cands), torch.from_numpy(probs), torch.from_numpy(init)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    tmp = tempfile.mkdtemp(prefix="pigeon_golden_")

    # bank + geocells for the refiner fixtures (200 cells x 12 protos)
    C_small = 200
    bank = synthetic.make_bank(C_small, 12, seed=2, empty_frac=0.05)
    proto_csv = os.path.join(tmp, "protos.csv")
    ds_dir = os.path.join(tmp, "hf_train")
    geo_small = synthetic.make_geocells(C_small, seed=0)
    geo_small_csv = os.path.join(tmp, "geocells_small.csv")
    synthetic.write_geocell_csv(geo_small_csv, geo_small)
    synthetic.write_bank_reference_files(bank, proto_csv, ds_dir)

    def want(name):
        return args.only is None or args.only == name

    if want("vit2"):
        ns = reference_loader.load(geo_small_csv, proto_csv, ds_dir)
        sd = synthetic.make_vit_weights(seed=11, layers=2, affine_jitter=True)
        vit = hf_vit(sd, 2)
        emb_ref = reference_loader.make_reference_embedder(ns, vit)
        px = synthetic.make_pixels(4, seed=77)
        with torch.no_grad():
            e = emb_ref(px)
            lhs = vit(pixel_values=px).last_hidden_state
        np.savez(os.path.join(GOLD, "vit2.npz"), embedding=e.numpy(),
                 lhs_rows=lhs[:, [0, 1, 2, 288, 575, 576]].numpy(),
                 meta=np.array([11, 2, 1, 4, 77]))   # weight seed, layers, jitter, n images, pixel seed
        print("vit2", e.shape, float(e.abs().mean()))

    if want("vit24"):
        ns = reference_loader.load(geo_small_csv, proto_csv, ds_dir)
        sd = synthetic.make_vit_weights(seed=0, layers=24)
        vit = hf_vit(sd, 24)
        emb_ref = reference_loader.make_reference_embedder(ns, vit)
        px = synthetic.make_pixels(4, seed=1234)
        with torch.no_grad():
            e = emb_ref(px)
        np.savez(os.path.join(GOLD, "vit24.npz"), embedding=e.numpy(), meta=np.array([0, 24, 0, 4, 1234]))
        print("vit24", e.shape, float(e.abs().mean()))
        # stress variant: jittered affine + 3x projection scale -> far less benign numerics
        sd = synthetic.make_vit_weights(seed=5, layers=24, affine_jitter=True, scale=2.0)
        vit = hf_vit(sd, 24)
        emb_ref = reference_loader.make_reference_embedder(ns, vit)
        px = synthetic.make_pixels(2, seed=99)
        with torch.no_grad():
            e = emb_ref(px)
        np.savez(os.path.join(GOLD, "vit24_stress.npz"), embedding=e.numpy(), meta=np.array([5, 24, 1, 2, 99]))
        print("vit24_stress", e.shape, float(e.abs().mean()))

    if want("head"):
        C = 10000
        geo = synthetic.make_geocells(C, seed=0)
        geo_csv = os.path.join(tmp, "geocells.csv")
        synthetic.write_geocell_csv(geo_csv, geo)
        ns = reference_loader.load(geo_csv, proto_csv, ds_dir)
        model = ns.SuperGuessr(None, panorama=True, num_candidates=50)
        W, b = synthetic.make_head_weights(C, seed=0)
        with torch.no_grad():
            model.cell_layer.weight.copy_(W)
            model.cell_layer.bias.copy_(b)
        model.eval()
        g = torch.Generator().manual_seed(321)
        emb = torch.randn((32, 4, 1024), generator=g) * 0.7 + 0.1
        with torch.no_grad():
            out = model(embedding=emb, labels=torch.zeros(32, 2, dtype=torch.float64),
                        labels_clf=torch.zeros(32, dtype=torch.long))
        logits = model.cell_layer(emb.mean(dim=1))
        # soft labels by distance (models/super_guessr.py:469-471: haversine_matrix + smooth_labels feeding the cross entropy)
        model_s = ns.SuperGuessr(None, panorama=True, num_candidates=50, should_smooth_labels=True)
        with torch.no_grad():
            model_s.cell_layer.weight.copy_(W)
            model_s.cell_layer.bias.copy_(b)
        model_s.eval()
        rng = np.random.default_rng(77)
        lab = torch.from_numpy(np.stack([rng.uniform(-180, 180, 32), rng.uniform(-90, 90, 32)], axis=1))
        with torch.no_grad():
            out_s = model_s(embedding=emb, labels=lab, labels_clf=torch.zeros(32, dtype=torch.long))
            out_s32 = model_s(embedding=emb, labels=lab.float(), labels_clf=torch.zeros(32, dtype=torch.long))
        np.savez(os.path.join(GOLD, "head.npz"), smooth_labels_in=lab.numpy(), loss_clf_smooth=float(out_s.loss_clf),
                 loss_clf_smooth_f32labels=float(out_s32.loss_clf),
                 preds_LLH=out.preds_LLH.numpy(), preds_geocell=out.preds_geocell.numpy(),
                 topk_values=out.top5_geocells.values.numpy(), topk_indices=out.top5_geocells.indices.numpy(),
                 logits_first8=logits[:, :8].detach().numpy(), loss_clf=float(out.loss_clf),
                 meta=np.array([C, 0, 32, 321, 50]))
        print("head", out.preds_geocell[:8].tolist())

    if want("pipeline"):
        ns = reference_loader.load(geo_small_csv, proto_csv, ds_dir)
        sd = synthetic.make_vit_weights(seed=11, layers=2, affine_jitter=True)
        vit = hf_vit(sd, 2)
        model = ns.SuperGuessr(vit, panorama=True, hierarchical=False, multi_task=False, heading=False,
                               freeze_base=True, num_candidates=50)
        W, b = synthetic.make_head_weights(C_small, seed=3)
        with torch.no_grad():
            model.cell_layer.weight.copy_(W * 8)      # sharper softmax -> non-degenerate candidate probs
            model.cell_layer.bias.copy_(b)
        model.eval()
        px = synthetic.make_pixels(8, seed=55, panorama=True)       # (2,12,336,336)
        with torch.no_grad():
            out = model(pixel_values=px, labels=torch.zeros(2, 2, dtype=torch.float64),
                        labels_clf=torch.zeros(2, dtype=torch.long))
        refiner = ns.ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6,
                                  proto_path=proto_csv, dataset_path=ds_dir)
        refiner.eval()
        with torch.no_grad():
            _, r_llh, r_cell = refiner(out.embedding, initial_preds=out.preds_LLH,
                                       candidate_cells=out.top5_geocells.indices,
                                       candidate_probs=out.top5_geocells.values)
        np.savez(os.path.join(GOLD, "pipeline.npz"),
                 embedding=out.embedding.numpy(), preds_LLH=out.preds_LLH.numpy(),
                 preds_geocell=out.preds_geocell.numpy(), topk_values=out.top5_geocells.values.numpy(),
                 topk_indices=out.top5_geocells.indices.numpy(),
                 refined_LLH=r_llh.numpy(), refined_cell=r_cell.numpy(),
                 meta=np.array([11, 2, 1, 8, 55, C_small, 12, 2, 3]))
        print("pipeline", out.preds_geocell.tolist(), r_llh.tolist())

    if want("refine"):
        ns = reference_loader.load(geo_small_csv, proto_csv, ds_dir)
        B, k = 48, 50
        emb, cands, probs, init = refine_queries(bank, B, k, seed=9)
        save = dict(embedding=emb.numpy(), candidate_cells=cands.numpy(), candidate_probs=probs.numpy(),
                    initial_preds=init.numpy(), meta=np.array([C_small, 12, 2]))
        base = ns.ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6,
                               proto_path=proto_csv, dataset_path=ds_dir)
        base.eval()
        # sanity: the prototypes the reference built == the bank's (same mean-of-members arithmetic)
        worst = 0.0
        for c in range(C_small):
            pr = base.protos[c]
            s, e = int(bank.cell_off[c]), int(bank.cell_off[c + 1])
            if pr is None:
                assert s == e
                continue
            ref_e = pr["embedding"][:].numpy()
            worst = max(worst, float(np.abs(ref_e - bank.proto_emb[s:e]).max()))
        print("max |reference proto - bank proto| =", worst)
        save["proto_build_max_abs_diff"] = np.array(worst)
        for tag, (topk, T, mr) in dict(default=(5, 1.6, 1000), evaluate=(20, 0.6, 100000),
                                       tight=(5, 1.0, 50)).items():
            ref = ns.ProtoRefiner(topk=topk, max_refinement=mr, temperature=T, proto_path=proto_csv,
                                  dataset_path=ds_dir, protos=base.protos)
            ref.eval()
            with torch.no_grad():
                _, llh, cell = ref(emb, initial_preds=init, candidate_cells=cands, candidate_probs=probs)
            # 3-D embedding input + no candidate_probs variants (proto_refiner.py:139-145)
            save[f"{tag}_LLH"] = llh.numpy()
            save[f"{tag}_cell"] = cell.numpy()
            save[f"{tag}_params"] = np.array([topk, T, mr], dtype=np.float64)
            print("refine", tag, cell[:10].tolist())
        ref = ns.ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, proto_path=proto_csv,
                              dataset_path=ds_dir, protos=base.protos)
        ref.eval()
        emb3 = emb[:, None, :] + torch.tensor([0.1, -0.1, 0.2, -0.2])[None, :, None]
        with torch.no_grad():
            _, llh, cell = ref(emb3, initial_preds=init, candidate_cells=cands, candidate_probs=None)
        save["noprobs3d_LLH"] = llh.numpy()
        save["noprobs3d_cell"] = cell.numpy()
        # veto edge: initial predictions placed max_refinement +- a few metres away from the point the refiner proposes, so
        # that the `distance > max_refinement` test (proto_refiner.py:198-205) is decided by the last digits of the
        # haversine -- which the reference evaluates with the refined point in float32 (deg2rad, cos) and the rest in float64
        ref = ns.ProtoRefiner(topk=5, max_refinement=10 ** 9, temperature=1.6, proto_path=proto_csv,
                              dataset_path=ds_dir, protos=base.protos)
        ref.eval()
        with torch.no_grad():
            _, prop, _ = ref(emb, initial_preds=init, candidate_cells=cands, candidate_probs=probs)   # no veto: proposals
        rng = np.random.default_rng(123)
        Rkm = 6378.137
        deltas = np.tile(np.array([-0.004, -0.002, -0.001, -0.0003, 0.0003, 0.001, 0.002, 0.004]), B // 8 + 1)[:B]
        p = prop.numpy().astype(np.float64)
        lam1, phi1 = np.radians(p[:, 0]), np.radians(p[:, 1])
        brg = rng.uniform(0, 2 * np.pi, B)
        ang = (1000.0 + deltas) / Rkm
        phi2 = np.arcsin(np.sin(phi1) * np.cos(ang) + np.cos(phi1) * np.sin(ang) * np.cos(brg))
        lam2 = lam1 + np.arctan2(np.sin(brg) * np.sin(ang) * np.cos(phi1), np.cos(ang) - np.sin(phi1) * np.sin(phi2))
        init_edge = np.stack([(np.degrees(lam2) + 540) % 360 - 180, np.degrees(phi2)], axis=1)
        ref.max_refinement = 1000
        with torch.no_grad():
            _, llh, cell = ref(emb, initial_preds=torch.from_numpy(init_edge), candidate_cells=cands, candidate_probs=probs)
        d_ref = ns.haversine(torch.from_numpy(init_edge), prop).numpy()
        save["vetoedge_init"], save["vetoedge_LLH"], save["vetoedge_cell"] = init_edge, llh.numpy(), cell.numpy()
        save["vetoedge_dist"] = d_ref
        d64 = ns.haversine(torch.from_numpy(init_edge), prop.double()).numpy()
        print("refine vetoedge: vetoed", int((d_ref > 1000).sum()), "of", B, "; decisions an all-float64 haversine would flip:",
              int(((d_ref > 1000) != (d64 > 1000)).sum()), "max |mixed - f64| km", float(np.abs(d_ref - d64).max()))
        np.savez(os.path.join(GOLD, "refine.npz"), **save)

    if want("geo"):
        # the reference's own great-circle / smoothing / metric helpers on seeded inputs (pins oracle/geo_oracle.py and, through
        # it or directly, pg_haversine_matrix / pg_haversine_pairs / pg_smooth_labels and evaluate.compute_geoguessr_metrics)
        ns = reference_loader.load(geo_small_csv, proto_csv, ds_dir)
        mt = reference_loader.load_metrics()
        rng = np.random.default_rng(17)
        N, M = 24, 300
        x64 = np.stack([rng.uniform(-180, 180, N), rng.uniform(-90, 90, N)], axis=1)
        y64 = np.stack([rng.uniform(-180, 180, M), rng.uniform(-90, 90, M)], axis=1)
        x64[0] = y64[0]                                   # zero distance
        x64[1] = [(y64[1, 0] + 360) % 360 - 180, -y64[1, 1]]                      # antipode of y[1]
        x32 = x64.astype(np.float32)
        save = dict(x=x64, y=y64)
        tx64, tx32, ty = torch.from_numpy(x64), torch.from_numpy(x32), torch.from_numpy(y64)
        save["matrix_f64"] = ns.haversine_matrix(tx64, ty.t()).numpy()           # (N,M) f64
        save["matrix_f32x"] = ns.haversine_matrix(tx32, ty.t()).numpy()          # fp32 labels, f64 geocells -> f64
        d = torch.from_numpy(save["matrix_f64"]).clone()
        d[2, 5] = float("nan"); d[3, :] = float("inf"); d[4, 7] = float("inf")
        save["smooth_in"] = d.numpy()
        save["smooth_out"] = ns.preprocessing.smooth_labels(d).numpy()           # LABEL_SMOOTHING_CONSTANT = 65
        save["smooth_constant"] = np.array(float(ns.config.LABEL_SMOOTHING_CONSTANT))
        # paired form with the refiner's dtypes: x float64, y float32 (and float64)
        yp64 = y64[:N]
        save["pairs_f32y"] = ns.haversine(tx64, torch.from_numpy(yp64.astype(np.float32))).numpy()
        save["pairs_f64y"] = ns.haversine(tx64, torch.from_numpy(yp64)).numpy()
        # metrics on 500 synthetic predictions
        n = 500
        labels = np.stack([rng.uniform(-180, 180, n), rng.uniform(-90, 90, n)], axis=1)
        scale = np.exp(rng.uniform(np.log(1e-3), np.log(60), n))[:, None]
        preds = labels + rng.normal(0, 1, (n, 2)) * scale
        preds[:, 1] = np.clip(preds[:, 1], -90, 90)
        preds = preds.astype(np.float32)                  # what evaluate_model collects (refiner output is fp32)
        cell_labels = rng.integers(0, 50, n)
        top5 = np.stack([rng.permutation(50)[:5] for _ in range(n)])
        cell_preds = top5[:, 0]
        dist = mt.haversine_np(preds, labels)
        m = {"Mean_km_error": np.mean(dist), "Median_km_error": np.median(dist),
             "Geoguessr_score": mt.geoguessr_score(dist),
             "Geocell_top5_accuracy": mt.topk_geocell_accuracy(cell_labels, top5)}
        for km in (1, 5, 10, 25, 50, 100, 200, 750, 1000, 2500):
            m[f"Under_{km}_km"] = mt.percentage_within_radius(dist, km)
        save.update(metric_preds=preds, metric_labels=labels, metric_cell_labels=cell_labels, metric_top5=top5,
                    metric_cell_preds=cell_preds, metric_distances=dist,
                    metric_names=np.array(sorted(m)), metric_values=np.array([float(m[k]) for k in sorted(m)]))
        np.savez(os.path.join(GOLD, "geo.npz"), **save)
        print("geo", save["matrix_f64"].shape, {k: round(float(v), 4) for k, v in m.items()})

    if want("vit24_trained"):
        # trained-regime stress: massive-activation channels + rows with |mean|/std >= 5 (synthetic.make_vit_weights_trained_like)
        ns = reference_loader.load(geo_small_csv, proto_csv, ds_dir)
        sd = synthetic.make_vit_weights_trained_like(seed=21, layers=24)
        vit = hf_vit(sd, 24)
        emb_ref = reference_loader.make_reference_embedder(ns, vit)
        px = synthetic.make_pixels(2, seed=314)
        with torch.no_grad():
            e = emb_ref(px)
            hs = vit(pixel_values=px, output_hidden_states=True).hidden_states
        # regime statistics of the residual stream the fixture really produces (layer 12 input), recorded for the test
        h = hs[12][0]
        ratio = (h.mean(dim=1).abs() / h.std(dim=1)).numpy()
        np.savez(os.path.join(GOLD, "vit24_trained.npz"), embedding=e.numpy(), meta=np.array([21, 24, 1, 2, 314]),
                 absmax_layer12=float(h.abs().max()), mean_over_std_median=float(np.median(ratio)),
                 mean_over_std_min=float(ratio.min()))
        print("vit24_trained", e.shape, float(e.abs().mean()), "absmax", float(h.abs().max()), "|mean|/std median",
              float(np.median(ratio)), "min", float(ratio.min()))

    if want("pipeline24"):
        # END TO END at full size: the real reference, 24 layers (the bench weights, seed 0), C = 10 000 geocells, 32 panoramas of
        # the seed-1234 pixel stream, num_candidates = 50; refinement at the class defaults (top-5, T 1.6, 1000 km) AND at
        # evaluate()'s settings (topk 40 of 50, T 0.6, 100000 km; evaluation/evaluate.py:44,79-80).
        C, NP = 10000, 32
        geo = synthetic.make_geocells(C, seed=0)
        geo_csv = os.path.join(tmp, "geocells10k.csv")
        synthetic.write_geocell_csv(geo_csv, geo)
        ns = reference_loader.load(geo_csv, proto_csv, ds_dir)
        sd = synthetic.make_vit_weights(seed=0, layers=24)
        vit = hf_vit(sd, 24)
        model = ns.SuperGuessr(vit, panorama=True, hierarchical=False, multi_task=False, heading=False,
                               freeze_base=True, num_candidates=50)
        model.eval()
        px = synthetic.make_pixels(4 * NP, seed=1234, panorama=True)          # (32,12,336,336)
        lab, labc = torch.zeros(NP, 2, dtype=torch.float64), torch.zeros(NP, dtype=torch.long)
        import time
        t0 = time.time()
        with torch.no_grad():
            outs = [model(pixel_values=px[i:i + 4], labels=lab[i:i + 4], labels_clf=labc[i:i + 4]) for i in range(0, NP, 4)]
        emb = torch.cat([o.embedding for o in outs])                         # (32,4,1024) reference embeddings
        print(f"pipeline24: reference ViT pass {time.time() - t0:.0f} s")
        # Head with realistic margins: a random-init ViT maps every image to nearly the same embedding (|e_i - mean| ~ 0.1 |e|),
        # so an untrained head would send all panoramas to one cell.  Centre the head on the mean embedding (bias = b0 - W.center)
        # and scale it so the logits spread with sigma = 4: distinct argmax cells, top-1 probabilities 0.05 .. 0.9, honest near-ties.
        pe = emb.mean(dim=1)
        center = pe.mean(dim=0)
        radius = float((pe - center).norm(dim=1).mean())
        W0, b0 = synthetic.make_head_weights(C, seed=0)
        sig = float(((pe - center) @ W0.t()).std())
        scale = float(2.0 ** np.round(np.log2(4.0 / sig)))
        W = W0 * scale
        bias = b0 - W @ center
        with torch.no_grad():
            model.cell_layer.weight.copy_(W)
            model.cell_layer.bias.copy_(bias)
            base, model.base_model = model.base_model, None                  # head pass on the reference's own embeddings
            out = model(embedding=emb, labels=lab, labels_clf=labc)
            model.base_model = base
            # one panorama again from pixels with the final head: same numbers through the one-call path
            chk = model(pixel_values=px[:2], labels=lab[:2], labels_clf=labc[:2])
        assert torch.equal(chk.preds_geocell, out.preds_geocell[:2]) and torch.equal(chk.embedding, emb[:2])
        logits = model.cell_layer(pe)
        top2 = torch.topk(logits, 2, dim=-1).values
        margin = (top2[:, 0] - top2[:, 1]).detach()
        print("pipeline24 cells", out.preds_geocell.tolist())
        print("pipeline24 top-1 prob", [round(float(v), 3) for v in out.top5_geocells.values[:, 0]])
        print("pipeline24 logit margins", [round(float(v), 3) for v in margin], "scale", scale, "radius", radius)
        # prototype bank around the embedding cloud (synthetic.make_bank center/radius), 10 000 cells x 4 prototypes
        bank24 = synthetic.make_bank(C, 4, seed=2, empty_frac=0.01, max_members=3, center=center.numpy(), radius=radius)
        proto24 = os.path.join(tmp, "protos24.csv")
        ds24 = os.path.join(tmp, "hf_train24")
        synthetic.write_bank_reference_files(bank24, proto24, ds24)
        ref = ns.ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, proto_path=proto24, dataset_path=ds24,
                              protos=[None] * C)
        # the reference builds prototypes per geocell (`_get_prototypes`, proto_refiner.py:288-313); only candidate cells are
        # ever consulted (:167), so only those are built here (10 000 Arrow round trips otherwise)
        import datasets as _ds
        _ds.disable_progress_bar()
        needed = sorted(set(out.top5_geocells.indices.flatten().tolist()))
        # (collected in a separate list and attached afterwards: `datasets` fingerprints the mapped method -- i.e. pickles the
        # whole refiner including .protos -- on every call, which would make the loop quadratic)
        built = [None] * C
        t0 = time.time()
        for n_done, c in enumerate(needed):
            built[c] = ref._get_prototypes(c)
            if n_done % 200 == 0:
                print(f"pipeline24: prototypes of {n_done}/{len(needed)} candidate cells, {time.time() - t0:.0f} s", flush=True)
        ref.protos = built
        save = dict(embedding=emb.numpy(), head_bias=bias.numpy(), center=center.numpy(),
                    meta=np.array([0, 24, NP, 1234, C, 4, 2, 3]), head_scale=np.array(scale), radius=np.array(radius),
                    preds_LLH=out.preds_LLH.numpy(), preds_geocell=out.preds_geocell.numpy(),
                    topk_values=out.top5_geocells.values.numpy(), topk_indices=out.top5_geocells.indices.numpy(),
                    logit_margin=margin.numpy())
        for tag, (topk, T, mr) in dict(default=(5, 1.6, 1000), evaluate=(40, 0.6, 100000)).items():
            ref.topk, ref.max_refinement = topk, mr
            ref.temperature.data = torch.tensor(T)
            ref.eval()
            with torch.no_grad():
                _, llh, cell = ref(out.embedding, initial_preds=out.preds_LLH, candidate_cells=out.top5_geocells.indices,
                                   candidate_probs=out.top5_geocells.values)
            save[f"{tag}_LLH"], save[f"{tag}_cell"] = llh.numpy(), cell.numpy()
            save[f"{tag}_params"] = np.array([topk, T, mr], dtype=np.float64)
            print("pipeline24 refine", tag, "changed", int((cell != out.preds_geocell).sum()), "of", NP)
        np.savez(os.path.join(GOLD, "pipeline24.npz"), **save)

    if want("pipeline24_wide"):
        # Round 3: the END-TO-END top-1 question at 128 panoramas (= one full bench step, 512 images), asked of the REAL reference:
        # pixels -> reference SuperGuessr(24-layer ViT) -> geocell argmax, with the head centred/scaled as in pipeline24 over all
        # 128 panoramas.  Stored per panorama: the reference's top-8 logits + cells (so a GPU run can measure its own logit
        # error on exactly the cells that decide the argmax) and the top-1/top-2 margin.  Refinement at the class defaults
        # (top-5, T 1.6, 1000 km) over all 128; evaluate()'s settings stay with the 32-panorama fixture (prototype builds of
        # 50 candidates x 128 panoramas through Arrow would take an hour).
        C, NP = 10000, 128
        geo = synthetic.make_geocells(C, seed=0)
        geo_csv = os.path.join(tmp, "geocells10k.csv")
        synthetic.write_geocell_csv(geo_csv, geo)
        ns = reference_loader.load(geo_csv, proto_csv, ds_dir)
        sd = synthetic.make_vit_weights(seed=0, layers=24)
        vit = hf_vit(sd, 24)
        model = ns.SuperGuessr(vit, panorama=True, hierarchical=False, multi_task=False, heading=False,
                               freeze_base=True, num_candidates=50)
        model.eval()
        px = synthetic.make_pixels(4 * NP, seed=4321, panorama=True)          # (128,12,336,336); another stream than pipeline24's
        lab, labc = torch.zeros(NP, 2, dtype=torch.float64), torch.zeros(NP, dtype=torch.long)
        import time
        t0 = time.time()
        outs = []
        with torch.no_grad():
            for i in range(0, NP, 4):
                outs.append(model(pixel_values=px[i:i + 4], labels=lab[i:i + 4], labels_clf=labc[i:i + 4]))
                if i % 16 == 0:
                    print(f"pipeline24_wide: reference ViT {i + 4}/{NP} panoramas, {time.time() - t0:.0f} s", flush=True)
        emb = torch.cat([o.embedding for o in outs])                         # (128,4,1024)
        pe = emb.mean(dim=1)
        center = pe.mean(dim=0)
        radius = float((pe - center).norm(dim=1).mean())
        W0, b0 = synthetic.make_head_weights(C, seed=0)
        sig = float(((pe - center) @ W0.t()).std())
        scale = float(2.0 ** np.round(np.log2(4.0 / sig)))
        W = W0 * scale
        bias = b0 - W @ center
        with torch.no_grad():
            model.cell_layer.weight.copy_(W)
            model.cell_layer.bias.copy_(bias)
            base, model.base_model = model.base_model, None
            out = model(embedding=emb, labels=lab, labels_clf=labc)
            model.base_model = base
            chk = model(pixel_values=px[:2], labels=lab[:2], labels_clf=labc[:2])   # the one-call path with the final head
            logits = model.cell_layer(pe)
        assert torch.equal(chk.preds_geocell, out.preds_geocell[:2]) and torch.equal(chk.embedding, emb[:2])
        top8 = torch.topk(logits, 8, dim=-1)
        margin = (top8.values[:, 0] - top8.values[:, 1]).detach()
        assert torch.equal(top8.indices[:, 0], out.preds_geocell)
        print("pipeline24_wide distinct cells", int(torch.unique(out.preds_geocell).numel()), "margins min/median",
              float(margin.min()), float(margin.median()), "scale", scale, "radius", radius)
        print("pipeline24_wide sorted margins (first 16)", [round(float(v), 4) for v in torch.sort(margin).values[:16]])
        bank24 = synthetic.make_bank(C, 4, seed=2, empty_frac=0.01, max_members=3, center=center.numpy(), radius=radius)
        proto24 = os.path.join(tmp, "protos24w.csv")
        ds24 = os.path.join(tmp, "hf_train24w")
        synthetic.write_bank_reference_files(bank24, proto24, ds24)
        ref = ns.ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, proto_path=proto24, dataset_path=ds24,
                              protos=[None] * C)
        import datasets as _ds
        _ds.disable_progress_bar()
        needed = sorted(set(out.top5_geocells.indices[:, :5].flatten().tolist()))
        built = [None] * C
        t0 = time.time()
        for n_done, c in enumerate(needed):
            built[c] = ref._get_prototypes(c)
            if n_done % 100 == 0:
                print(f"pipeline24_wide: prototypes of {n_done}/{len(needed)} candidate cells, {time.time() - t0:.0f} s", flush=True)
        ref.protos = built
        ref.eval()
        with torch.no_grad():
            _, llh, cell = ref(out.embedding, initial_preds=out.preds_LLH, candidate_cells=out.top5_geocells.indices,
                               candidate_probs=out.top5_geocells.values)
        print("pipeline24_wide refine default changed", int((cell != out.preds_geocell).sum()), "of", NP)
        np.savez(os.path.join(GOLD, "pipeline24_wide.npz"), embedding=emb.numpy(), head_bias=bias.numpy(), center=center.numpy(),
                 meta=np.array([0, 24, NP, 4321, C, 4, 2, 3]), head_scale=np.array(scale), radius=np.array(radius),
                 preds_LLH=out.preds_LLH.numpy(), preds_geocell=out.preds_geocell.numpy(),
                 topk_values=out.top5_geocells.values.numpy(), topk_indices=out.top5_geocells.indices.numpy(),
                 top8_logits=top8.values.detach().numpy(), top8_cells=top8.indices.numpy(), logit_margin=margin.numpy(),
                 default_LLH=llh.numpy(), default_cell=cell.numpy(), default_params=np.array([5, 1.6, 1000], dtype=np.float64))

    if want("pipeline24_spread"):
        # Round 4 (VERDICT r03 next #1a): the top-1 question on a tower whose embeddings SPREAD like a trained one's
        # (synthetic.make_vit_weights_spread: input-selected global attention; pairwise cos-sim of the image embeddings ~0.7
        # instead of the 0.96 of a default-init tower), through the REAL reference at 24 layers on 128 panoramas (512 images),
        # with the head at its NATURAL scale: nn.Linear's default init, no centring, no scaling.  Stored: reference embeddings,
        # top-8 logits + cells, margins, the per-panorama sigma of the logits over the cells (so that a GPU run can quote its logit
        # error in units of sigma), cos-sim statistics of the embeddings, default refinement of all 128.
        C, NP = 10000, 128
        geo = synthetic.make_geocells(C, seed=0)
        geo_csv = os.path.join(tmp, "geocells10k.csv")
        synthetic.write_geocell_csv(geo_csv, geo)
        ns = reference_loader.load(geo_csv, proto_csv, ds_dir)
        sd = synthetic.make_vit_weights_spread(seed=31, layers=24)
        vit = hf_vit(sd, 24)
        model = ns.SuperGuessr(vit, panorama=True, hierarchical=False, multi_task=False, heading=False,
                               freeze_base=True, num_candidates=50)
        W0, b0 = synthetic.make_head_weights(C, seed=0)
        with torch.no_grad():
            model.cell_layer.weight.copy_(W0)
            model.cell_layer.bias.copy_(b0)
        model.eval()
        px = synthetic.make_pixels(4 * NP, seed=9731, panorama=True)          # (128,12,336,336)
        lab, labc = torch.zeros(NP, 2, dtype=torch.float64), torch.zeros(NP, dtype=torch.long)
        import time
        t0 = time.time()
        outs = []
        with torch.no_grad():
            for i in range(0, NP, 4):
                outs.append(model(pixel_values=px[i:i + 4], labels=lab[i:i + 4], labels_clf=labc[i:i + 4]))
                if i % 16 == 0:
                    print(f"pipeline24_spread: reference {i + 4}/{NP} panoramas, {time.time() - t0:.0f} s", flush=True)
        emb = torch.cat([o.embedding for o in outs])                         # (128,4,1024)
        preds_geocell = torch.cat([o.preds_geocell for o in outs])
        preds_LLH = torch.cat([o.preds_LLH for o in outs])
        tk_val = torch.cat([o.top5_geocells.values for o in outs])
        tk_idx = torch.cat([o.top5_geocells.indices for o in outs])
        pe = emb.mean(dim=1)
        with torch.no_grad():
            logits = model.cell_layer(pe)
        top8 = torch.topk(logits, 8, dim=-1)
        margin = (top8.values[:, 0] - top8.values[:, 1]).detach()
        sigma = logits.std(dim=1).detach()
        assert torch.equal(top8.indices[:, 0], preds_geocell)
        ie = emb.reshape(-1, 1024)
        ien = ie / ie.norm(dim=1, keepdim=True)
        cs = (ien @ ien.t())[~torch.eye(ie.shape[0], dtype=torch.bool)]
        pen = pe / pe.norm(dim=1, keepdim=True)
        csp = (pen @ pen.t())[~torch.eye(NP, dtype=torch.bool)]
        print("pipeline24_spread image cos-sim mean/min/max", float(cs.mean()), float(cs.min()), float(cs.max()),
              "panorama cos-sim mean/max", float(csp.mean()), float(csp.max()))
        print("pipeline24_spread distinct cells", int(torch.unique(preds_geocell).numel()), "sigma(logit) mean", float(sigma.mean()),
              "margin/sigma min/median", float((margin / sigma).min()), float((margin / sigma).median()))
        print("pipeline24_spread sorted margin/sigma (first 16)", [round(float(v), 5) for v in torch.sort(margin / sigma).values[:16]])
        center = pe.mean(dim=0)
        radius = float((pe - center).norm(dim=1).mean())
        bank24 = synthetic.make_bank(C, 4, seed=2, empty_frac=0.01, max_members=3, center=center.numpy(), radius=radius)
        proto24 = os.path.join(tmp, "protos24s.csv")
        ds24 = os.path.join(tmp, "hf_train24s")
        synthetic.write_bank_reference_files(bank24, proto24, ds24)
        ref = ns.ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, proto_path=proto24, dataset_path=ds24,
                              protos=[None] * C)
        import datasets as _ds
        _ds.disable_progress_bar()
        needed = sorted(set(tk_idx[:, :5].flatten().tolist()))
        built = [None] * C
        t0 = time.time()
        for n_done, c in enumerate(needed):
            built[c] = ref._get_prototypes(c)
            if n_done % 100 == 0:
                print(f"pipeline24_spread: prototypes of {n_done}/{len(needed)} candidate cells, {time.time() - t0:.0f} s", flush=True)
        ref.protos = built
        ref.eval()
        with torch.no_grad():
            _, llh, cell = ref(emb, initial_preds=preds_LLH, candidate_cells=tk_idx, candidate_probs=tk_val)
        print("pipeline24_spread refine default changed", int((cell != preds_geocell).sum()), "of", NP)
        np.savez(os.path.join(GOLD, "pipeline24_spread.npz"), embedding=emb.numpy(), center=center.numpy(), radius=np.array(radius),
                 meta=np.array([31, 24, NP, 9731, C, 4, 2, 3]),
                 preds_LLH=preds_LLH.numpy(), preds_geocell=preds_geocell.numpy(),
                 topk_values=tk_val.numpy(), topk_indices=tk_idx.numpy(),
                 top8_logits=top8.values.detach().numpy(), top8_cells=top8.indices.numpy(), logit_margin=margin.numpy(),
                 logit_sigma=sigma.numpy(), image_cos_sim=np.array([float(cs.mean()), float(cs.min()), float(cs.max())]),
                 panorama_cos_sim=np.array([float(csp.mean()), float(csp.min()), float(csp.max())]),
                 default_LLH=llh.numpy(), default_cell=cell.numpy(), default_params=np.array([5, 1.6, 1000], dtype=np.float64))

    if want("refiner_cache"):
        # The refiner cache exactly as the reference writes it: evaluation/evaluate.py:72-75 builds
        # ProtoRefiner(20, False, 10000, proto_path, dataset_path, temperature=1) and `torch.save(refiner, proto_model_path)`s the whole
        # module; a later run reads `torch.load(proto_model_path).protos` (:65-69).  56 cells x 2 prototypes (evaluate() asks the head for 50 candidates, so C >= 50); the GPU
        # box has no /root/reference, so the file itself is the fixture (pigeon_amd.proto_refiner.load_refiner_cache reads it
        # without the reference package).  The bank / training rows are regenerated from the seeds in the side-car .npz.
        import types
        C, ppc, bseed = 56, 2, 2
        bank_rc = synthetic.make_bank(C, ppc, seed=bseed, empty_frac=0.05, max_members=3)
        csv_rc, ds_rc = os.path.join(tmp, "protos_rc.csv"), os.path.join(tmp, "hf_train_rc")
        synthetic.write_bank_reference_files(bank_rc, csv_rc, ds_rc)
        geo_rc = os.path.join(tmp, "geocells_rc.csv")
        synthetic.write_geocell_csv(geo_rc, synthetic.make_geocells(C, seed=0))
        ns = reference_loader.load(geo_rc, csv_rc, ds_rc)
        ref = ns.ProtoRefiner(20, False, 10000, proto_path=csv_rc, dataset_path=ds_rc, temperature=1)
        # `datasets` pickles a memory-mapped table as the PATH of its Arrow cache file, so the reference's cache is only readable
        # on the machine that wrote it (fine for the reference: same box).  A committed fixture must travel: move every per-cell
        # dataset into memory (same rows, same features, same torch format) before pickling.
        import datasets as _ds
        ref.protos = [None if p is None else _ds.Dataset.from_dict(p.with_format(None)[:], features=p.features).with_format("torch")
                      for p in ref.protos]
        tr = ref.dataset["train"]                     # the refiner also carries the training DatasetDict it was built from
        ref.dataset = _ds.DatasetDict(train=_ds.Dataset.from_dict(tr.with_format(None)[:], features=tr.features).with_format(**tr.format))
        mods = {"models": types.ModuleType("models"), "models.proto_refiner": types.ModuleType("models.proto_refiner")}
        mods["models.proto_refiner"].ProtoRefiner = ns.ProtoRefiner
        saved = {k: sys.modules.get(k) for k in mods}
        sys.modules.update(mods)
        try:
            torch.save(ref, os.path.join(GOLD, "proto.refiner"))
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
        np.savez(os.path.join(GOLD, "refiner_cache.npz"), meta=np.array([C, ppc, bseed, 3]),
                 n_empty=np.array(int((np.diff(bank_rc.cell_off) == 0).sum())))
        print("refiner_cache: pickled reference refiner,", os.path.getsize(os.path.join(GOLD, "proto.refiner")), "bytes")

    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
