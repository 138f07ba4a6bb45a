"""The deferred exact tier on the GPU (round 6; run with -m gpu on an MI355X):

  * csrc/requeue.hip's kernels through the C ABI against oracle/requeue_oracle.py (a torch restatement of include/pigeon_hip.h's
    contract, written independently of the kernels): flags / causes / slots incl. NaN tolerances, queue wrap and overflow; row copies
    of every vector width; the remapped scatter; pg_head_wstats against torch;
  * pigeon_amd.deferred.DeferredExact on the real SuperGuessr / ProtoRefiner (2-layer tower): the deferred form hands out, for every
    step, bit for bit what settling every step before it returns does; re-encoded rows carry the exact encoder's chain, the others the
    fast path's; `evaluate_model` (which runs on it) equals the explicit per-batch chain.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def env():
    from oracle import requeue_oracle
    from pigeon_amd import _lib, hip_ops, synthetic
    _lib.require_gpu()
    return dict(ops=hip_ops, orc=requeue_oracle, syn=synthetic)


def test_requeue_append_against_oracle(env):
    ops, orc = env["ops"], env["orc"]
    g = torch.Generator().manual_seed(0)
    for B, cap, flushed0, app0 in ((1, 4, 0, 0), (5, 8, 0, 0), (300, 64, 37, 40), (700, 1000, 12345, 12400), (130, 16, 100, 110), (64, 0, 0, 0)):
        ht = torch.rand(B, generator=g)
        rt = torch.rand(B, generator=g)
        ht[::7] = float("nan")                                         # a NaN tolerance is never certain
        rt[3::11] = float("nan")
        rc = torch.randint(0, 5, (B,), generator=g, dtype=torch.int32) * 1000
        for use_r in (True, False):
            for force in (False, True):
                cnt_o = torch.tensor([app0, 0], dtype=torch.int64)
                sd_o = torch.full((max(cap, 1),), -7, dtype=torch.int64)
                c_o, cause_o, rs_o = orc.requeue_append(ht, rt if use_r else None, rc if use_r else None, 0.4, force, dst_base=1000,
                                                        flushed=flushed0, cap=cap, counters=cnt_o, slot_dst=sd_o)
                cnt = torch.tensor([app0, 0], dtype=torch.int64, device=DEV)
                sd = torch.full((max(cap, 1),), -7, dtype=torch.int64, device=DEV)
                c, cause, rs = ops.requeue_append(ht.to(DEV), rt.to(DEV) if use_r else None, rc.to(DEV) if use_r else None, 0.4, force,
                                                  dst_base=1000, flushed=flushed0, cap=cap, counters=cnt if cap else None,
                                                  slot_dst=sd[:cap] if cap else None)
                assert torch.equal(c.cpu(), c_o) and torch.equal(cause.cpu(), cause_o), (B, cap, use_r, force)
                if cap:
                    assert torch.equal(rs.cpu(), rs_o) and torch.equal(cnt.cpu(), cnt_o) and torch.equal(sd.cpu()[:cap], sd_o[:cap])
                    if B == 130:
                        assert int(cnt_o[1]) > 0 and bool((rs_o == -2).any())          # the overflow case really overflowed
                else:
                    assert rs is None
    # B = 0 is a no-op
    c, cause, rs = ops.requeue_append(torch.empty(0, device=DEV), None, None, 0.5)
    assert c.numel() == 0 and cause.numel() == 0


def test_row_copies_scatter_take_against_oracle(env):
    ops, orc = env["ops"], env["orc"]
    g = torch.Generator().manual_seed(1)
    B, cap = 37, 16
    slots = torch.full((B,), -1, dtype=torch.int32)
    pick = torch.randperm(B, generator=g)[:cap]
    slots[pick] = torch.randperm(cap, generator=g).to(torch.int32)
    slots[0] = -2
    for shape, dt in (((12 * 336 * 336,), torch.float32), ((12 * 336 * 336,), torch.float16), ((4, 1024), torch.float32), ((50,), torch.int64),
                      ((2,), torch.float64), ((3,), torch.float32), ((), torch.int64), ((5,), torch.uint8), ((), torch.bool)):
        src = (torch.rand((B,) + shape, generator=g) * 100).to(dt)
        dst_o = torch.zeros((cap,) + shape, dtype=dt)
        orc.rows_to_slots(src, slots, dst_o)
        dst = torch.zeros((cap,) + shape, dtype=dt, device=DEV)
        ops.rows_to_slots(src.to(DEV), slots.to(DEV), dst)
        assert torch.equal(dst.cpu(), dst_o), (shape, dt)
        # scatter: rows to arbitrary destinations, -1 skipped; plain and through the gathered-ring -> local remap
        n, rows = 29, 64
        s2 = (torch.rand((n,) + shape, generator=g) * 100).to(dt)
        d_row = torch.randperm(rows, generator=g)[:n].to(torch.int64)
        d_row[::5] = -1
        for remap in (None, (16, 4, 8), (8, 8, 0)):
            out_o = torch.zeros((rows,) + shape, dtype=dt)
            orc.scatter_rows(s2, d_row, out_o, remap)
            out = torch.zeros((rows,) + shape, dtype=dt, device=DEV)
            ops.scatter_rows(s2.to(DEV), d_row.to(DEV), out, remap)
            assert torch.equal(out.cpu(), out_o), (shape, dt, remap)
    sd = torch.arange(100, 100 + cap, dtype=torch.int64)
    for head, nv, npad in ((0, 3, 3), (14, 5, 8), (1000003, 16, 16), (5, 0, 4)):
        assert torch.equal(ops.requeue_take(sd.to(DEV), head, nv, npad).cpu(), orc.requeue_take(sd, head, nv, npad))


def test_head_wstats_against_torch(env):
    ops, orc = env["ops"], env["orc"]
    g = torch.Generator().manual_seed(2)
    for C in (1, 5, 1000, 10007):
        W = torch.randn((C, 1024), generator=g) * 0.3
        beta = torch.randn(1024, generator=g) * 1e-4
        for b in (beta, None):
            got = ops.head_wstats(W.to(DEV), None if b is None else b.to(DEV)).cpu()
            want = orc.head_wstats(W, b)
            assert torch.allclose(got, want, rtol=2e-5, atol=1e-9), (C, got, want)
    W[3, 7] = float("nan")
    assert bool(torch.isnan(ops.head_wstats(W.to(DEV), None)[0]))                 # a NaN weight poisons the bound


def _small_setup(env, tmp_path, C=60, kappa=3.6, rel_tol=None):
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.proto_refiner import ProtoRefiner
    from pigeon_amd.super_guessr import SuperGuessr
    syn = env["syn"]
    gp = os.path.join(str(tmp_path), "g.csv")
    syn.write_geocell_csv(gp, syn.make_geocells(C, seed=0))
    sd = syn.make_vit_weights(seed=11, layers=2, affine_jitter=True)
    vit = HipCLIPVisionModel(sd, layers=2).to(DEV)
    W, b = syn.make_head_weights(C, seed=1)
    kw = {} if rel_tol is None else {"margin_rel_tol": rel_tol}
    m = SuperGuessr(vit, panorama=True, freeze_base=True, num_candidates=5, geocell_path=gp, exact_top1=True, margin_kappa=kappa,
                    margin_autocalibrate=False, **kw)
    with torch.no_grad():
        m.cell_layer.weight.copy_(W * 64); m.cell_layer.bias.copy_(b)
    m = m.to(DEV).eval()
    hb = syn.make_bank(C, 9, seed=5, empty_frac=0.1, max_members=6)
    ref = ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, bank=hb, device=DEV).eval()
    return m, ref, vit


def test_deferred_engine_on_the_real_classes(env, tmp_path):
    """2-layer tower, 9 steps of 6 panoramas (one short), a threshold chosen so that roughly a third of the rows are uncertain: the
    deferred engine == settling every step, bit for bit; the exact rows == the exact encoder's chain; the rest == the fast path's."""
    from pigeon_amd.deferred import DeferredExact
    ops, syn = env["ops"], env["syn"]
    m, ref, vit = _small_setup(env, tmp_path, rel_tol=1e-3)
    enc = vit._encoder(torch.device(DEV))
    steps = [syn.make_pixels(4 * (6 if i != 5 else 4), seed=50 + i, panorama=True).to(DEV) for i in range(9)]
    # pick kappa so that ~1/3 of the rows fall below the threshold (tolerances are data: measure the smaller of the head's and the
    # refiner's on the first steps, with a threshold of 0 -- nothing is flagged, nothing re-encoded)
    from pigeon_amd.evaluate import certain_forward
    m.certainty.kappa = 0.0
    tols = []
    for px in steps[:3]:
        _, info = certain_forward(m, ref, pixel_values=px)
        tols.append(torch.minimum(info["head_tol"], info["refine_tol"]).clamp(max=1e9))
    m.certainty.kappa = float(torch.quantile(torch.cat(tols), 0.35)) / m.certainty.rel_tol
    assert m.certainty.kappa > 0

    def run(**kw):
        eng = DeferredExact(m, ref, **kw)
        got = {}
        for i, px in enumerate(steps):
            for r in eng.submit(px, meta=i):
                got[r["meta"]] = r
        for r in eng.flush():
            got[r["meta"]] = r
        return got, eng
    now, eng_now = run(immediate=True)
    later, eng = run(min_flush=5, max_lag=4)
    assert sorted(later) == list(range(9)) and eng.check_nothing_dropped() == 0

    def same(x, y, ex_rows, what):
        """fast rows bit for bit; rows from the exact tier to its own reproducibility: the K-part count of its GEMMs is chosen by batch
        shape (vit.hip precise_parts), so two passes of different size differ in fp32 summation order (~1e-7)."""
        if x.dtype.is_floating_point and x.dim() >= 1 and x.shape[0] == ex_rows.shape[0]:
            assert torch.equal(x[~ex_rows], y[~ex_rows]), what
            if bool(ex_rows.any()):
                xe, ye = x[ex_rows].double(), y[ex_rows].double()
                fin = torch.isfinite(xe) & torch.isfinite(ye)
                assert torch.equal(torch.isfinite(xe), torch.isfinite(ye)), what
                assert float((xe[fin] - ye[fin]).abs().max()) <= 2e-5 * float(xe[fin].abs().max()) + 1e-30 or what[-1] in ("tol", "refine_tol", "margin", "sens"), what
        else:
            assert torch.equal(x, y), what
    n_exact = 0
    for i in range(9):
        a, b = now[i], later[i]
        ex = b["exact"]
        assert torch.equal(a["exact"], ex)
        for k in a:
            if torch.is_tensor(a[k]):
                same(a[k], b[k], ex, (i, k))
        for k in a["state"]:
            if torch.is_tensor(a["state"][k]):
                same(a["state"][k], b["state"][k], ex, (i, "state", k))
        px = steps[i]
        nb = px.shape[0]
        fast = enc.forward(px.reshape(-1, 3, 336, 336)).reshape(nb, 4, 1024)
        exact = enc.forward_precise(px.reshape(-1, 3, 336, 336)).reshape(nb, 4, 1024)
        n_exact += int(ex.sum())
        assert torch.equal(b["embedding"][~ex], fast[~ex])
        if bool(ex.any()):
            assert float((b["embedding"][ex] - exact[ex]).norm() / exact[ex].norm()) < 2e-6
        # head and refinement of what was handed out ARE the chain on the handed-out embeddings (one refinement, the right one)
        hx = ops.head_forward(b["embedding"].contiguous(), m.cell_layer.weight.data, m.cell_layer.bias.data, m.lla_geocells.data, 9)
        assert torch.equal(b["preds_geocell"], hx["preds_geocell"]) and torch.equal(b["topk_indices"], hx["topk_indices"][:, :5])
        assert torch.equal(b["state"]["logits"], hx["logits"])
        _, llh, cell = ref(b["embedding"], initial_preds=b["preds_LLH"], candidate_cells=b["topk_indices"], candidate_probs=b["topk_values"],
                           quiet=True)
        assert torch.equal(b["refined_LLH"], llh) and torch.equal(b["refined_geocell"], cell)
        assert int(ex.sum()) == b["queued"][0]
    assert 3 <= n_exact <= 45, n_exact                                   # the threshold really split the rows
    sizes = [f["slots_run"] for f in eng.flush_log]
    assert len(sizes) < len(eng_now.flush_log) and sum(sizes) == n_exact and all(s >= 5 for s in sizes[:-1])


def test_evaluate_model_runs_deferred_and_equals_the_chain(env, tmp_path):
    """evaluate_model (reference training/train_eval_loop.py:77-140) over pixels: its concatenated predictions equal the explicit
    per-batch chain certain_forward -> collect, it reports how many samples stay uncertain, and it took fewer exact passes than batches."""
    from pigeon_amd.evaluate import certain_forward, evaluate_model
    syn = env["syn"]
    m, ref, vit = _small_setup(env, tmp_path, rel_tol=1e-3)
    px_all = syn.make_pixels(4 * 22, seed=9, panorama=True)                     # 22 panoramas, batches of 4: 5 full + one of 2
    m.certainty.kappa = 0.0
    _, info = certain_forward(m, ref, pixel_values=px_all[:12].to(DEV))
    m.certainty.kappa = float(torch.quantile(torch.minimum(info["head_tol"], info["refine_tol"]).clamp(max=1e9), 0.4)) / m.certainty.rel_tol
    labels = torch.zeros((22, 2), dtype=torch.float64)
    clf = torch.arange(22) % 60

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 22

        def __getitem__(self, i):
            return {"pixel_values": px_all[i], "labels": labels[i], "labels_clf": clf[i]}

    ds = DS()
    res = evaluate_model(m, ds, None, None, ref, batch_size=4)
    want_p, want_c, want_cert = [], [], []
    for s in range(0, 22, 4):
        out, info = certain_forward(m, ref, pixel_values=px_all[s:s + 4].to(DEV))
        want_p.append(info["refined_LLH"].cpu().numpy()); want_c.append(out.preds_geocell.cpu().numpy()); want_cert.append(info["certain"].cpu().numpy())
    # (discrete outputs: equal although the exact passes have other sizes -- their embeddings differ by ~1e-7, see above)
    assert np.array_equal(res["preds"], np.concatenate(want_p)) and np.array_equal(res["preds_geocells"], np.concatenate(want_c))
    assert np.array_equal(res["geocell_certain"], np.concatenate(want_cert))
    assert res["uncertain_after_exact"] == int((~np.concatenate(want_cert)).sum())
    assert 1 <= len(res["exact_passes"]) < 6 and sum(f["slots_run"] for f in res["exact_passes"]) >= 3


def test_deferred_engine_random_stress(env, tmp_path):
    """40 steps of random size (1 .. 6 panoramas; the largest first so that the ring is sized once), a pixel dtype that flips between
    fp32 and fp16 twice, a small queue (min_flush 3: it wraps many times), about a third of the rows uncertain: every step is handed out
    once, in order; its discrete outputs, flags and re-encoded set equal those of settling every step at once; nothing is dropped."""
    from pigeon_amd.deferred import DeferredExact
    from pigeon_amd.evaluate import certain_forward
    syn = env["syn"]
    m, ref, vit = _small_setup(env, tmp_path, rel_tol=1e-3)
    g = torch.Generator().manual_seed(99)
    sizes = [6] + torch.randint(1, 7, (39,), generator=g).tolist()
    steps = []
    for i, b in enumerate(sizes):
        px = syn.make_pixels(4 * b, seed=700 + i, panorama=True).to(DEV)
        steps.append(px.half() if 12 <= i < 25 else px)
    m.certainty.kappa = 0.0
    tols = []
    for px in steps[:4]:
        _, info = certain_forward(m, ref, pixel_values=px)
        tols.append(torch.minimum(info["head_tol"], info["refine_tol"]).clamp(max=1e9))
    m.certainty.kappa = float(torch.quantile(torch.cat(tols), 0.33)) / m.certainty.rel_tol

    def run(**kw):
        eng = DeferredExact(m, ref, **kw)
        got, order = {}, []
        for i, px in enumerate(steps):
            for r in eng.submit(px, meta=i):
                got[r["meta"]] = r; order.append(r["meta"])
        for r in eng.flush():
            got[r["meta"]] = r; order.append(r["meta"])
        return got, order, eng
    now, order_now, _ = run(immediate=True)
    later, order, eng = run(min_flush=3, max_lag=5)
    assert order_now == list(range(40)) and order == list(range(40)) and eng.check_nothing_dropped() == 0
    n_exact = 0
    for i in range(40):
        a, b = now[i], later[i]
        for k in ("preds_geocell", "topk_indices", "refined_geocell", "refined_LLH", "preds_LLH", "exact", "certain", "cause", "index"):
            assert torch.equal(a[k], b[k]), (i, k)
        ex = b["exact"]
        assert torch.equal(a["embedding"][~ex], b["embedding"][~ex])
        if bool(ex.any()):
            assert float((a["embedding"][ex] - b["embedding"][ex]).norm() / a["embedding"][ex].norm()) < 2e-6
        n_exact += int(ex.sum())
    total = sum(sizes)
    assert 0.1 * total < n_exact < 0.7 * total, (n_exact, total)
    assert len(eng.flush_log) < sum(1 for i in range(40) if bool(now[i]["exact"].any()))       # fewer, larger exact passes
