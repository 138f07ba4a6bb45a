"""pigeon_amd/certainty.py on the CPU: the calibration arithmetic (systematic part fitted on the even samples, residual measured on the odd
ones, the floor, the guard that sends a whole tower to the exact encoder) and what the tolerance formula of csrc/certainty.hip MEANS
under the error model it assumes -- written out here in numpy, no GPU, no library.

The error model (certainty.py header): e_fast - e_ref = |e| (beta + r), beta one vector per set of weights, r of relative RMS norm eps
and unknown direction.  A decision with margin m and gradient g has tolerance t = (m - |e| g.beta) / (|e| |g| / 32) and is called
certain when t > kappa eps.
"""
import math

import numpy as np
import pytest
import torch

from pigeon_amd.certainty import Certainty

D = 1024


def _pairs(n, beta_norm, eps, seed, scale_spread=True):
    """n 'exact' embeddings and their 'fast' versions under the error model."""
    g = torch.Generator().manual_seed(seed)
    exact = torch.randn((n, D), generator=g) * (torch.rand((n, 1), generator=g) * 3 + 0.5 if scale_spread else 1.0)
    beta = torch.randn((D,), generator=g)
    beta = beta / beta.norm() * beta_norm
    r = torch.randn((n, D), generator=g) * (eps / math.sqrt(D))
    fast = exact + exact.norm(dim=1, keepdim=True) * (beta + r)
    return fast, exact, beta


def test_calibrate_recovers_systematic_part_and_residual():
    fast, exact, beta = _pairs(64, beta_norm=2.6e-4, eps=5e-5, seed=1)
    c = Certainty()
    st = c.calibrate(fast, exact)
    assert c.calibrated and st["samples"] == 64 and st["drift_used"]
    assert abs(st["fast_vs_exact_rms"] - math.hypot(2.6e-4, 5e-5)) < 0.03 * 2.6e-4
    assert abs(st["drift_norm"] - 2.6e-4) < 0.02 * 2.6e-4
    # out of sample: the residual of the odd samples against the mean of the 32 even ones = eps * sqrt(1 + 1/32)
    assert abs(st["residual_rms"] - 5e-5 * math.sqrt(1 + 1 / 32)) < 0.05 * 5e-5
    assert c.rel_tol == pytest.approx(1.1 * st["residual_rms"])
    assert float((c.drift - beta).norm()) < 0.2 * 5e-5 * 2                       # the mean over 64 samples: eps / 8 off
    assert c.threshold() == pytest.approx(3.6 * c.rel_tol) and c.threshold(exact=True) == pytest.approx(3.6 * 5e-6)
    assert not c.force_exact and "calibrated on 64 samples" in c.describe()


def test_calibrate_debias_takes_the_systematic_part_out_of_the_images():
    """`debias` (the default): with the per-image embeddings at hand the systematic part becomes a per-image BIAS that is subtracted
    from the embeddings (pg_embedding_debias; `apply_bias` is its arithmetic) -- the certainty kernels then get no drift, and rel_tol is
    the held-out residual of the corrected panel means."""
    n, P = 64, 4
    fi, ei, beta = _pairs(n * P, beta_norm=2.6e-4, eps=5e-5, seed=11)
    common = torch.randn((1, D), generator=torch.Generator().manual_seed(12)) * 6.0       # images of one tower share most of their embedding
    ei = ei + common
    fi = ei + ei.norm(dim=1, keepdim=True) * ((fi - (ei - common)) / (ei - common).norm(dim=1, keepdim=True))   # same beta + r, relative to the new |e|
    fast, exact = fi.reshape((n, P, D)).mean(dim=1), ei.reshape((n, P, D)).mean(dim=1)
    c = Certainty(debias=True)
    st = c.calibrate(fast, exact, fast_images=fi, exact_images=ei)
    assert st["debias"] and st["drift_used"] and c.drift is None and c.drift_on(torch.device("cpu")) is None
    assert c.bias is not None and float((c.bias - beta).norm()) < 0.1 * 5e-5 and c.bias_on(torch.device("cpu")) is c.bias
    assert abs(st["drift_norm"] - 2.6e-4) < 0.02 * 2.6e-4
    # four images with independent rests: the panel mean's residual is about half an image's (a little more: the norms differ)
    assert 0.4 * 5e-5 < st["residual_rms"] < 0.75 * 5e-5 and c.rel_tol == pytest.approx(1.1 * st["residual_rms"])
    fixed = Certainty.apply_bias(fi, c.bias)
    err = ((fixed - ei).norm(dim=1) / ei.norm(dim=1))
    assert float(err.pow(2).mean().sqrt()) < 1.1 * 5e-5 and st["image_rel_err_debiased"] < 0.25 * st["image_rel_err"]
    assert "subtracted from every fast embedding" in c.describe() and "margin / (|e|" in c.describe()
    # the switch: the former behaviour (a drift vector for the kernels, nothing subtracted)
    c0 = Certainty(debias=False)
    st0 = c0.calibrate(fast, exact, fast_images=fi, exact_images=ei)
    assert not st0["debias"] and st0["drift_used"] and c0.bias is None and c0.drift is not None
    # no systematic part: nothing is subtracted
    fi2, ei2, _ = _pairs(n * P, beta_norm=0.0, eps=3e-4, seed=13)
    c2 = Certainty(debias=True)
    st2 = c2.calibrate(fi2.reshape((n, P, D)).mean(dim=1), ei2.reshape((n, P, D)).mean(dim=1), fast_images=fi2, exact_images=ei2)
    assert not st2["debias"] and c2.bias is None and c2.drift is None and c2.rel_tol == pytest.approx(1.1 * st2["fast_vs_exact_rms"])


def test_calibrate_without_a_systematic_part_keeps_none():
    fast, exact, _ = _pairs(64, beta_norm=0.0, eps=3e-4, seed=2)
    c = Certainty()
    st = c.calibrate(fast, exact)
    # fitting a mean to pure noise does not shrink the held-out residual below 0.9 x the total: no drift vector is kept
    assert not st["drift_used"] and c.drift is None
    assert c.rel_tol == pytest.approx(1.1 * st["fast_vs_exact_rms"]) and abs(st["fast_vs_exact_rms"] - 3e-4) < 0.03 * 3e-4


def test_calibrate_small_sample_floor_and_switches():
    fast, exact, _ = _pairs(6, beta_norm=2e-4, eps=5e-5, seed=3)
    c = Certainty()
    st = c.calibrate(fast, exact)                                              # fewer than 8 samples: no split, no drift
    assert not st["drift_used"] and c.drift is None and c.rel_tol == pytest.approx(1.1 * st["fast_vs_exact_rms"])
    fast, exact, _ = _pairs(32, beta_norm=2e-4, eps=5e-5, seed=3)
    st = Certainty().calibrate(fast, exact, use_drift=False)
    assert not st["drift_used"] and st["residual_rms"] == st["fast_vs_exact_rms"]
    # the floor: an error below the exact tier's own (rel_tol_exact) is not believed
    fast, exact, _ = _pairs(32, beta_norm=0.0, eps=1e-6, seed=4)
    c = Certainty()
    c.calibrate(fast, exact)
    assert c.rel_tol == pytest.approx(2.0 * c.rel_tol_exact)
    with pytest.raises(ValueError):
        Certainty().calibrate(torch.zeros((0, D)), torch.zeros((0, D)))
    # uncalibrated: the contract's tolerance
    c = Certainty()
    assert not c.calibrated and c.threshold() == pytest.approx(3.6e-3) and c.drift_on(torch.device("cpu")) is None
    assert "uncalibrated" in c.describe()


def test_calibrate_guard_sends_a_tower_outside_the_contract_to_the_exact_encoder():
    fast, exact, _ = _pairs(32, beta_norm=3e-4, eps=5e-5, seed=5)
    g = torch.Generator().manual_seed(6)
    ei = torch.randn((128, D), generator=g)

    def images(rel, worst=None):
        d = torch.randn((128, D), generator=g)
        d = d / d.norm(dim=1, keepdim=True) * ei.norm(dim=1, keepdim=True) * rel
        if worst is not None:
            d[7] = d[7] / d[7].norm() * ei[7].norm() * worst
        return ei + d

    c = Certainty()
    st = c.calibrate(fast, exact, fast_images=images(7e-4), exact_images=ei)
    assert not c.force_exact and abs(st["image_rel_err"] - 7e-4) < 2e-5 and abs(st["worst_image_rel_err"] - 7e-4) < 2e-5
    c = Certainty()
    c.calibrate(fast, exact, fast_images=images(8.7e-4), exact_images=ei)        # whole sample above 0.85 x the contract
    assert c.force_exact and "OUTSIDE the embedding contract" in c.describe()
    c = Certainty()
    st = c.calibrate(fast, exact, fast_images=images(5e-4, worst=9.7e-4), exact_images=ei)   # one image above 0.95 x
    assert c.force_exact and st["image_rel_err"] < 6e-4 and st["worst_image_rel_err"] > 9.5e-4
    c = Certainty()
    c.calibrate(fast, exact, fast_images=images(5e-4, worst=9.7e-4), exact_images=ei, contract=2e-3)
    assert not c.force_exact


def _tolerance(m, g, beta, e):
    """csrc/certainty.hip tol_of, in float64: the largest eps, in standard deviations, the decision survives."""
    en = np.linalg.norm(e, axis=-1)
    return (m - en * (g @ beta)) / (en * np.linalg.norm(g, axis=-1) / 32.0)


def test_tolerance_is_a_z_score_under_the_error_model():
    """Margins move by |e| g.beta + N(0, (eps |e| |g| / 32)^2): (observed change - systematic part) / (eps |e| |g| / 32) must be a unit
    Gaussian, and a decision flips exactly when that draw exceeds its tolerance / eps.  With kappa = 3.6 the flips among the decisions
    called certain are those of a 3.6-sigma tail (1.6e-4 of the decisions NEAR the band; none in a sample of this size), while a band
    without the systematic part misses flips."""
    rng = np.random.default_rng(0)
    n, eps, kappa = 20000, 5e-5, 3.6
    beta = rng.standard_normal(D)
    beta *= 2.6e-4 / np.linalg.norm(beta)
    e = rng.standard_normal((n, D)) * rng.uniform(0.5, 3.0, (n, 1))
    g = rng.standard_normal((n, D)) * rng.uniform(0.02, 0.2, (n, 1))             # W[c0] - W[cj]
    en, gn = np.linalg.norm(e, axis=1), np.linalg.norm(g, axis=1)
    sigma = eps * en * gn / 32.0
    m = np.abs(rng.standard_normal(n)) * 6 * sigma + en * np.abs(g @ beta) * rng.uniform(0, 2, n)   # margins near both terms
    r = rng.standard_normal((n, D)) * (eps / 32.0)                                # relative RMS norm eps
    assert abs(np.sqrt((r ** 2).sum(1).mean()) - eps) < 0.01 * eps
    # m: the margin this path measures (on e_fast); the reference's: m(e_ref) = m - g.(e_fast - e_ref)
    m_ref = m - en * np.einsum("nd,nd->n", g, beta[None] + r)
    z = (m - m_ref - en * (g @ beta)) / sigma
    assert abs(z.std() - 1.0) < 0.02 and abs(z.mean()) < 0.03
    t = _tolerance(m, g, beta, e)
    flips = m_ref < 0
    np.testing.assert_array_equal(flips, z > t / eps)
    certain = t > kappa * eps
    assert flips.sum() > 200 and certain.sum() > 2000                            # the sample really straddles the band
    assert not (flips & certain).any()
    # the same band WITHOUT the systematic part calls decisions certain that flip
    t0 = m / sigma * eps
    assert (flips & (t0 > kappa * eps)).sum() > 0
    # ... and the band is not much wider than it has to be: a fair share of what it flags does flip
    assert flips.sum() > 0.1 * (~certain).sum()


def test_tolerance_worst_direction_is_exact():
    """An error of relative norm rho in the WORST direction (along g) moves the margin by rho |e| |g|: the decision holds for
    rho < (m - |e| g.beta) / (|e| |g|) = t / 32 and flips just above -- the statement tests/test_gpu_certainty.py checks on the kernels."""
    rng = np.random.default_rng(1)
    e, g, beta = rng.standard_normal(D) * 2, rng.standard_normal(D) * 0.1, rng.standard_normal(D) * 1e-5
    m = 0.05
    t = float(_tolerance(m, g, beta, e))
    en = np.linalg.norm(e)
    for rho, holds in ((0.99 * t / 32, True), (1.01 * t / 32, False)):
        delta = en * (beta + rho * g / np.linalg.norm(g))                        # e_fast - e_ref; the reference's margin is m - g.delta
        assert ((m - g @ delta) > 0) == holds


# ---- pigeon_amd.deferred / evaluate.certain_forward: the host logic around the kernels, with scripted stand-ins (tests/_scripted.py;
# CPU tensors, oracle/requeue_oracle.py as the device ops; nothing of the real arithmetic runs) -------------------------------------
def test_certain_forward_control_flow():
    from _scripted import FAST_THR, ScriptedModel, ScriptedRefiner, make_pixels
    from pigeon_amd.evaluate import certain_forward
    #            sample:   0 certain   1 head     2 refiner     3 both (the head is named)   4 refiner
    px = make_pixels(head_fast=[1, 0, 1, 0, 1], head_exact=[1, 1, 1, 0, 1], ref_fast=[1, 1, 0.25, 0, 0], ref_exact=[1, 1, 0, 1, 1])
    m, r = ScriptedModel(), ScriptedRefiner()
    st, info = certain_forward(m, r, pixel_values=px)
    assert info["cause"].tolist() == [0, 1, 3000, 1, 3000]
    assert m.calls == [("exact", 4)] and info["reencoded"].tolist() == [1, 2, 3, 4]
    assert r.calls == [("fast", 5), ("exact", 4)]                                 # the systematic part only on the fast pass
    # after the exact pass: 1 and 4 resolved; 2 still below the exact floor at the refiner; 3 still uncertain at the head
    assert info["certain"].tolist() == [True, True, False, False, True] and torch.equal(m.last_certain, info["certain"])
    assert info["refine_code"].tolist() == [0, 0, 3000, 0, 0] and info["boundary_checked"] is True
    assert info["head_tol"].tolist() == pytest.approx([1.0, 1e-3, 1e-3, 0.0, 1e-3]) and float(info["refine_tol"][2]) == 0.0
    # row 0 keeps the fast pass's values, rows 1..4 carry the exact tier's -- embedding, head outputs AND refinement (run once each)
    rows = px.reshape(5, -1)
    fast, exact = m._embed(rows, True), m._embed(rows, False)
    assert torch.equal(st["embedding"][0], fast[0]) and torch.equal(st["embedding"][1:], exact[1:])
    hx = m._head(exact)
    assert torch.equal(st["preds_geocell"][1:], hx["preds_geocell"][1:]) and torch.equal(st["topk_indices"][1:], hx["topk_indices"][1:])
    assert torch.equal(st["logits"][1:], hx["logits"][1:]) and torch.equal(st["logits"][0], m._head(fast)["logits"][0])
    assert torch.equal(info["refined_LLH"][0], (m._head(fast)["preds_LLH"][0] + 1).float())
    assert torch.equal(info["refined_LLH"][1:], (hx["preds_LLH"][1:] + 2).float())
    assert torch.equal(info["refined_geocell"][1:], hx["topk_indices"][1:, 1])

    # the 16-bit path alone (exact_top1=False): nothing is re-encoded, the flags are reported as they are
    m, r = ScriptedModel(exact_top1=False), ScriptedRefiner()
    st, info = certain_forward(m, r, pixel_values=px)
    assert m.calls == [] and r.calls == [("fast", 5)] and info["reencoded"].numel() == 0
    assert info["certain"].tolist() == [True, False, False, False, False] and info["cause"].tolist() == [0, 1, 3000, 1, 3000]

    # embeddings handed in (no pixels to re-encode): flags only
    m = ScriptedModel()
    emb = torch.zeros((5, 4, 8))
    emb[:, 0, 0] = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0])
    st, info = certain_forward(m, None, embedding=emb)
    assert m.calls == [] and info["certain"].tolist() == [True, False, True, False, True] and info["refine_tol"] is None
    assert info["cause"].tolist() == [0, 1, 0, 1, 0]

    # no refiner, pixels: only the head's flags, one exact pass
    m = ScriptedModel()
    st, info = certain_forward(m, None, pixel_values=px)
    assert m.calls == [("exact", 2)] and info["reencoded"].tolist() == [1, 3] and info["certain"].tolist() == [True, True, True, False, True]

    # everything certain: no exact pass at all
    m, r = ScriptedModel(), ScriptedRefiner()
    st, info = certain_forward(m, r, pixel_values=make_pixels([1] * 5))
    assert m.calls == [] and r.calls == [("fast", 5)] and bool(info["certain"].all()) and not bool(info["cause"].any())
    assert FAST_THR == 0.5


def _run_engine(steps, **kw):
    """The scripted steps through one engine -> {step: emitted result}, the order of emission, the engine."""
    from _scripted import ScriptedModel, ScriptedRefiner
    from oracle import requeue_oracle
    from pigeon_amd.deferred import DeferredExact
    m, r = ScriptedModel(), ScriptedRefiner()
    eng = DeferredExact(m, r, ops=requeue_oracle, **kw)
    out, order, lag = {}, [], []
    for i, px in enumerate(steps):
        for res in eng.submit(px, meta=i):
            out[res["meta"]] = res
            order.append(res["meta"])
            lag.append(i - res["meta"])
    for res in eng.flush():
        out[res["meta"]] = res
        order.append(res["meta"])
    return out, order, lag, eng, m


def test_deferred_engine_equals_settling_every_step():
    """The deferred form (queue on the device, the count one step late, one exact pass per >= min_flush queued rows) hands out, for
    every step, exactly what settling every step before it returns does -- every tensor, bit for bit -- in order, each step once."""
    from _scripted import make_pixels
    g = torch.Generator().manual_seed(7)
    steps = []
    for i in range(14):
        B = 5 if i != 9 else 3                                                    # one short batch in the middle (a ragged loader)
        flags = lambda p: (torch.rand(B, generator=g) > p).float().tolist()        # noqa: E731
        steps.append(make_pixels(flags(0.25), flags(0.1), flags(0.2), flags(0.1), seed=100 + i))
    steps[4] = make_pixels([1] * 5, seed=104)                                      # a step with nothing to fix
    now, order_now, _, eng_now, m_now = _run_engine(steps, immediate=True)
    later, order, lag, eng, m = _run_engine(steps, min_flush=4, max_lag=5)
    assert order_now == list(range(14)) and order == list(range(14))               # in order, each step once
    assert max(lag) <= 5 + 1 and max(lag) >= 1                                      # handed out late, never later than max_lag (+ the count's step)
    for i in range(14):
        a, b = now[i], later[i]
        for k in a:
            if torch.is_tensor(a[k]):
                assert torch.equal(a[k], b[k]), (i, k)
        for k in a["state"]:
            if torch.is_tensor(a["state"][k]):
                assert torch.equal(a["state"][k], b["state"][k]), (i, "state", k)
        assert a["queued"] == b["queued"]
    # fewer, larger exact passes: every pass but the last (flush) / a max_lag one runs on >= min_flush rows
    sizes = [n for _, n in m.calls]
    assert sum(sizes) == sum(n for _, n in m_now.calls) == sum(r["queued"][0] for r in later.values())
    assert len(sizes) < len(m_now.calls) and all(n >= 4 for n in sizes[:-1])
    assert [f["slots_run"] for f in eng.flush_log] == sizes and eng.check_nothing_dropped() == 0
    # something was actually re-encoded and something was not
    assert any(bool(r["exact"].any()) for r in later.values()) and any(not bool(r["exact"].all()) for r in later.values())


def test_deferred_engine_pass_quantum_takes_whole_rounds_and_leaves_the_rest_queued():
    """With a pass quantum a min_flush pass takes a whole number of quanta from the HEAD of the queue (the size at which the exact
    encoder's row panels fill whole rounds of the CUs: pigeon_amd.deferred.round_quantum) and leaves the rest for the next pass;
    what is handed out is still, bit for bit, what settling every step gives -- in order, each step once, nothing dropped."""
    from _scripted import make_pixels
    g = torch.Generator().manual_seed(11)
    steps = []
    for i in range(18):
        flags = lambda p: (torch.rand(6, generator=g) > p).float().tolist()        # noqa: E731
        steps.append(make_pixels(flags(0.3), flags(0.15), flags(0.2), flags(0.1), seed=300 + i))
    now, order_now, _, _, m_now = _run_engine(steps, immediate=True)
    later, order, lag, eng, m = _run_engine(steps, min_flush=3, max_lag=8, pass_quantum=3)
    assert order == order_now == list(range(18)) and eng.check_nothing_dropped() == 0
    for i in range(18):
        for k in now[i]:
            if torch.is_tensor(now[i][k]):
                assert torch.equal(now[i][k], later[i][k]), (i, k)
        for k in now[i]["state"]:
            if torch.is_tensor(now[i]["state"][k]):
                assert torch.equal(now[i]["state"][k], later[i]["state"][k]), (i, "state", k)
    sizes = [n for _, n in m.calls]
    assert sum(sizes) == sum(n for _, n in m_now.calls) and eng.cap == 3 + 3 + 2 * 6        # min_flush + one quantum + two unseen steps
    # every pass but the closing flush (and a max_lag one, which takes everything) is a whole number of quanta; some pass left rows behind
    assert all(n % 3 == 0 for n in sizes[:-1]) and len(sizes) >= 4
    assert max(lag) <= 8 + 1
    # default (no quantum off the GPU): min_flush falls back to 10 and a pass takes all that is queued
    _, _, _, eng0, _ = _run_engine(steps[:3])
    assert eng0.pass_quantum == 0 and eng0.min_flush == 10
    # min_flush below the quantum: a pass that cannot fill one quantum takes what there is
    later2, order2, _, eng2, m2 = _run_engine(steps, min_flush=2, max_lag=8, pass_quantum=5)
    assert order2 == list(range(18)) and sum(n for _, n in m2.calls) == sum(sizes)
    for i in range(18):
        assert torch.equal(now[i]["embedding"], later2[i]["embedding"]) and torch.equal(now[i]["refined_geocell"], later2[i]["refined_geocell"])


def test_pass_size_is_a_pure_function_of_the_gathered_queue_lengths():
    """DeferredExact._pass_size: one rank -- whole quanta once the queue holds min_flush; several ranks -- the LOWER MEDIAN queue
    triggers (all ranks run the same number of slots: triggering on the longest of 8 queues runs the others a third empty,
    tools/pass_policy_sim.py), the longest queue triggers one quantum later and then takes all but about one quantum of it; without a
    quantum: everything, once the longest queue holds min_flush."""
    from pigeon_amd.deferred import DeferredExact
    eng = DeferredExact.__new__(DeferredExact)
    eng.pass_quantum, eng.min_flush = 7, 7
    assert [eng._pass_size([n]) for n in (0, 6, 7, 13, 14, 15, 30)] == [0, 0, 7, 7, 14, 14, 28]
    assert eng._pass_size([5, 6, 7, 4, 8, 7, 6, 9]) == 0                     # sorted 4 5 6 [6] 7 7 8 9: the lower median holds 6
    assert eng._pass_size([5, 7, 7, 4, 8, 7, 6, 9]) == 7                     # ... 7
    assert eng._pass_size([3, 4, 5, 2, 13, 4, 5, 6]) == 0                    # the longest is not a quantum ahead yet
    assert eng._pass_size([3, 4, 5, 2, 14, 4, 5, 6]) == 7 and eng._pass_size([0, 0, 0, 30]) == 21
    assert eng._pass_size([14, 15]) == 14 and eng._pass_size([6, 13]) == 0 and eng._pass_size([6, 14]) == 7
    eng.pass_quantum, eng.min_flush = 0, 10                                   # no quantum (off the GPU): the first form of the round
    assert eng._pass_size([9, 3]) == 0 and eng._pass_size([12, 3]) == 12
    eng.pass_quantum, eng.min_flush = 5, 2                                    # min_flush below the quantum: what there is
    assert eng._pass_size([3]) == 3 and eng._pass_size([1]) == 0


def test_deferred_engine_max_lag_and_queue_wrap():
    """A single uncertain row does not wait forever (max_lag), and the circular queue wraps without losing or mixing rows."""
    from _scripted import make_pixels
    steps = [make_pixels([0, 1, 1], seed=1)] + [make_pixels([1, 1, 1], seed=2 + i) for i in range(6)]
    out, order, lag, eng, m = _run_engine(steps, min_flush=50, max_lag=3)
    assert order == list(range(7)) and m.calls == [("exact", 1)] and eng.flush_log[0]["at_step"] <= 4
    assert out[0]["exact"].tolist() == [True, False, False] and max(lag) <= 4
    # wrap: capacity = min_flush + 2 B = 2 + 6 = 8 slots, 40 rows go through it
    steps = [make_pixels([0, 0, 1], seed=10 + i) for i in range(20)]
    now, _, _, _, _ = _run_engine(steps, immediate=True)
    later, order, _, eng, m = _run_engine(steps, min_flush=2, max_lag=6)
    assert eng.cap == 8 and sum(n for _, n in m.calls) == 40 and eng.check_nothing_dropped() == 0
    for i in range(20):
        assert torch.equal(now[i]["embedding"], later[i]["embedding"]) and torch.equal(now[i]["refined_LLH"], later[i]["refined_LLH"])


def test_deferred_engine_empty_batch_keeps_its_place():
    """An empty batch (a rank whose shard is empty must still be able to step) queues nothing and is handed out in its turn."""
    from _scripted import make_pixels
    steps = [make_pixels([0, 1, 1], seed=1), make_pixels([], seed=2), make_pixels([1, 0, 1], seed=3)]
    out, order, lag, eng, m = _run_engine(steps, min_flush=50, max_lag=5)
    assert order == [0, 1, 2] and out[1]["embedding"].shape == (0, 4, 8) and out[1]["certain"].numel() == 0
    assert out[0]["exact"].tolist() == [True, False, False] and out[2]["exact"].tolist() == [False, True, False] and m.calls == [("exact", 2)]
    out, order, _, _, _ = _run_engine([make_pixels([], seed=2)], immediate=True)
    assert order == [0] and out[0]["refined_LLH"].shape == (0, 2)


def test_deferred_engine_pixel_dtype_change_settles_and_restarts_the_queue():
    """fp32 pixel tensors, then fp16 ones (what the GPU preprocessing hands over): the rows queued in the old geometry are settled
    first, the queue restarts, every step is handed out once and in order (bench.py's ingest leg did exactly this and failed)."""
    from _scripted import ScriptedModel, ScriptedRefiner, make_pixels
    from oracle import requeue_oracle
    from pigeon_amd.deferred import DeferredExact
    m, r = ScriptedModel(), ScriptedRefiner()
    eng = DeferredExact(m, r, ops=requeue_oracle, min_flush=50, max_lag=6)
    steps = [make_pixels([0, 1, 1], seed=1), make_pixels([1, 0, 1], seed=2), make_pixels([1, 1, 0], seed=3).half(), make_pixels([0, 1, 1], seed=4).half()]
    order = []
    for i, px in enumerate(steps):
        order += [res["meta"] for res in eng.submit(px, meta=i)]
    assert order[:2] == [0, 1] and m.calls[0] == ("exact", 2)              # the dtype change settled the two fp32 rows at once
    order += [res["meta"] for res in eng.flush()]
    assert order == [0, 1, 2, 3] and m.calls == [("exact", 2), ("exact", 2)] and eng.q_pixels.dtype == torch.float16
    assert eng.check_nothing_dropped() == 0
