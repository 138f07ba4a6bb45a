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
    assert c.threshold() == pytest.approx(3.6 * c.rel_tol) and c.threshold(exact=True) == pytest.approx(3.6 * 2e-5)
    assert not c.force_exact and "calibrated on 64 samples" in c.describe()


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


# ---- evaluate.certain_forward: the control flow around the kernels, with scripted stand-ins (CPU tensors; nothing computes) ---------
class _ScriptedModel:
    """SuperGuessr's surface as certain_forward uses it.  `head_ok` / `head_ok_exact`: which samples the head calls certain after the
    fast / the exact pass."""

    def __init__(self, head_ok, head_ok_exact, exact_top1=True):
        self.cell_layer = torch.nn.Linear(D, 7)
        self.certainty = Certainty()
        self.certainty.calibrate(*_pairs(16, 2e-4, 5e-5, seed=9)[:2])
        self.exact_top1 = exact_top1
        self.head_ok, self.head_ok_exact = torch.tensor(head_ok), torch.tensor(head_ok_exact)
        self.calls = []
        self.last_certain = None

    def wstats(self, exact=False):
        return torch.tensor([1.0, 0.0 if exact else 0.5])

    def encode_head(self, pixel_values=None, embedding=None):
        B = self.head_ok.numel()
        emb = torch.zeros((B, 4, D)) if embedding is None else embedding
        return dict(embedding=emb, pixel_values=pixel_values, topk_values=torch.rand((B, 9)), topk_indices=torch.zeros((B, 9), dtype=torch.int64),
                    preds_geocell=torch.zeros(B, dtype=torch.int64), preds_LLH=torch.zeros((B, 2), dtype=torch.float64),
                    tol=self.head_ok.float(), certain=self.head_ok.clone(), reencoded=torch.empty((0,), dtype=torch.int64))

    def reencode_rows(self, st, idx):
        self.calls.append(("reencode", idx.tolist()))
        if idx.numel():
            st["embedding"][idx] = 1.0                                            # "the exact encoder's embedding"
            st["certain"][idx] = self.head_ok_exact[idx]
            st["reencoded"] = idx

    def package(self, st, labels=None, labels_clf=None):
        return st


class _ScriptedRefiner:
    """forward_certain: tolerance / code per sample, scripted for the fast and for the exact pass (recognised by drift=None and the
    exact tier's wstats)."""

    def __init__(self, tol_fast, code_fast, tol_exact):
        self.tol_fast, self.code_fast, self.tol_exact = torch.tensor(tol_fast), torch.tensor(code_fast, dtype=torch.int32), torch.tensor(tol_exact)
        self.calls = []

    def forward_certain(self, emb, initial_preds, candidate_cells, candidate_probs, head_weight, wstats, drift=None):
        exact = float(wstats[1]) == 0.0
        self.calls.append(("exact" if exact else "fast", int(emb.shape[0]), drift is not None))
        B = emb.shape[0]
        if exact:
            rows = torch.nonzero(emb[:, 0, 0] == 1.0).flatten()
            assert rows.numel() == B                                              # only re-encoded samples are judged at the exact floor
            return None, None, self.tol_exact[:B].clone(), torch.full((B,), 1234, dtype=torch.int32), True
        return None, None, self.tol_fast.clone(), self.code_fast.clone(), True


def test_certain_forward_control_flow():
    from pigeon_amd.evaluate import certain_forward
    #          sample:   0 certain   1 head     2 refiner (nearest prototype)   3 both (the head is named)   4 refiner (underflow)
    head_ok = [True, False, True, False, True]
    m = _ScriptedModel(head_ok, head_ok_exact=[True, True, True, False, True])
    thr, thr_x = m.certainty.threshold(), m.certainty.threshold(exact=True)
    r = _ScriptedRefiner(tol_fast=[1.0, 1.0, 0.5 * thr, 0.0, 0.0], code_fast=[0, 0, 3000, 1002, -9],
                         tol_exact=[1.0, 0.5 * thr_x, 1.0, 1.0])                  # of the re-encoded [1, 2, 3, 4]: sample 2 stays uncertain
    st, info = certain_forward(m, r, pixel_values=torch.zeros((5, 12, 2, 2)))
    assert info["cause"].tolist() == [0, 1, 3000, 1, -9]
    assert m.calls == [("reencode", [1, 2, 3, 4])] and info["reencoded"].tolist() == [1, 2, 3, 4]
    assert r.calls == [("fast", 5, True), ("exact", 4, False)]                    # drift only on the fast pass
    # after the exact pass: 1 and 4 resolved; 2 still below the exact floor at the refiner; 3 still uncertain at the head
    assert info["certain"].tolist() == [True, True, False, False, True] and torch.equal(m.last_certain, info["certain"])
    assert info["refine_code"].tolist() == [0, 1234, 1234, 1234, 1234] and info["boundary_checked"] is True
    assert info["refine_tol"][2] == pytest.approx(0.5 * thr_x) and info["head_tol"].tolist() == [1.0, 0.0, 1.0, 0.0, 1.0]

    # the 16-bit path alone (exact_top1=False): nothing is re-encoded, the flags are reported as they are
    m = _ScriptedModel(head_ok, head_ok, exact_top1=False)
    r = _ScriptedRefiner([1.0, 1.0, 0.5 * thr, 0.0, 0.0], [0, 0, 3000, 1002, -9], [1.0] * 4)
    st, info = certain_forward(m, r, pixel_values=torch.zeros((5, 12, 2, 2)))
    assert m.calls == [] and r.calls == [("fast", 5, True)] and info["reencoded"].numel() == 0
    assert info["certain"].tolist() == [True, False, False, False, False] and info["cause"].tolist() == [0, 1, 3000, 1, -9]

    # embeddings handed in (no pixels to re-encode): the same
    m = _ScriptedModel(head_ok, head_ok)
    st, info = certain_forward(m, None, embedding=torch.zeros((5, 4, D)))
    assert m.calls == [] and info["certain"].tolist() == head_ok and info["refine_tol"] is None and info["cause"].tolist() == [0, 1, 0, 1, 0]

    # no refiner, pixels: only the head's flags, one re-encode
    m = _ScriptedModel(head_ok, head_ok_exact=[True, True, True, False, True])
    st, info = certain_forward(m, None, pixel_values=torch.zeros((5, 12, 2, 2)))
    assert m.calls == [("reencode", [1, 3])] and info["certain"].tolist() == [True, True, True, False, True]

    # everything certain: the re-encode is called with an empty set (the one host synchronisation still happens), no second refiner pass
    m = _ScriptedModel([True] * 5, [True] * 5)
    r = _ScriptedRefiner([1.0] * 5, [0] * 5, [1.0] * 4)
    st, info = certain_forward(m, r, pixel_values=torch.zeros((5, 12, 2, 2)))
    assert m.calls == [("reencode", [])] and r.calls == [("fast", 5, True)] and info["certain"].all() and not info["cause"].any()
