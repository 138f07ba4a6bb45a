"""The error model behind "is this discrete output certain to be the reference's?" (round 5; kernels: csrc/certainty.hip).

The reference is fp32 end to end (models/super_guessr.py:447-459, models/proto_refiner.py:154-222).  This path's embedding of a
sample differs from the reference's by

    e_fast - e_ref  =  |e| * (beta + r)

`beta` is the SYSTEMATIC part -- the same vector for every image of a given set of weights, relative to |e| (the 16-bit rounding of the
weights is identical for every token of every image and survives the 577-token mean) -- and `r` the rest: relative RMS norm
`rel_tol`, direction unknown.  Both are MEASURED once per set of weights by sending a few samples through the fast and the exact
encoder (`SuperGuessr.calibrate_certainty`).  A decision with margin m and gradient g then survives when

    (m - |e| g.beta) / (|e| |g| / 32)  >  kappa * rel_tol

(left side: what pg_head_certainty / pg_refine_certainty return per sample, the minimum over the sample's decisions; kappa: a z-score).

What happens to `beta` (round 6, second session).  Until then it only entered that test -- and a sample whose margin was smaller than
the systematic shift |e| g.beta was sent through the exact encoder although the shift is KNOWN: on the bench's tower beta is 5.4 x the
rest (2.57e-4 against 4.7e-5), and HALF of the samples the fast mode flagged sat there (profiles/r06/certainty_audit_ref_32768_final.txt:
430 of 818 flagged with a tolerance below half a residual, where a uniform margin density puts 60).  With `debias` (default; env
PIGEON_DEBIAS=0 / `Certainty(debias=False)` restores the former behaviour) the measured systematic part is taken OUT of every fast
embedding, per image, before anything downstream reads it:

    e_img  <-  e_img - |e_img| * bias                   (pg_embedding_debias; bias = mean over calibration IMAGES of (fast - exact) / |exact|)

so the embeddings this path returns are the fast encoder's minus its own measured bias -- closer to the fp32 reference (on the bench's
tower 5e-5 instead of 2.7e-4 per image) -- the head and the refiner decide on the corrected embedding, and the certainty kernels see no
systematic part any more (beta = 0): a decision survives when m / (|e| |g| / 32) > kappa * rel_tol, rel_tol being the held-out residual
of the CORRECTED panel means.  Nothing else changes: which samples are certain is still a statement about `r` alone, the exact tier's
results carry no correction, and the audits against the reference module are what backs it (profiles/r06/certainty_audit_ref_debias_*.txt).
Data-parallel jobs: the vector enters outputs, so replicas that should return the same embedding for the same image must measure the
same vector -- call `calibrate_certainty` with one batch every rank shares (bench.py does); left to the first-batch auto-calibration
each rank measures its own (they differ by the calibration noise, ~residual / sqrt(images) = a few 1e-6 relative).
The exact tier's own floor is `rel_tol_exact`: 5e-6 = 3 x the exact encoder's measured error against the real reference (1.6e-6 at 24
layers, 1.7e-6 on the stress towers; the reference's CPU result itself moves by ~1e-6 with the thread partition).  Until round 6 it was
2e-5, which left a third of the re-encoded samples `uncertain` although nothing more exact exists to send them to.

What this buys is a STATISTICAL statement, not a bound: with kappa = 3.6 on 1.1 x the measured residual (z ~ 4) a sample called
certain has its discrete outputs equal to the fp32 reference's with probability 1 - O(1e-5) per decision under the Gaussian error
model -- which the audits support (profiles/r05/certainty_audit_*.txt, profiles/r06/: no wrong output among tens of thousands of
samples called certain; the measured tail of the margin changes is no heavier than a unit Gaussian's) and which nothing proves.
Which samples are re-encoded depends on the calibration samples (the first batch unless `calibrate_certainty` is called with a fixed
set), so the returned embeddings of borderline samples can differ at the 3e-4 level between runs that calibrate on different data;
their discrete outputs do not.
"""
from __future__ import annotations

import os
from typing import Optional

import torch


class Certainty:
    def __init__(self, kappa: float = 3.6, rel_tol: float = 1e-3, rel_tol_exact: float = 5e-6, debias: Optional[bool] = None):
        self.debias = (os.environ.get('PIGEON_DEBIAS', '1') not in ('', '0')) if debias is None else bool(debias)
        self.bias: Optional[torch.Tensor] = None      # (1024,) fp32 per-IMAGE relative bias that pg_embedding_debias subtracts, or None
        self.kappa = float(kappa)
        self.rel_tol = float(rel_tol)                 # uncalibrated default: the contract's embedding tolerance
        self.rel_tol_exact = float(rel_tol_exact)
        self.drift: Optional[torch.Tensor] = None     # (1024,) fp32 on the model's device, or None
        self.calibrated = False
        self.stats = {}
        # the fast path measured OUTSIDE the embedding contract on this set of weights (see `calibrate`): every sample then goes
        # through the exact encoder
        self.force_exact = False

    def threshold(self, exact: bool = False) -> float:
        return self.kappa * (self.rel_tol_exact if exact else self.rel_tol)

    def drift_on(self, device) -> Optional[torch.Tensor]:
        """The systematic part the certainty kernels account for (None once it is subtracted from the embeddings instead: `bias_on`)."""
        if self.drift is None:
            return None
        if self.drift.device != device:
            self.drift = self.drift.to(device)
        return self.drift

    def bias_on(self, device) -> Optional[torch.Tensor]:
        """The per-image bias `pg_embedding_debias` takes out of every fast embedding (None: nothing to take out)."""
        if self.bias is None:
            return None
        if self.bias.device != device:
            self.bias = self.bias.to(device)
        return self.bias

    @staticmethod
    def apply_bias(images: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
        """What pg_embedding_debias computes, in torch (calibration arithmetic and tests): images (m,1024) -> images - |images| bias."""
        x = images.float()
        return x - x.norm(dim=1, keepdim=True) * bias.to(x.device)

    @torch.no_grad()
    def calibrate(self, fast: torch.Tensor, exact: torch.Tensor, safety: float = 1.1, use_drift: bool = True,
                  fast_images: Optional[torch.Tensor] = None, exact_images: Optional[torch.Tensor] = None,
                  contract: float = 1e-3) -> dict:
        """fast, exact: (n,1024) embeddings of the same n samples (panel means for panoramas).  Sets `rel_tol` (and `drift` when a
        systematic part explains a worthwhile share of the error), freezes them, returns the measured statistics.

        The systematic part is fitted on the even samples and the residual measured on the ODD ones (out of sample, so that
        `rel_tol` is not flattered by the fit); the vector finally kept is the mean over all samples.

        `fast_images` / `exact_images` (m,1024): the per-IMAGE embeddings behind them.  The contract is "embeddings within `contract`
        relative" per image; a tower on which the 16-bit path itself measures outside it (whole-sample error above 0.85 x, or the
        worst calibration image above 0.95 x the contract -- seen only on a synthetic extreme: every attention head at high q.k
        gain, tests/test_gpu_precise.py) sets `force_exact`: every sample is then encoded by the exact encoder."""
        self.force_exact = False                      # a verdict of THIS calibration only (an earlier one's must not linger)
        rel = (fast.float() - exact.float()) / exact.float().norm(dim=1, keepdim=True).clamp_min(1e-30)
        n = int(rel.shape[0])
        if n == 0:
            raise ValueError('calibrate: no samples')
        total = float(rel.norm(dim=1).pow(2).mean().sqrt())
        st = {'samples': n, 'fast_vs_exact_rms': total, 'drift_norm': 0.0, 'residual_rms': total, 'drift_used': False}
        drift = bias = None
        st['debias'] = False
        P = 0
        if fast_images is not None and exact_images is not None and n and fast_images.shape[0] % n == 0:
            P = int(fast_images.shape[0]) // n              # images per sample (4 panels of a panorama, or 1)
        if use_drift and n >= 8 and self.debias and P >= 1:
            # the systematic part is taken out of the EMBEDDINGS (per image); fitted on the images of the even samples, the residual is
            # what is left of the odd samples' panel means after the correction they would get at run time (fast norm, fitted bias)
            fi, ei = fast_images.float().reshape((n, P, -1)), exact_images.float().reshape((n, P, -1))
            rel_img = (fi - ei) / ei.norm(dim=2, keepdim=True).clamp_min(1e-30)
            b_half = rel_img[0::2].reshape((-1, rel_img.shape[-1])).mean(dim=0)
            held_f, held_e = fi[1::2], ei[1::2]
            corr = self.apply_bias(held_f.reshape((-1, held_f.shape[-1])), b_half).reshape(held_f.shape).mean(dim=1)
            ref = held_e.mean(dim=1)
            resid = float(((corr - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-30)).pow(2).mean().sqrt())
            st['residual_rms'] = resid
            st['drift_norm'] = float(rel_img.reshape((-1, rel_img.shape[-1])).mean(dim=0).norm())
            if resid < 0.9 * total:
                bias = rel_img.reshape((-1, rel_img.shape[-1])).mean(dim=0).contiguous()
                st['drift_used'] = st['debias'] = True
            else:
                st['residual_rms'] = total
        elif use_drift and n >= 8:
            fit, held = rel[0::2], rel[1::2]
            beta_half = fit.mean(dim=0)
            resid = float((held - beta_half).norm(dim=1).pow(2).mean().sqrt())
            st['residual_rms'] = resid
            st['drift_norm'] = float(rel.mean(dim=0).norm())
            if resid < 0.9 * total:
                drift = rel.mean(dim=0).contiguous()
                st['drift_used'] = True
        if fast_images is not None and exact_images is not None and fast_images.numel():
            fi, ei = fast_images.float(), exact_images.float()
            st['image_rel_err'] = float((fi - ei).norm() / ei.norm().clamp_min(1e-30))
            st['worst_image_rel_err'] = float(((fi - ei).norm(dim=1) / ei.norm(dim=1).clamp_min(1e-30)).max())
            # (the verdict is taken on the RAW 16-bit error also when a bias is subtracted afterwards: the conservative reading)
            self.force_exact = st['image_rel_err'] > 0.85 * contract or st['worst_image_rel_err'] > 0.95 * contract
            if bias is not None:
                ci = self.apply_bias(fi, bias)              # (in sample: the bias was fitted on these images)
                st['image_rel_err_debiased'] = float((ci - ei).norm() / ei.norm().clamp_min(1e-30))
                st['worst_image_rel_err_debiased'] = float(((ci - ei).norm(dim=1) / ei.norm(dim=1).clamp_min(1e-30)).max())
        st['force_exact'] = self.force_exact
        eps = st['residual_rms'] if (drift is not None or bias is not None) else total
        self.rel_tol = max(safety * eps, 2.0 * self.rel_tol_exact)
        self.drift = drift
        self.bias = bias
        self.calibrated = True
        st['rel_tol'] = self.rel_tol
        st['kappa'] = self.kappa
        self.stats = st
        return st

    def describe(self) -> str:
        s = self.stats
        how = (f"calibrated on {s.get('samples')} samples through the fast and the exact encoder: total relative error RMS "
               f"{s.get('fast_vs_exact_rms', 0):.3g}, systematic part |beta| {s.get('drift_norm', 0):.3g} "
               f"({'subtracted from every fast embedding' if s.get('debias') else 'used' if s.get('drift_used') else 'not used'}), "
               f"residual RMS (out of sample) {s.get('residual_rms', 0):.3g}"
               + (f"; per image: {s.get('image_rel_err', 0):.3g} overall, worst {s.get('worst_image_rel_err', 0):.3g}"
                  + (" -- OUTSIDE the embedding contract: every sample goes through the exact encoder" if self.force_exact else "")
                  if 'image_rel_err' in s else "")
               if self.calibrated else "uncalibrated: the contract's embedding tolerance")
        return (f"a sample is certain when every discrete decision downstream of its embedding (top-1 cell against every other cell; "
                f"with a refiner: winning candidate, candidate-set boundary, nearest prototype, farthest member) keeps "
                f"{'margin' if s.get('debias') else '(margin - |e| grad.beta)'} / (|e| |grad| / 32) > kappa x rel_tol = {self.kappa:g} x {self.rel_tol:.3g} "
                f"(pg_head_certainty / pg_refine_certainty); {how}")
