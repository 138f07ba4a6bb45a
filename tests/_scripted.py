"""Scripted stand-ins for SuperGuessr / ProtoRefiner as pigeon_amd.deferred.DeferredExact uses them (CPU tensors, nothing of the real
arithmetic), so that the HOST logic of the deferred exact tier -- queueing, the one-step-late count, the flush decision, patching,
the order in which steps are handed out, the data-parallel protocol -- can be tested without a GPU.  The device kernels' stand-in is
oracle/requeue_oracle.py (checked against the kernels themselves in tests/test_gpu_requeue.py).

A sample is a (12, 2, 2) "pixel" block whose first four values SCRIPT what the stand-ins report for it:
    px[0]: the head's tolerance after the fast pass        px[1]: ... after the exact pass
    px[2]: the refiner's tolerance after the fast pass     px[3]: ... after the exact pass
(compared with the thresholds FAST_THR / the certainty object's exact threshold), the rest is payload: the "embedding" is a fixed
random projection of the block -- plus 0.01 on the fast path, so that a re-encoded row is recognisable by its values.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import requeue_oracle  # noqa: E402
from pigeon_amd.certainty import Certainty  # noqa: E402

D = 8
FAST_THR = 0.5


def make_pixels(head_fast, head_exact=None, ref_fast=None, ref_exact=None, seed=0):
    """(B, 12, 2, 2) blocks with the scripted tolerances in place (defaults: certain everywhere)."""
    B = len(head_fast)
    g = torch.Generator().manual_seed(seed)
    px = torch.rand((B, 48), generator=g) + 2.0
    px[:, 0] = torch.tensor(head_fast, dtype=torch.float32)
    px[:, 1] = torch.tensor(head_exact if head_exact is not None else [1.0] * B, dtype=torch.float32)
    px[:, 2] = torch.tensor(ref_fast if ref_fast is not None else [1.0] * B, dtype=torch.float32)
    px[:, 3] = torch.tensor(ref_exact if ref_exact is not None else [1.0] * B, dtype=torch.float32)
    return px.reshape(B, 12, 2, 2)


class ScriptedModel:
    def __init__(self, cells=11, k=3, exact_top1=True, exact_cost_s=0.0):
        g = torch.Generator().manual_seed(123)
        self.Wp = torch.randn((48, D), generator=g)
        self.cell_layer = torch.nn.Linear(D, cells)
        with torch.no_grad():
            self.cell_layer.weight.copy_(torch.randn((cells, D), generator=g))
        self.cen = torch.rand((cells, 2), generator=g, dtype=torch.float64) * 100
        self.num_candidates, self.kx = k, min(cells, k + 4)
        self.certainty = Certainty()
        self.exact_top1 = exact_top1
        self.exact_cost_s = exact_cost_s
        self.calls = []                                   # ('exact', slots) per exact pass
        self.last_certain = None
        self._eng = {}

    def wstats(self, exact=False):
        return torch.tensor([1.0, 0.0 if exact else 0.5])

    def _head(self, emb):
        # row by row, in a Python loop: a row's head outputs must not depend on the batch it rides in (torch's CPU kernels pick
        # their vectorisation by shape), as they do not in the HIP kernels (one block per row)
        W = self.cell_layer.weight.data
        mean = emb.mean(dim=1)
        n = emb.shape[0]
        logits = torch.stack([(W * mean[i][None]).sum(dim=-1) for i in range(n)]) if n else mean.new_zeros((0, W.shape[0]))
        probs = torch.stack([torch.softmax(logits[i], dim=-1) for i in range(n)]) if n else logits
        top = torch.topk(probs, self.kx, dim=-1)
        cells = top.indices[:, 0].contiguous()
        n = emb.shape[0]
        return dict(embedding=emb, logits=logits, topk_values=top.values.contiguous(), topk_indices=top.indices.contiguous(),
                    preds_geocell=cells, preds_LLH=self.cen[cells], margin=torch.ones(n), sens=torch.ones(n))

    def _embed(self, rows, fast):
        if rows.shape[0] == 0:
            return torch.zeros((0, 4, D))
        emb = torch.stack([(r[:, None] * self.Wp).sum(dim=0) for r in rows])[:, None, :].repeat(1, 4, 1) + (0.01 if fast else 0.0)
        emb[:, :, -1] = rows[:, 2 if fast else 3, None]    # what the scripted refiner will report for the row
        return emb

    def encode_head(self, pixel_values=None, embedding=None):
        if pixel_values is not None:
            rows = pixel_values.reshape(pixel_values.shape[0], 48).float()
            st = self._head(self._embed(rows, fast=True))
            st['tol'] = rows[:, 0].clone()
            st.update(pixel_values=pixel_values, exact_tier=False, thr=FAST_THR, drift=torch.zeros(D), wstats=self.wstats(False))
        else:
            st = self._head(embedding.float())
            st['tol'] = embedding[:, 0, 0].float().clone()
            st.update(pixel_values=None, exact_tier=False, thr=FAST_THR, drift=None, wstats=self.wstats(True))
        return st

    def exact_rows(self, pixel_rows):
        rows = torch.cat([t.reshape(t.shape[0], -1) for t in pixel_rows]).float()
        self.calls.append(('exact', int(rows.shape[0])))
        if self.exact_cost_s:
            import time
            time.sleep(self.exact_cost_s * rows.shape[0])
        st = self._head(self._embed(rows, fast=False))
        st['tol'] = rows[:, 1].clone() * 1e-3              # 1.0 -> certain (> 1.8e-5), 0.0 -> still not
        return st

    def package(self, st, labels=None, labels_clf=None):
        self.last_certain = st['certain']
        return st

    def engine(self, refiner=None):
        from pigeon_amd.deferred import DeferredExact
        if id(refiner) not in self._eng:
            self._eng[id(refiner)] = DeferredExact(self, refiner, ops=requeue_oracle, immediate=True)
        return self._eng[id(refiner)]


class ScriptedRefiner:
    """forward_certain: refined point = initial + 1 (fast) / + 2 (exact: no drift handed in), refined cell = the second candidate,
    tolerance = the row's scripted value (x 1e-3 on the exact tier), code 3000 where it fails."""

    def __init__(self):
        self.calls = []

    def forward_certain(self, emb, initial_preds, candidate_cells, candidate_probs, head_weight, wstats, drift=None):
        exact = drift is None
        self.calls.append(('exact' if exact else 'fast', int(emb.shape[0])))
        tol = emb[:, 0, -1].clone() * (1e-3 if exact else 1.0)
        code = torch.where(tol > (1.8e-5 if exact else FAST_THR), 0, 3000).to(torch.int32)
        return ((initial_preds + (2.0 if exact else 1.0)).float(), candidate_cells[:, 1].contiguous(), tol, code, True)
