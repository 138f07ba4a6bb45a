"""SuperGuessr geocell classification model on the HIP kernels, behind the reference's call surface.

Mirrors reference models/super_guessr.py (same constructor signature, `load_geocells`, `load_state`,
`forward(pixel_values | embedding, ..., labels_clf, ...)` returning `ModelOutput` or the serving tuple).
Only the inference branch that evaluation/evaluate.py:42-44 constructs is accelerated and supported:
`hierarchical=False, multi_task=False, heading=False`.  The training-only options raise NotImplementedError
(SURVEY.md section 2 row 4 lists them as out of scope).
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch
from torch import nn, Tensor
from torch.nn.parameter import Parameter

from . import hip_ops
from .certainty import Certainty
from .geo_utils import haversine_matrix, smooth_labels
from .clip_embedder import HipCLIPVisionModel
from .config import CLIP_EMBED_DIM, GEOCELL_PATH, GEOCELL_PATH_YFCC
from .utils import ModelOutput, TopK, resolve_name


# candidates the head computes beyond `num_candidates`: never exposed, they tell the certainty pass how far the next cells are
EXTRA_CANDIDATES = 4


class SuperGuessr(nn.Module):
    def __init__(self, base_model: nn.Module, panorama: bool = False, hierarchical: bool = False,
                 should_smooth_labels: bool = False, multi_task: bool = False, heading: bool = False,
                 yfcc: bool = False, serving: bool = False, freeze_base: bool = False,
                 num_candidates: int = 5, embed_dim: int = CLIP_EMBED_DIM, **kwargs):
        """Same arguments as reference models/super_guessr.py:31-34.

        base_model: None (run on precomputed embeddings), a `HipCLIPVisionModel`, or any module exposing a
        transformers-CLIPVisionModel state dict (e.g. a HuggingFace CLIPVisionModel): its weights are packed
        into the HIP encoder on first use.
        One extra keyword, `geocell_path`, overrides config.GEOCELL_PATH(_YFCC) (the reference hard-wires the
        path through its config module, :87-88).

        The geocell argmax is the reference's for every sample called certain -- a z ~ 4 statistical statement, see pigeon_amd/certainty.py
        -- and for the re-encoded ones (round 5: on by default).  The reference's `torch.argmax(geocell_probs)`
        (:454) is fp32 end to end; this path's embeddings carry the rounding of 16-bit MFMA operands (2.7e-4 relative on default-init
        weights, 7e-4 on a high-gain tower), so a sample whose margins are smaller than what that error moves could come out with a
        runner-up.  Every forward therefore measures, per sample, the TOLERANCE of its top-1 against every other cell
        (pg_head_certainty; error model and calibration: pigeon_amd/certainty.py) and -- `exact_top1`, default True, env
        PIGEON_EXACT_TOP1=0 switches it off -- re-encodes the samples that are not certain FROM THEIR PIXELS in the encoder's exact mode
        (pg_vit_forward_precise: split-fp16 GEMM operands, fp32 attention; ~2e-6 relative) and recomputes their head outputs.
        After every forward (ModelOutput keeps its 12 fields):
          .last_tol      (B,) fp32  the tolerance (see certainty.py), compared with .certainty.threshold()
          .last_certain  (B,) bool  the top-1 is the reference's (after the re-encode: judged at the exact tier's floor)
          .last_margin   (B,) fp32  logit(top-1) - logit(top-2);  .last_bound (B,) fp32 the margin change the threshold stands for
          .last_reencoded (n,) int64  the samples the exact tier re-encoded (empty when it is off)
        With a ProtoRefiner, `pigeon_amd.evaluate.certain_forward` extends the same statement to the refined cell and point.
        Caller-supplied `embedding`s are judged against `embedding_rel_tol` (keyword; default `margin_rel_tol_exact`: they carry no
        16-bit error of this forward).
        Extra keywords: `debias` (default True / env PIGEON_DEBIAS: the calibrated systematic part of the 16-bit encoder's error is
        subtracted from every fast embedding -- pg_embedding_debias -- instead of only being accounted for in the certainty test),
        `exact_top1`, `margin_kappa` (z-score, default 3.6), `margin_rel_tol` (default 1e-3 = the contract's embedding
        tolerance until `calibrate_certainty` -- called explicitly, or by the first forward that sees >= 8 samples with pixels, or
        once 16 samples have come in through smaller batches -- replaces it by the measured error of THIS set of weights), `margin_rel_tol_exact` (5e-6).
        """
        super(SuperGuessr, self).__init__()
        geocell_path = kwargs.pop('geocell_path', None)
        exact_top1 = kwargs.pop('exact_top1', None)
        self.exact_top1 = (os.environ.get('PIGEON_EXACT_TOP1', '1') not in ('', '0')) if exact_top1 is None else bool(exact_top1)
        self.certainty = Certainty(kappa=float(kwargs.pop('margin_kappa', os.environ.get('PIGEON_MARGIN_KAPPA', 3.6))),
                                   rel_tol=float(kwargs.pop('margin_rel_tol', os.environ.get('PIGEON_MARGIN_REL_TOL', 1e-3))),
                                   rel_tol_exact=float(kwargs.pop('margin_rel_tol_exact', 5e-6)), debias=kwargs.pop('debias', None))
        self.margin_autocalibrate = bool(kwargs.pop('margin_autocalibrate', True))
        self.embedding_rel_tol = float(kwargs.pop('embedding_rel_tol', self.certainty.rel_tol_exact))
        self.last_margin = self.last_certain = self.last_tol = None
        self.last_state = None
        if len(kwargs) > 0:
            print(f'Not using keyword arguments: {list(kwargs.keys())}')
        if hierarchical or multi_task or heading:
            raise NotImplementedError('pigeon_amd.SuperGuessr implements the inference configuration of '
                                      'evaluation/evaluate.py:42-44 (hierarchical=False, multi_task=False, heading=False)')

        self.base_model = base_model
        self.panorama = panorama
        self.hidden_size = embed_dim
        self.serving = serving
        self.should_smooth_labels = should_smooth_labels
        self.multi_task = multi_task
        self.heading = heading
        self.yfcc = yfcc
        self.freeze_base = freeze_base
        self.hierarchical = hierarchical
        self.num_candidates = num_candidates

        self._set_hidden_size()
        if geocell_path is None:
            geocell_path = GEOCELL_PATH_YFCC if self.yfcc else GEOCELL_PATH
        self.lla_geocells = self.load_geocells(geocell_path)
        self.num_cells = self.lla_geocells.size(0)
        self.input_dim = self.hidden_size

        self.cell_layer = nn.Linear(self.input_dim, self.num_cells)
        self.softmax = nn.Softmax(dim=-1)
        self._freeze_params()
        self.loss_fnc = nn.CrossEntropyLoss()
        self._hip_base = None
        self._wnorm = {}                                     # exact? -> (key, device tensor): see wstats()
        self._cal_buffer = []                                # pixels of small first batches, until there are enough to calibrate on
        self._engines = {}                                   # refiner -> settle-before-return engine (see `engine`)
        self._last_exact = None
        self._rel_tol0 = self.certainty.rel_tol              # the constructor's uncalibrated tolerance: what a weight load goes back to
        if self.exact_top1 and isinstance(self.base_model, HipCLIPVisionModel):
            self.base_model.enable_precise(True)             # pack the split-weight copy with the first build, not inside a request
        print(f'Initialized SuperGuessr classification model with {self.num_cells} geocells.')

    # legacy names of the certainty parameters (round 4)
    @property
    def margin_kappa(self) -> float:
        return self.certainty.kappa

    @margin_kappa.setter
    def margin_kappa(self, v: float):
        self.certainty.kappa = float(v)

    @property
    def margin_rel_tol(self) -> float:
        return self.certainty.rel_tol

    @margin_rel_tol.setter
    def margin_rel_tol(self, v: float):
        self.certainty.rel_tol = float(v)

    @property
    def margin_rel_tol_exact(self) -> float:
        return self.certainty.rel_tol_exact

    def _set_hidden_size(self):
        if self.base_model is not None:
            self.hidden_size = self.base_model.config.hidden_size
            self.mode = 'transformer'

    def _freeze_params(self):
        if self.base_model is not None and self.freeze_base:
            for param in self.base_model.parameters():
                param.requires_grad = False

    def load_geocells(self, path: str) -> Tensor:
        """reference models/super_guessr.py:162-174: CSV columns lng,lat -> (C,2) float64 Parameter"""
        geo_df = pd.read_csv(path)
        lla_coords = torch.tensor(np.ascontiguousarray(geo_df[['lng', 'lat']].values))
        return nn.parameter.Parameter(data=lla_coords, requires_grad=False)

    def load_state(self, path: str):
        """reference models/super_guessr.py:222-238: name-wise copy of a saved state dict"""
        own_state = self.state_dict()
        state_dict = torch.load(path, map_location=torch.device('cuda') if torch.cuda.is_available() else 'cpu')
        matched = 0
        for name, param in state_dict.items():
            name = resolve_name(name, own_state)         # transformers 4.23.1 `base_model.vision_model.*` -> `base_model.*`
            if name not in own_state:
                print(f'Parameter {name} not in model\'s state.')
                continue
            if isinstance(param, Parameter):
                param = param.data
            own_state[name].copy_(param)
            matched += 1
        if len(state_dict) > 0 and matched == 0:
            raise KeyError(f'load_state: none of the {len(state_dict)} parameters in {path} matched the model')
        self._hip_base = None
        if isinstance(self.base_model, HipCLIPVisionModel):
            self.base_model._weights_changed()
        # other weights: the measured error is void -- back to the constructor's tolerance (not a hard-coded one), nothing half-collected
        self.certainty = Certainty(self.certainty.kappa, self._rel_tol0, self.certainty.rel_tol_exact, debias=self.certainty.debias)
        self._cal_buffer = []
        self._engines = {}

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        sd.pop('base_model._dummy', None)
        if isinstance(self.base_model, HipCLIPVisionModel):
            sd.update(self.base_model.state_dict(prefix='base_model.'))
        return sd

    # ------------------------------------------------------------------------------------------ hot path
    def _encoder(self) -> HipCLIPVisionModel:
        if isinstance(self.base_model, HipCLIPVisionModel):
            return self.base_model
        if self._hip_base is None:                       # e.g. a transformers CLIPVisionModel: pack its weights once
            self._hip_base = HipCLIPVisionModel(self.base_model.state_dict())
            if self.exact_top1:
                self._hip_base.enable_precise(True)
            self._hip_base.to(self.cell_layer.weight.device)
        return self._hip_base

    def _assert_requirements(self, pixel_values=None, embedding=None, heading=None):
        if self.base_model is not None:
            assert pixel_values is not None, 'Parameter "pixel_values" must be supplied if model has a base model.'
        else:
            assert embedding is not None, 'Parameter "embedding" must be supplied if model does not have a base model.'

    def _to_one_hot(self, tensor: Tensor) -> Tensor:
        if tensor.dim() == 0:
            one_hot = torch.zeros(self.num_cells, device=tensor.device)
            one_hot[tensor.item()] = 1
            return one_hot
        return tensor

    def wstats(self, exact: bool = False) -> Tensor:
        """(2,) fp32 on the head's device: [largest row norm of cell_layer.weight, max over cells of |W[c] . drift|] -- what bounds
        |W[a] - W[c]| and (W[a] - W[c]).drift for the cells the certainty pass does not visit one by one (pg_head_wstats).  `exact`: the
        exact tier has no systematic part (second entry 0).  Recomputed when the weight tensor (in-place edits bump its version) or the
        calibrated drift changes."""
        W = self.cell_layer.weight
        drift = None if exact else self.certainty.drift_on(W.device)
        key = (W.data_ptr(), W._version, str(W.device), None if drift is None else (drift.data_ptr(), drift._version))
        if not isinstance(self._wnorm, dict):
            self._wnorm = {}
        hit = self._wnorm.get(exact)
        if hit is None or hit[0] != key:
            Wd = W.data if W.dtype == torch.float32 else W.data.float()
            self._wnorm[exact] = (key, hip_ops.head_wstats(Wd.contiguous(), drift))
        return self._wnorm[exact][1]

    def _panels(self) -> int:
        return 4 if self.panorama else 1

    def _head_rows(self, layer_input: Tensor) -> Tensor:
        if self.panorama:
            head_in = layer_input if layer_input.dim() == 3 else layer_input[:, None, :]   # mean over panels :437
        elif layer_input.dim() == 3 and layer_input.size(1) == 4:
            head_in = layer_input[:, 0].contiguous()                            # :440-441
        else:
            head_in = layer_input
        return head_in.contiguous()

    def _head_with_tol(self, head_in: Tensor, exact: bool) -> dict:
        """cell_layer + softmax + top-(k + extra) + argmax + centroid gather (:447-459), and the tolerance of the top-1."""
        W = self.cell_layer.weight.data
        kx = min(self.num_cells, self.num_candidates + EXTRA_CANDIDATES)
        o = hip_ops.head_forward(head_in, W, self.cell_layer.bias.data, self.lla_geocells.data, kx)
        drift = None if exact else self.certainty.drift_on(W.device)
        o['tol'], o['code'], o['margin'], o['sens'] = hip_ops.head_certainty(o['logits'], head_in, W, o['topk_indices'], drift,
                                                                             self.wstats(exact))
        return o

    @torch.no_grad()
    def encode_head(self, pixel_values: Tensor = None, embedding: Tensor = None) -> dict:
        """Fast pass: ViT + token mean (:395-398), head, tolerance of the top-1.  No re-encode.  Returns the step's STATE: a dict
        with `embedding` (as ModelOutput carries it), `head_in`, the head outputs over k + extra candidates (`topk_values`,
        `topk_indices`, `logits`, `preds_geocell`, `preds_LLH`), `tol`, `certain`, `exact` (rows on the exact tier: none yet) and
        `pixel_values` (the reshaped device pixels, or None).  `package` turns it into the reference's outputs; pigeon_amd.deferred.DeferredExact
        settles the rows that are not certain."""
        dev = self.cell_layer.weight.device
        px = None
        if self.panorama and pixel_values is not None:                          # :386-388
            num_samples = pixel_values.size(0)
            pixel_values = pixel_values.reshape((num_samples * 4, 3, 336, 336))
        if self.base_model is not None and pixel_values is not None:
            if pixel_values.dim() > 4:
                pixel_values = pixel_values.squeeze(1)                          # :392-393
            px = pixel_values.to(dev)
            if self.exact_top1 and self.margin_autocalibrate and not self.certainty.calibrated:
                self._autocalibrate(px)
            # a tower on which the calibration measured the 16-bit path OUTSIDE the embedding contract: every sample through the exact encoder
            exact_tier = bool(self.certainty.force_exact and self.exact_top1)
            embedding = self._encoder().embed_precise(px) if exact_tier else self._encoder().embed(px)   # :395-398 (ViT + token mean)
            bias = None if exact_tier else self.certainty.bias_on(dev)
            if bias is not None:
                # the 16-bit encoder's measured systematic error, taken out of every image's embedding (pigeon_amd/certainty.py `debias`):
                # what the head, the refiner and the caller see is the corrected embedding; the certainty kernels get no drift
                hip_ops.embedding_debias(embedding, bias)
            if self.panorama:
                embedding = embedding.reshape((num_samples, 4, embedding.shape[-1]))   # :404-405 (explicit width: B = 0 stays legal)
        else:
            embedding = embedding.to(dev, torch.float32).contiguous()
            exact_tier = False
        head_in = self._head_rows(embedding)
        # Which error the tolerances are compared with: the calibrated fast-path error (pixels through the 16-bit encoder), the exact
        # tier's floor (pixels through the exact encoder), or -- caller-supplied embeddings, which carry no error of THIS forward --
        # `embedding_rel_tol` (default: the exact tier's floor; whoever feeds embeddings written by `run.py embed`'s 16-bit path
        # passes that path's error instead).  No systematic part in the last two cases.
        at_floor = exact_tier or px is None
        st = self._head_with_tol(head_in, exact=at_floor)
        st['embedding'], st['head_in'], st['pixel_values'], st['exact_tier'] = embedding, head_in, px, exact_tier
        st['thr'] = (self.certainty.kappa * self.embedding_rel_tol) if px is None else self.certainty.threshold(exact=exact_tier)
        st['drift'] = None if at_floor else self.certainty.drift_on(dev)
        st['wstats'] = self.wstats(at_floor)
        return st

    @torch.no_grad()
    def exact_rows(self, pixel_rows) -> dict:
        """The exact tier for queued rows (pigeon_amd.deferred): `pixel_rows` is a list of (n_i, P*3*336*336) pixel-row tensors (the two
        segments of a circular queue); they go through pg_vit_forward_precise, the head and the tolerance pass at the exact tier's
        floor.  Returns the head state of the n = sum n_i rows (`embedding` as ModelOutput carries it)."""
        P = self._panels()
        n = sum(int(t.shape[0]) for t in pixel_rows)
        dev = self.cell_layer.weight.device
        emb = torch.empty((n * P, CLIP_EMBED_DIM), dtype=torch.float32, device=dev)
        enc, off = self._encoder(), 0
        for t in pixel_rows:
            m = int(t.shape[0])
            if m:
                enc.embed_precise(t.reshape((m * P, 3, 336, 336)), out=emb[off * P:(off + m) * P])
            off += m
        if self.panorama:
            emb = emb.reshape((n, P, CLIP_EMBED_DIM))
        o = self._head_with_tol(self._head_rows(emb), exact=True)
        o['embedding'] = emb
        return o

    def _publish(self, st: dict) -> None:
        if 'certain' not in st:                          # a state straight from `encode_head` (no engine behind it)
            st['certain'] = st['tol'] > st['thr']
            st['exact'] = torch.ones_like(st['certain']) if st.get('exact_tier', False) else torch.zeros_like(st['certain'])
        self.last_tol, self.last_margin, self.last_certain = st['tol'], st['margin'], st['certain']
        self._last_exact = st['exact']
        self.last_state = st

    @property
    def last_reencoded(self):
        """(n,) int64: the samples of the last forward that went through the exact tier (a host synchronisation when read)."""
        ex = self._last_exact
        return None if ex is None else torch.nonzero(ex).flatten()

    @property
    def last_bound(self):
        """(B,) fp32: the margin change the threshold stands for, per sample of the last forward (its tier's threshold x `sens`)."""
        st = self.last_state
        if st is None:
            return None
        thr = torch.where(st['exact'], self.certainty.threshold(True), self.certainty.threshold(False))
        return st['sens'] * thr

    def package(self, st: dict, labels: Tensor = None, labels_clf: Tensor = None):
        """State -> the reference's outputs (:459-483).  The state's reference to the input pixels is dropped here (it was only
        needed for a possible re-encode): `last_state` must not keep a whole batch of pixels alive."""
        self._publish(st)
        st['pixel_values'] = None
        k = self.num_candidates
        geocell_topk = TopK(st['topk_values'][:, :k], st['topk_indices'][:, :k])
        if not self.training and self.serving:                              # :462-466
            return st['preds_LLH'], geocell_topk, st['embedding']
        dev = st['logits'].device
        loss_clf = None
        if labels_clf is not None:                                          # :456, :474 (logged only)
            label_probs = self._to_one_hot(labels_clf.to(dev))
            if self.should_smooth_labels and labels is not None:            # :469-471 soft labels by distance
                distances = haversine_matrix(labels.to(dev), self.lla_geocells.data.t())
                label_probs = smooth_labels(distances)
            loss_clf = self.loss_fnc(st['logits'], label_probs)
        return ModelOutput(loss_clf, loss_clf, 0, 0, 0, st['preds_LLH'], st['preds_geocell'], None, None, None,
                           geocell_topk, st['embedding'])

    def forward(self, pixel_values: Tensor = None, embedding: Tensor = None, heading: Tensor = None,
                labels: Tensor = None, labels_clf: Tensor = None, labels_multi_task: Tensor = None,
                labels_climate: Tensor = None, labels_month: Tensor = None, index: Tensor = None):
        """Inference branch of reference models/super_guessr.py:350-483.

        pixel_values (B,12,336,336) [panorama] or (B,3,336,336); or embedding (B,4,1024)/(B,1024).
        Returns ModelOutput (or, with serving=True in eval mode, the (pred_LLH, topk, embedding) tuple, :462-466).
        """
        self._assert_requirements(pixel_values, embedding, heading)
        if not self.cell_layer.weight.is_cuda:
            raise RuntimeError('pigeon_amd.SuperGuessr runs on the GPU only: call .to("cuda") first (no CPU fallback)')
        with torch.no_grad():
            # fast pass, tolerance of the top-1 against every cell, exact re-encode of the samples that are not certain
            # (pigeon_amd.deferred, settled before this call returns: one host synchronisation)
            res = self.engine(None).submit(pixel_values, embedding)[0]
            return self.package(dict(res['state']), labels, labels_clf)

    def engine(self, refiner=None, **kw):
        """The settle-before-return form of pigeon_amd.deferred.DeferredExact for (this model, `refiner`), built once."""
        from .deferred import DeferredExact
        key = id(refiner)
        hit = self._engines.get(key)
        if hit is None or hit[0] is not refiner:
            self._engines[key] = hit = (refiner, DeferredExact(self, refiner, immediate=True, **kw))
        return hit[1]

    @torch.no_grad()
    def _autocalibrate(self, px: Tensor) -> None:
        """First use without an explicit `calibrate_certainty`: a batch of >= 8 samples calibrates right away; smaller batches (a
        server answering one panorama at a time) are collected until 16 samples have been seen, then calibrate once.  Frozen after."""
        P = self._panels()
        if px.shape[0] == 0:
            return
        if px.shape[0] >= 8 * P and not self._cal_buffer:
            self._calibrate(px)
            return
        self._cal_buffer.append(px.detach().clone())
        if sum(t.shape[0] for t in self._cal_buffer) >= 16 * P:
            buf, self._cal_buffer = torch.cat([t.to(px.dtype) for t in self._cal_buffer]), []
            self._calibrate(buf)

    @torch.no_grad()
    def _calibrate(self, px: Tensor, max_samples: int = 32) -> dict:
        P = self._panels()
        n = min(max_samples, px.shape[0] // P)
        px = px[:n * P]
        enc = self._encoder()
        fast_i, exact_i = enc.embed(px), enc.embed_precise(px)
        st = self.certainty.calibrate(fast_i.reshape((-1, P, CLIP_EMBED_DIM)).mean(dim=1), exact_i.reshape((-1, P, CLIP_EMBED_DIM)).mean(dim=1),
                                      fast_images=fast_i, exact_images=exact_i)
        if self.certainty.force_exact:
            print(f'pigeon_amd.SuperGuessr: the 16-bit encoder measures {st["image_rel_err"]:.2e} (worst image '
                  f'{st["worst_image_rel_err"]:.2e}) against the exact encoder on these weights -- outside the 1e-3 embedding '
                  f'contract; every sample will be encoded in the exact mode (about 4x the time per image).')
        return st

    @torch.no_grad()
    def calibrate_certainty(self, pixel_values: Tensor, max_samples: int = 32) -> float:
        """Measure what the 16-bit path's embedding error IS on this model -- once per set of weights -- and set the certainty
        threshold from it: up to `max_samples` samples go through the fast and the exact encoder (`Certainty.calibrate`: the
        systematic part of their difference and the RMS of the rest).  Frozen afterwards: which samples a later forward re-encodes
        does not depend on what earlier batches held.  Returns the total RMS relative difference.  Needs pixels and a base model."""
        if self.base_model is None or pixel_values is None:
            raise ValueError('calibrate_certainty needs pixel_values and a base model')
        dev = self.cell_layer.weight.device
        px = pixel_values.reshape((-1, 3, 336, 336)).to(dev)
        return self._calibrate(px, max_samples)['fast_vs_exact_rms']

    def __str__(self):
        shown = (('base_model', self.base_model is not None), ('panorama', self.panorama), ('hierarchical', self.hierarchical),
                 ('multi-task', self.multi_task), ('yfcc', self.yfcc), ('embedding_size', self.hidden_size), ('input_dim', self.input_dim),
                 ('num_geocells', self.num_cells), ('label_smoothing', self.should_smooth_labels), ('uses_headings', self.heading),
                 ('freeze_base', self.freeze_base), ('serving', self.serving))
        width = max(len(k) for k, _ in shown) + 1
        return 'SuperGuessr(\n' + ''.join(f'\t{k.ljust(width)}= {v}\n' for k, v in shown) + ')'
