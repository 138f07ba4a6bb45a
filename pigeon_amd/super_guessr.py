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
from .geo_utils import haversine_matrix, smooth_labels
from .clip_embedder import HipCLIPVisionModel
from .config import CLIP_EMBED_DIM, GEOCELL_PATH, GEOCELL_PATH_YFCC
from .utils import ModelOutput, TopK, resolve_name


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

        Certainty of the top-1 (round 4).  The reference's `torch.argmax(geocell_probs)` (:454) is fp32 end to end; this path's
        embeddings carry the rounding of 16-bit MFMA operands (2.7e-4 relative on default-init weights, 6e-4 on the high-gain
        `pipeline24_spread` tower), so a panorama whose top-1 / top-2 logit margin is smaller than that error moves the logits may
        come out with the runner-up cell.  After every forward the model exposes, per sample (ModelOutput keeps its 12 fields):
          .last_margin   (B,) fp32  logit(top-1) - logit(top-2)
          .last_bound    (B,) fp32  margin_kappa * rel_tol * |emb| |W[top1] - W[top2]| / sqrt(1024): the margin change a relative
                                    embedding error of `rel_tol` in a random direction causes, times a safety factor
          .last_certain  (B,) bool  margin > bound: the reference's argmax is this cell
          .last_reencoded (n,) int64  the samples the exact mode re-encoded (empty when it is off)
        Extra keywords: `exact_top1` (default: env PIGEON_EXACT_TOP1=1) -- samples that are not certain are re-encoded FROM THE
        PIXELS in the encoder's exact mode (pg_vit_forward_precise: split-fp16 GEMM operands, fp32 attention; ~1e-6 relative, ~5x
        the time per image) and their head outputs recomputed, so that their top-1 is the fp32 one; `margin_rel_tol` (default
        1e-3 = the embedding tolerance of the contract, 1.5-4x the measured error; in exact mode it is re-calibrated on the fly to
        1.25 x the RMS difference between the fast and the exact embeddings of the re-encoded samples, `margin_autocalibrate`)
        and `margin_kappa` (default 4: a z-score, the margin change being Gaussian in units of rel_tol * sens).
        """
        super(SuperGuessr, self).__init__()
        geocell_path = kwargs.pop('geocell_path', None)
        exact_top1 = kwargs.pop('exact_top1', None)
        self.exact_top1 = (os.environ.get('PIGEON_EXACT_TOP1', '0') not in ('', '0')) if exact_top1 is None else bool(exact_top1)
        self.margin_rel_tol = float(kwargs.pop('margin_rel_tol', os.environ.get('PIGEON_MARGIN_REL_TOL', 1e-3)))
        self.margin_rel_tol_exact = float(kwargs.pop('margin_rel_tol_exact', 2e-5))
        self.margin_kappa = float(kwargs.pop('margin_kappa', os.environ.get('PIGEON_MARGIN_KAPPA', 4.0)))
        self.margin_autocalibrate = bool(kwargs.pop('margin_autocalibrate', True))
        self._cal_sumsq, self._cal_n = 0.0, 0
        self.last_margin = self.last_bound = self.last_certain = self.last_reencoded = None
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
        print(f'Initialized SuperGuessr classification model with {self.num_cells} geocells.')

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
        dev = self.cell_layer.weight.device
        with torch.no_grad():
            if self.panorama and pixel_values is not None:                      # :386-388
                num_samples = pixel_values.size(0)
                pixel_values = pixel_values.reshape((num_samples * 4, 3, 336, 336))
            if self.base_model is not None and pixel_values is not None:
                if pixel_values.dim() > 4:
                    pixel_values = pixel_values.squeeze(1)                      # :392-393
                embedding = self._encoder().embed(pixel_values.to(dev))         # :395-398 (ViT + token mean)
                if self.panorama:
                    embedding = embedding.reshape((num_samples, 4, embedding.shape[-1]))   # :404-405 (explicit width: B = 0 stays legal)
            else:
                embedding = embedding.to(dev, torch.float32).contiguous()

            layer_input = embedding
            if self.panorama:
                head_in = layer_input if layer_input.dim() == 3 else layer_input[:, None, :]   # mean over panels :437
            elif layer_input.dim() == 3 and layer_input.size(1) == 4:
                head_in = layer_input[:, 0].contiguous()                        # :440-441
            else:
                head_in = layer_input
            head_in = head_in.contiguous()
            o = hip_ops.head_forward(head_in, self.cell_layer.weight.data, self.cell_layer.bias.data,
                                     self.lla_geocells.data, self.num_candidates)          # :447-459
            embedding = self._certainty(o, head_in, embedding, pixel_values)
            logits = o['logits']
            geocell_preds = o['preds_geocell']
            pred_LLH = o['preds_LLH']
            geocell_topk = TopK(o['topk_values'], o['topk_indices'])

            if not self.training and self.serving:                              # :462-466
                return pred_LLH, geocell_topk, embedding

            loss_clf = None
            if labels_clf is not None:                                          # :456, :474 (logged only)
                label_probs = self._to_one_hot(labels_clf.to(dev))
                if self.should_smooth_labels and labels is not None:            # :469-471 soft labels by distance
                    distances = haversine_matrix(labels.to(dev), self.lla_geocells.data.t())
                    label_probs = smooth_labels(distances)
                loss_clf = self.loss_fnc(logits, label_probs)
            loss = loss_clf
            return ModelOutput(loss, loss_clf, 0, 0, 0, pred_LLH, geocell_preds, None, None, None,
                               geocell_topk, embedding)

    @torch.no_grad()
    def calibrate_certainty(self, pixel_values: Tensor, max_samples: int = 16) -> float:
        """Measure what the 16-bit path's embedding error IS on this model -- once per set of weights -- and set the certainty bound
        from it: up to `max_samples` samples go through the fast and the exact encoder, `margin_rel_tol` becomes 1.25 x the RMS
        relative difference of their (panel-mean) embeddings.  Returns that RMS.  Needs pixels and a base model; the encoder is
        (re)packed with the exact mode's split-weight copy if it did not have it."""
        if self.base_model is None or pixel_values is None:
            raise ValueError('calibrate_certainty needs pixel_values and a base model')
        dev = self.cell_layer.weight.device
        P = 4 if self.panorama else 1
        px = pixel_values[:max_samples].reshape((-1, 3, 336, 336)).to(dev)
        enc = self._encoder()
        fast = enc.embed(px).reshape((-1, P, CLIP_EMBED_DIM)).mean(dim=1)
        exact = enc.embed_precise(px).reshape((-1, P, CLIP_EMBED_DIM)).mean(dim=1)
        rel2 = ((fast - exact).norm(dim=1) / exact.norm(dim=1).clamp_min(1e-30)) ** 2
        self._cal_sumsq += float(rel2.sum())
        self._cal_n += int(rel2.numel())
        rms = (self._cal_sumsq / self._cal_n) ** 0.5
        self.margin_rel_tol = max(1.25 * rms, 4 * self.margin_rel_tol_exact)
        return rms

    def _certainty(self, o, head_in: Tensor, embedding: Tensor, pixel_values) -> Tensor:
        """Margin / bound / certain per sample; with exact_top1, re-encode the uncertain samples from their pixels in the
        encoder's exact mode and overwrite their rows of the head outputs `o` (and of the returned embedding)."""
        W = self.cell_layer.weight.data
        margin, sens, _ = hip_ops.head_margin(o['logits'], head_in, W)
        bound = sens * (self.margin_kappa * self.margin_rel_tol)
        certain = margin > bound
        self.last_reencoded = torch.empty((0,), dtype=torch.int64, device=margin.device)
        if self.exact_top1 and pixel_values is not None and self.base_model is not None and not bool(certain.all()):
            idx = torch.nonzero(~certain).flatten()
            P = 4 if self.panorama else 1
            px = pixel_values.reshape((-1, P, 3, 336, 336))[idx.to(pixel_values.device)].reshape((-1, 3, 336, 336))
            emb_x = self._encoder().embed_precise(px.to(W.device))
            emb_x = emb_x.reshape((idx.numel(), P, emb_x.shape[-1])) if self.panorama else emb_x
            # what the 16-bit path's embedding error IS on this model: the exact re-encode of the uncertain samples measures it
            # (panel-mean embeddings, relative L2).  1.25 x its running RMS replaces the conservative default of `margin_rel_tol`
            # once 8 samples have been seen -- the margin change it explains is Gaussian in those units (observed max over 128
            # panoramas: 2.8 x the RMS), so kappa is a z-score
            fast_mean = head_in[idx].reshape(idx.numel(), -1, head_in.shape[-1]).mean(dim=1) if head_in.dim() == 3 else head_in[idx]
            exact_mean = emb_x.mean(dim=1) if emb_x.dim() == 3 else emb_x
            rel2 = ((fast_mean - exact_mean).norm(dim=1) / exact_mean.norm(dim=1).clamp_min(1e-30)) ** 2
            self._cal_sumsq += float(rel2.sum())
            self._cal_n += int(rel2.numel())
            if self.margin_autocalibrate and self._cal_n >= 8:
                self.margin_rel_tol = max(1.25 * (self._cal_sumsq / self._cal_n) ** 0.5, 4 * self.margin_rel_tol_exact)
            embedding = embedding.clone()
            embedding[idx] = emb_x
            if self.panorama:
                hin = emb_x
            elif emb_x.dim() == 3 and emb_x.size(1) == 4:
                hin = emb_x[:, 0].contiguous()
            else:
                hin = emb_x
            o2 = hip_ops.head_forward(hin.contiguous(), W, self.cell_layer.bias.data, self.lla_geocells.data, self.num_candidates)
            for k in ('logits', 'topk_values', 'topk_indices', 'preds_geocell', 'preds_LLH'):
                o[k][idx] = o2[k]
            m2, s2, _ = hip_ops.head_margin(o2['logits'], hin.contiguous(), W)
            b2 = s2 * (self.margin_kappa * self.margin_rel_tol_exact)
            margin[idx], bound[idx], certain[idx] = m2, b2, m2 > b2
            self.last_reencoded = idx
        self.last_margin, self.last_bound, self.last_certain = margin, bound, certain
        return embedding

    def __str__(self):
        rep = 'SuperGuessr(\n'
        rep += f'\tbase_model\t= {self.base_model is not None}\n'
        rep += f'\tpanorama\t= {self.panorama}\n'
        rep += f'\thierarchical\t= {self.hierarchical}\n'
        rep += f'\tmulti-task\t= {self.multi_task}\n'
        rep += f'\tyfcc\t\t= {self.yfcc}\n'
        rep += f'\tembedding_size\t= {self.hidden_size}\n'
        rep += f'\tinput_dim\t= {self.input_dim}\n'
        rep += f'\tnum_geocells\t= {self.num_cells}\n'
        rep += f'\tlabel_smoothing\t= {self.should_smooth_labels}\n'
        rep += f'\tuses_headings\t= {self.heading}\n'
        rep += f'\tfreeze_base\t= {self.freeze_base}\n'
        rep += f'\tserving\t\t= {self.serving}\n'
        rep += ')'
        return rep
