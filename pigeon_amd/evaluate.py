"""Evaluation entry points behind the reference's names: `evaluate()` (evaluation/evaluate.py:10-85) builds
SuperGuessr + ProtoRefiner, `evaluate_model()` (training/train_eval_loop.py:35-161) runs the batch loop
`model(**data)` -> `refiner(...)` -> collect.  Plus `PanoramaPipeline`, the data-parallel step the benchmark
times: ViT + head on the local shard, ONE all-gather of embeddings / candidates, refinement.

Out of scope here (SURVEY.md section 2 rows 14,17): TensorBoard writers, the geopandas country metric; the
distance / geocell metrics of evaluation/metrics.py:89-181 are provided under the reference's keys.
"""
from __future__ import annotations

import logging
from typing import Callable, Dict, Optional

import numpy as np
import torch

from .config import DECAY_CONSTANT, EVAL_BATCH_SIZE_PER_GPU
from .distributed import Communicator
from .geo_utils import haversine_np
from .proto_refiner import ProtoRefiner, load_refiner_cache
from .super_guessr import SuperGuessr

logger = logging.getLogger('train')


def percentage_within_radius(distances: np.ndarray, km: float) -> float:
    """reference evaluation/metrics.py:89-100 (a FRACTION, strict `<`, despite the name)"""
    return (distances < km).sum() / len(distances)


def geoguessr_score(distances: np.ndarray) -> float:
    """reference evaluation/metrics.py:102-114; DECAY_CONSTANT = 1492.7 (config.py:52)"""
    return np.mean(np.round(5000 * np.exp(-distances / DECAY_CONSTANT)))


def topk_geocell_accuracy(cell_labels: np.ndarray, topk_preds: np.ndarray) -> float:
    """reference evaluation/metrics.py:116-136"""
    num_correct = 0
    for label, top5 in zip(cell_labels, topk_preds):
        if label in top5:
            num_correct += 1
    return num_correct / len(cell_labels)


def compute_geoguessr_metrics(results) -> Dict[str, float]:
    """reference evaluation/metrics.py:138-181 on the 11-tuple evaluate_model builds (train_eval_loop.py:138-140):
    the km-error statistics, the Under_*_km fractions, the GeoGuessr score and the geocell (top-k) accuracies, under the
    reference's keys.  Left out: `Country_accuracy` (geopandas country polygons, absent here)
    and the multi-task regressions (training-only heads).  Host numpy on the collected predictions, as in the reference."""
    predictions, cell_preds, _, _, _, top5_geocells, labels, cell_labels, _, _, _ = results
    cell_labels = np.asarray(cell_labels)
    if cell_labels.ndim > 1:
        cell_labels = np.argmax(cell_labels, axis=-1)
    distances = haversine_np(np.asarray(predictions), np.asarray(labels))      # dtypes as collected (fp32 preds: np.radians in fp32)
    eval_dict = {'Mean_km_error': np.mean(distances), 'Median_km_error': np.median(distances)}
    for km in (1, 5, 10, 25, 50, 100, 200, 750, 1000, 2500):
        eval_dict[f'Under_{km}_km'] = percentage_within_radius(distances, km)
    eval_dict['Geoguessr_score'] = geoguessr_score(distances)
    eval_dict['Geocell_accuracy'] = float(np.mean(cell_labels == np.asarray(cell_preds)))      # sklearn accuracy_score
    eval_dict['Geocell_top5_accuracy'] = topk_geocell_accuracy(cell_labels, top5_geocells)
    return eval_dict


@torch.no_grad()
def certain_forward(model: SuperGuessr, refiner: Optional[ProtoRefiner], pixel_values=None, embedding=None, labels=None,
                    labels_clf=None, **_unused):
    """`model(**data)` followed by `refiner(...)` with the reference's outputs guaranteed (round 5): one fast pass, the tolerance of
    every discrete decision downstream of the embedding -- the top-1 cell (pg_head_certainty) and, with a refiner, the winning
    candidate, the candidate-set boundary, the nearest prototype and the farthest member (pg_refine_certainty) -- and ONE exact
    re-encode (pg_vit_forward_precise) of the union of the samples that are not certain, after which their head outputs are
    recomputed.  One host synchronisation (the uncertain set is data dependent).  The refinement itself is left to the caller (it runs
    on the corrected embeddings / candidates: `evaluate_model` right away, `PanoramaPipeline.step` after its all-gather).
    Returns (outputs as `model.forward` returns them, info) with info = dict(certain (B,) bool: every output of the sample is the
    reference's; cause (B,) int32: why a sample was not certain after the fast pass; reencoded (n,) int64; head_tol, refine_tol (B,)
    f32 or None; boundary_checked)."""
    st = model.encode_head(pixel_values, embedding)
    info = dict(head_tol=st['tol'], refine_tol=None, refine_code=None, boundary_checked=None)
    can_fix = model.exact_top1 and st['pixel_values'] is not None
    flag = ~st['certain']
    # why a sample was sent to the exact tier (before anything is patched): 0 = certain, 1 = the head's top-1, else the refiner's
    # decision code (pg_refine_certainty: 1xxx winner, 2xxx set boundary, 3xxx nearest prototype, 4xxx farthest member, -9 underflow)
    cause = flag.to(torch.int32)
    W = model.cell_layer.weight.data
    if refiner is not None:
        _, _, rtol, rcode, checked = refiner.forward_certain(st['embedding'], st['preds_LLH'], st['topk_indices'], st['topk_values'],
                                                            W, model.wstats(False), model.certainty.drift_on(W.device))
        info.update(refine_tol=rtol, refine_code=rcode, boundary_checked=checked)
        r_unc = ~(rtol > model.certainty.threshold())
        cause = torch.where((cause == 0) & r_unc, rcode, cause)
        flag = flag | r_unc
    if can_fix:
        idx = torch.nonzero(flag).flatten()                                       # the step's one host synchronisation
        model.reencode_rows(st, idx)
        if refiner is not None and idx.numel():
            # the re-encoded samples, judged again at the exact tier's floor (no systematic part there)
            _, _, rt2, rc2, _ = refiner.forward_certain(st['embedding'][idx], st['preds_LLH'][idx], st['topk_indices'][idx],
                                                        st['topk_values'][idx], W, model.wstats(True), None)
            info['refine_tol'][idx], info['refine_code'][idx] = rt2, rc2
            flag[idx] = ~st['certain'][idx] | ~(rt2 > model.certainty.threshold(exact=True))
        elif idx.numel():
            flag[idx] = ~st['certain'][idx]
    info['certain'] = ~flag
    info['cause'] = cause
    info['reencoded'] = st['reencoded']
    out = model.package(st, labels, labels_clf)
    model.last_certain = info['certain']
    return out, info


def evaluate_model(model: SuperGuessr, dataset, metrics: Optional[Callable] = None, train_args=None,
                   refiner: Optional[ProtoRefiner] = None, yfcc: bool = False, writer=None, step: int = 0,
                   batch_size: Optional[int] = None, num_workers: int = 0):
    """reference training/train_eval_loop.py:35-161 (eval loop :77-112).

    dataset items are dicts of forward() keyword arguments (pixel_values | embedding, labels, labels_clf, ...).
    Returns the dict of concatenated numpy results; if `metrics` is given it is called with the same 11-tuple
    the reference builds (:138-140) and its dict is merged in.
    """
    from torch.utils.data import DataLoader
    logger.warning('Starting evaluation ...')
    if batch_size is None:
        batch_size = getattr(train_args, 'per_device_eval_batch_size', EVAL_BATCH_SIZE_PER_GPU)
    eval_data = DataLoader(dataset, batch_size, shuffle=False, pin_memory=False, num_workers=num_workers)
    model.eval()
    if refiner is not None:
        refiner.eval()
    combined_preds, combined_geocell_preds, combined_top5_cells, combined_top5_probs = [], [], [], []
    combined_loss = 0.0
    combined_certain = []
    n_seen = 0
    with torch.no_grad():
        for data in eval_data:
            # :80 `model(**data)` -- through certain_forward, which also looks at what the refiner will consume
            outputs, _info = certain_forward(model, refiner, **data)
            if outputs.loss_clf is not None:
                combined_loss += float(outputs.loss_clf) * len(data)              # :81-82 (`len(data)` as the reference)
            if refiner is not None:                                               # :98-103
                _, preds_LLH, _ = refiner(outputs.embedding, initial_preds=outputs.preds_LLH,
                                          candidate_cells=outputs.top5_geocells.indices,
                                          candidate_probs=outputs.top5_geocells.values)
                combined_preds.append(preds_LLH.cpu().detach().numpy())
            else:
                combined_preds.append(outputs.preds_LLH.cpu().detach().numpy())
            combined_geocell_preds.append(outputs.preds_geocell.cpu().detach().numpy())  # :106-112
            top5 = outputs.top5_geocells
            combined_top5_cells.append(top5.indices.cpu().detach().numpy())
            combined_top5_probs.append(top5.values.cpu().detach().numpy())
            combined_certain.append(_info['certain'].cpu().numpy())            # every output of the sample is the reference's
            n_seen += outputs.preds_geocell.shape[0]
    preds = np.concatenate(combined_preds, axis=0)
    preds_geocells = np.concatenate(combined_geocell_preds, axis=0)
    top5_geocells = np.concatenate(combined_top5_cells, axis=0)
    results = dict(preds=preds, preds_geocells=preds_geocells, top5_geocells=top5_geocells,
                   top5_probs=np.concatenate(combined_top5_probs, axis=0), loss_clf=combined_loss / max(n_seen, 1))
    if combined_certain:        # beyond the reference's keys: which samples' outputs are certain to be the fp32 reference's
        results['geocell_certain'] = np.concatenate(combined_certain, axis=0)
    if metrics is not None:                                                       # :122-140
        labels_lla, labels_cell = dataset['labels'], dataset['labels_clf']
        if isinstance(labels_lla, np.ndarray) == False:
            labels_lla, labels_cell = np.asarray(labels_lla), np.asarray(labels_cell)
        results.update(metrics((preds, preds_geocells, None, None, None, top5_geocells,
                                labels_lla, labels_cell, None, None, None)))
    model.train()
    logger.warning('Back to training ...')
    return results


def evaluate(model: str, dataset, yfcc: bool, landmarks: bool, base_model=None, heading: bool = False,
             refine: bool = True, geocell_path: Optional[str] = None, proto_path: Optional[str] = None,
             dataset_path=None, bank=None, head_state: Optional[str] = None):
    """reference evaluation/evaluate.py:10-85.

    `base_model`: as in the reference a STRING -- config.CLIP_MODEL for the pretrained tower, or the path of a checkpoint whose
    weights are copied over it by name (`load_state_dict`, :36-40; the pretrained tower is resolved from local files only, see
    `clip_embedder.load_pretrained_clip`) -- or, additionally, a ready module (`HipCLIPVisionModel`, or any module with a
    transformers CLIPVisionModel state dict), or None to evaluate on precomputed embeddings.  `model` is the path of
    the head checkpoint (`full_model.load_state(model)`, :46); evaluate() uses the reference's two refiner
    parameter sets (:73-80): first build -> ProtoRefiner(20, False, 10000, temperature=1); cached prototypes ->
    ProtoRefiner(40, False, 100000, temperature=0.6).  "Cached" means, in this order: a `bank` argument, the packed CSR
    file `<proto_model_path>.npz` this package writes at its first build, or the reference's own pickle at
    `proto_model_path` (`torch.save(refiner, ...)`, read with `load_refiner_cache` -- an existing
    saved_models/refiner/proto.refiner keeps working).
    """
    import os
    from . import config as cfg
    if isinstance(base_model, str):                                               # reference :36-40
        from .clip_embedder import HipCLIPVisionModel, load_pretrained_clip
        from .utils import load_state_dict
        path = base_model
        try:
            base_model = load_pretrained_clip()                                   # CLIP_MODEL, or the directory env PIGEON_CLIP_MODEL names
            if path != cfg.CLIP_MODEL:
                state_dict = torch.load(path, map_location='cpu')
                load_state_dict(base_model, state_dict)
                print(f'Initialized base model with weights from: {path}')
        except RuntimeError as why:
            if path == cfg.CLIP_MODEL or not os.path.exists(path):
                raise
            # no pretrained tower on this machine, but the checkpoint may carry all of it (base_model.* / vision_model.* names)
            state_dict = torch.load(path, map_location='cpu')
            state_dict = {('.'.join(k.split('.')[1:]) if 'base_model' in k.split('.')[0] else k): v for k, v in state_dict.items()}
            keys = {k[len('vision_model.'):] if k.startswith('vision_model.') else k for k in state_dict}
            if 'embeddings.patch_embedding.weight' not in keys or 'encoder.layers.0.mlp.fc2.weight' not in keys:
                raise RuntimeError(f'evaluate: {path!r} does not hold a complete vision tower and the pretrained one is unavailable '
                                   f'({why})') from why
            base_model = HipCLIPVisionModel(state_dict)
            print(f'Initialized base model with weights from: {path} (pretrained tower unavailable: checkpoint only)')
    full_model = SuperGuessr(base_model, panorama=True, hierarchical=False, multi_task=False, heading=heading,
                             freeze_base=True, yfcc=yfcc, num_candidates=50, geocell_path=geocell_path)
    # the reference's torch.load raises on a missing checkpoint (:46); only the explicit random-init names skip loading
    for ckpt in (head_state, model):
        if ckpt in (None, '', 'none', 'random'):
            continue
        if not os.path.exists(ckpt):
            raise FileNotFoundError(f'evaluate: checkpoint {ckpt!r} does not exist (pass "none" / "random" to evaluate '
                                    f'a randomly initialised head on purpose)')
        full_model.load_state(ckpt)
    full_model.to('cuda')
    print(full_model)
    refiner = None
    if refine:
        proto_model_path = cfg.PROTO_MODEL_YFCC_PATH if yfcc else cfg.PROTO_MODEL_PATH
        proto_path = proto_path or (cfg.PROTO_PATH_YFCC if yfcc else cfg.PROTO_PATH)
        dataset_path = dataset_path or (cfg.DATASET_PATH_YFCC if yfcc else cfg.DATASET_PATH)
        if landmarks:
            proto_path, proto_model_path = cfg.PROTO_PATH_LANDMARKS, cfg.PROTO_MODEL_LANDMARKS_PATH
            dataset_path = [cfg.DATASET_PATH_YFCC, cfg.DATASET_PATH_LANDMARKS]
        packed = proto_model_path + '.npz'
        protos = None
        if bank is None and not os.path.exists(packed):
            try:                                                                  # the reference's own cache (:65-69):
                protos = load_refiner_cache(proto_model_path)                     # torch.load(proto_model_path).protos
            except FileNotFoundError:
                pass
        if bank is not None:
            refiner = ProtoRefiner(40, False, 100000, bank=bank, temperature=0.6)
        elif os.path.exists(packed):                                              # cached bank, packed CSR form
            refiner = ProtoRefiner(40, False, 100000, bank=packed, temperature=0.6, verbose=False)
        elif protos is not None:                                                  # cached prototypes (:76-80)
            refiner = ProtoRefiner(40, False, 100000, proto_path=proto_path, dataset_path=dataset_path,
                                   protos=protos, temperature=0.6, verbose=False)
        else:                                                                     # first build (:72-75)
            refiner = ProtoRefiner(20, False, 10000, proto_path=proto_path, dataset_path=dataset_path, temperature=1)
            os.makedirs(os.path.dirname(packed) or '.', exist_ok=True)
            refiner.host_bank.save(packed)
        print(refiner)
    return evaluate_model(full_model, dataset, compute_geoguessr_metrics, None, refiner)


class PanoramaPipeline:
    """The data-parallel inference step (BASELINE.json configs[3]/[4]): every rank runs the ViT + geocell head on
    its shard of panoramas, ONE grouped all-gather moves per-image embeddings (B,4,1024) f32 + top-k candidates + initial
    predictions + sample indices to every rank (the reference's accelerator.gather, preprocessing/embed.py:36-37),
    each rank refines its 1/W slice of the gathered batch against its replica of the prototype bank, and a second, tiny
    grouped all-gather (12 bytes per panorama) concatenates the refined (lng,lat) / geocell of all slices, so that every
    rank -- rank 0 in particular -- holds the whole batch's result as the reference's collection loop does
    (training/train_eval_loop.py:98-112).  All outputs are rank-major; `distributed.restore_order(res['index'], ...)`
    puts them back in sample order."""

    def __init__(self, model: SuperGuessr, refiner: Optional[ProtoRefiner], comm: Optional[Communicator] = None):
        self.model, self.refiner = model, refiner
        self.comm = comm or Communicator()
        self.refine_events = None        # set to a list to collect (start, end) stream events around the refinement launches
        self.last_info = None            # certain_forward's info of the last step (certain, reencoded, tolerances)
        self.split_marks = None          # set to a list to collect five time stamps per step (see split_ms): compute vs gather(-wait)

    def _mark(self, marks, device):
        """A time stamp on the device's stream (a HIP event) or, for host tensors, the host clock."""
        if marks is None:
            return
        if device.type == 'cuda':
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
        else:
            import time
            marks.append(time.perf_counter())

    @staticmethod
    def split_ms(marks):
        """(compute ms, gather ms) of one step's five marks: [start, before gather 1, after it, before gather 2, end].  On the
        device the gather spans include the time this rank WAITS for the slowest rank to arrive at the collective."""
        if isinstance(marks[0], float):
            d = [(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])]
        else:
            d = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
        return d[0] + d[2], d[1] + d[3]

    @torch.no_grad()
    def step(self, pixel_values: torch.Tensor, index: Optional[torch.Tensor] = None):
        marks = [] if self.split_marks is not None else None
        self._mark(marks, pixel_values.device)
        if hasattr(self.model, 'encode_head'):
            # fast pass, tolerance of every decision the head AND the refinement below will take, one exact re-encode of the
            # samples that are not certain (model.exact_top1; off: the certainty is still reported)
            out, self.last_info = certain_forward(self.model, self.refiner, pixel_values=pixel_values)
        else:
            out, self.last_info = self.model(pixel_values=pixel_values, labels_clf=None), None     # stub models (bench.py --dry-run)
        B = out.preds_geocell.shape[0]
        if index is None:
            index = torch.arange(B, device=out.embedding.device) + self.comm.rank * B
        self._mark(marks, pixel_values.device)
        # The gather BEFORE the refinement is what `north_star` / the reference's accelerator.gather ask for, not a data dependency of
        # this step: every rank refines exactly the rows it produced (its own slice below), against its own replica of the bank.
        emb, topi, topv, llh, idx = self.comm.gather_many([out.embedding, out.top5_geocells.indices,
                                                           out.top5_geocells.values, out.preds_LLH, index.to(out.embedding.device)])
        self._mark(marks, pixel_values.device)
        res = dict(embedding=emb, index=idx, preds_geocell=topi[:, 0], preds_LLH=llh,
                   topk_indices=topi, topk_values=topv)
        if self.refiner is not None:
            r = self.comm.rank
            sl = slice(r * B, (r + 1) * B)                                        # this rank's slice of the gathered batch
            if self.refine_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _, ref_llh, ref_cell = self.refiner(emb[sl], initial_preds=llh[sl], candidate_cells=topi[sl],
                                                candidate_probs=topv[sl], quiet=True)
            if self.refine_events is not None:
                ev[1].record()
                self.refine_events.append(ev)
            self._mark(marks, pixel_values.device)
            res['refined_LLH'], res['refined_geocell'] = self.comm.gather_many([ref_llh, ref_cell])
        else:
            self._mark(marks, pixel_values.device)
        self._mark(marks, pixel_values.device)
        if marks is not None:
            self.split_marks.append(marks)
        return res
