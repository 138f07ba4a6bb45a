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
    """`model(**data)` followed by `refiner(...)` with the reference's discrete outputs (a z ~ 4 statistical statement for the samples called certain, pigeon_amd/certainty.py): one fast pass, the tolerance of every
    discrete decision downstream of the embedding -- the top-1 cell (pg_head_certainty) and, with a refiner, the winning candidate,
    the candidate-set boundary, the nearest prototype and the farthest member (pg_refine_certainty) -- and ONE exact pass
    (pg_vit_forward_precise) over the samples that are not certain, after which their head outputs AND their refinement are
    recomputed.  This is the settle-before-return form of `pigeon_amd.deferred.DeferredExact` (one host synchronisation: how many
    samples are uncertain is data dependent); loops over many batches use the engine's deferred form (`evaluate_model`,
    `PanoramaPipeline.submit`).
    Returns (outputs as `model.forward` returns them, info) with info = dict(certain (B,) bool: every output of the sample is the
    reference's; cause (B,) int32: why a sample was not certain after the fast pass; reencoded (n,) int64; head_tol, refine_tol,
    refine_code (B,) or None; boundary_checked; refined_LLH (B,2) f32 / refined_geocell (B,) i64: the refinement's result -- the
    caller need not run the refiner again)."""
    res = model.engine(refiner).submit(pixel_values, embedding)[0]
    st = dict(res['state'])
    out = model.package(st, labels, labels_clf)
    info = dict(certain=st['certain'], cause=st['cause'], head_tol=st['tol'], refine_tol=st.get('refine_tol'),
                refine_code=st.get('refine_code'), boundary_checked=model.engine(refiner).boundary_checked if refiner is not None else None,
                reencoded=torch.nonzero(st['exact']).flatten(), refined_LLH=st.get('refined_LLH'),
                refined_geocell=st.get('refined_geocell'))
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
    # The reference's loop (:77-112) computes a batch and appends its predictions; the predictions are only read after the loop
    # (:114-140).  Here a batch's rows that the 16-bit encoder cannot settle are queued on the device and re-encoded by the exact
    # encoder 40+ images at a time (pigeon_amd.deferred): a batch is appended when all its rows are settled -- a few iterations
    # late, in order -- and `flush()` settles the rest after the loop.  No host synchronisation per batch.
    from .deferred import DeferredExact
    engine = DeferredExact(model, refiner)

    def collect(done):
        # device tensors only: a `.cpu()` / `float()` here would wait for the step that has just been queued
        nonlocal combined_loss, n_seen
        for res in done:
            meta = res['meta']
            outputs = model.package(dict(res['state']), meta.get('labels'), meta.get('labels_clf'))
            if outputs.loss_clf is not None:
                combined_loss = combined_loss + outputs.loss_clf.detach() * meta['n_keys']   # :81-82 (`len(data)` as the reference)
            # :98-103: the refinement is the one the engine already ran (and re-ran for the rows the exact tier re-encoded)
            combined_preds.append(res['refined_LLH'] if refiner is not None else outputs.preds_LLH)
            combined_geocell_preds.append(outputs.preds_geocell)                 # :106-112
            top5 = outputs.top5_geocells
            combined_top5_cells.append(top5.indices)
            combined_top5_probs.append(top5.values)
            combined_certain.append(res['certain'])                              # every output of the sample is the reference's
            n_seen += outputs.preds_geocell.shape[0]

    with torch.no_grad():
        for data in eval_data:
            # :80 `model(**data)` and :98 `refiner(...)`, with what the refiner will consume checked as well
            keep = {'labels': data.get('labels'), 'labels_clf': data.get('labels_clf'), 'n_keys': len(data)}
            collect(engine.submit(data.get('pixel_values'), data.get('embedding'), meta=keep))
        collect(engine.flush())
    dropped = engine.check_nothing_dropped()
    if dropped:
        raise RuntimeError(f'evaluate_model: {dropped} uncertain rows did not fit the exact tier\'s queue (internal sizing error)')
    to_np = lambda parts: np.concatenate([t.cpu().detach().numpy() for t in parts], axis=0)     # noqa: E731
    preds = to_np(combined_preds)
    preds_geocells = to_np(combined_geocell_preds)
    top5_geocells = to_np(combined_top5_cells)
    combined_certain = [t.cpu().numpy() for t in combined_certain]
    results = dict(preds=preds, preds_geocells=preds_geocells, top5_geocells=top5_geocells,
                   top5_probs=to_np(combined_top5_probs), loss_clf=float(combined_loss) / max(n_seen, 1))
    if combined_certain:        # beyond the reference's keys: which samples' outputs are certain to be the fp32 reference's
        results['geocell_certain'] = np.concatenate(combined_certain, axis=0)
        # how many samples are STILL not certain after the exact tier (judged at its floor): their outputs are returned as computed
        results['uncertain_after_exact'] = int((~results['geocell_certain']).sum())
        results['exact_passes'] = [dict(f) for f in engine.flush_log]
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
    grouped all-gather concatenates the refined (lng,lat) / geocell of all slices (plus, round 6, the certainty flags and every rank's
    exact-tier queue length), so that every rank -- rank 0 in particular -- holds the whole batch's result as the reference's
    collection loop does (training/train_eval_loop.py:98-112).  All outputs are rank-major; `distributed.restore_order(res['index'],
    ...)` puts them back in sample order.

    Round 6: the step is `pigeon_amd.deferred.DeferredExact.submit` -- no host synchronisation, the rows the 16-bit encoder cannot
    settle wait in a device queue for ONE exact pass per `min_flush` panoramas (default: what fills one round of the CUs with the
    exact encoder's panels -- 7 panoramas on 256 CUs, `deferred.round_quantum`), taken by every rank in the same step on the same
    number of slots.  `submit` returns the steps that became final (a few steps late, in order), `flush` the rest; `step` = submit +
    flush (settled before it returns: the form tests, `smoke()` and single calls use)."""

    def __init__(self, model: SuperGuessr, refiner: Optional[ProtoRefiner], comm: Optional[Communicator] = None,
                 min_flush: Optional[int] = None, max_lag: int = 12, ops=None, pass_quantum: Optional[int] = None):
        from .deferred import DeferredExact
        self.model, self.refiner = model, refiner
        self.comm = comm or Communicator()
        self.engine = DeferredExact(model, refiner, self.comm, ops=ops, min_flush=min_flush, max_lag=max_lag,
                                    pass_quantum=pass_quantum)
        self.last_info = None            # of the newest result handed out: certain, cause, exact, queued

    @property
    def split_marks(self):
        return self.engine.marks

    @split_marks.setter
    def split_marks(self, v):
        """Set to a list to collect five time stamps per submitted step (see split_ms): compute vs gather(-wait)."""
        self.engine.marks = v

    @staticmethod
    def split_ms(marks):
        """(compute ms, gather ms) of one step's five marks: [start, before gather 1, after it, before gather 2, end].  On the
        device the gather spans include the time this rank WAITS for the slowest rank to arrive at the collective."""
        if isinstance(marks[0], float):
            d = [(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])]
        else:
            d = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
        return d[0] + d[2], d[1] + d[3]

    def _note(self, done):
        if done:
            r = done[-1]
            self.last_info = dict(certain=r['certain'], cause=r['cause'], exact=r['exact'], queued=r['queued'], step=r['step'],
                                  boundary_checked=self.engine.boundary_checked)
        return done

    @torch.no_grad()
    def submit(self, pixel_values: torch.Tensor, index: Optional[torch.Tensor] = None, meta=None):
        return self._note(self.engine.submit(pixel_values, index=index, meta=meta))

    @torch.no_grad()
    def flush(self):
        return self._note(self.engine.flush())

    @torch.no_grad()
    def step(self, pixel_values: torch.Tensor, index: Optional[torch.Tensor] = None):
        done = self.submit(pixel_values, index) + self.flush()
        return done[-1]
