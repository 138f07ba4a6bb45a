"""ProtoRefiner (prototype-distance guess refinement) on the HIP kernels, behind the reference's call surface.

Mirrors reference models/proto_refiner.py: same constructor arguments, same `forward(embedding, geo_tensor,
initial_preds, candidate_cells, candidate_probs, cluster) -> (loss, preds_LLH, preds_geocell)` and the same
status print.  The per-sample / per-candidate Python loop (:154-222) is replaced by pg_refine_forward over a
CSR prototype bank resident in HBM.  `hedge=True` is out of scope (disabled in the reference's final model,
models/README.md:11) and raises.
"""
from __future__ import annotations

import json
from typing import List

import numpy as np
import torch
from torch import nn, Tensor
from torch.nn.parameter import Parameter

from . import hip_ops
from .config import DATASET_PATH, PROTO_PATH


class HostBank:
    """CSR prototype bank on the host (numpy), the layout include/pigeon_hip.h `pg_bank` documents."""
    FIELDS = ("proto_emb", "cell_off", "proto_lnglat", "proto_count", "member_off", "member_idx",
              "train_emb", "train_lnglat")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    @property
    def num_cells(self):
        return self.cell_off.shape[0] - 1

    def save(self, path: str):
        """Packed binary bank (one .npz): loads in seconds where the reference's 64-process Arrow build
        (proto_refiner.py:257-313) takes minutes."""
        np.savez(path, **{f: getattr(self, f) for f in self.FIELDS})

    @classmethod
    def load(cls, path: str) -> "HostBank":
        z = np.load(path)
        return cls(**{f: z[f] for f in cls.FIELDS})


def _load_indices(index_json, verbose=False):
    """reference models/proto_refiner.py:92-108"""
    try:
        return json.loads(index_json)
    except TypeError:
        if verbose:
            print('Couldn\'t load a geocell.')
        return []


def _training_arrays(dataset_path):
    """`DatasetDict.load_from_disk(dataset_path)['train']` -> (Ntr,1024) f32 panel-averaged embeddings and
    (Ntr,2) f32 [lng,lat] labels (reference :48-67, :247-255, :370-376)."""
    import datasets
    if type(dataset_path) == list:
        if len(dataset_path) > 2:
            raise NotImplementedError('Can\'t concatentate more than 2 datasets.')
        parts = [datasets.DatasetDict.load_from_disk(p)['train'] for p in dataset_path]
        train = datasets.concatenate_datasets([p.remove_columns([c for c in ('labels_climate',) if c in p.column_names])
                                               for p in parts])
    else:
        train = datasets.DatasetDict.load_from_disk(dataset_path)['train']
    train = train.with_format('numpy')
    emb = np.ascontiguousarray(np.asarray(train['embedding'][:], dtype=np.float32))
    if emb.ndim == 3:                                             # (N,4,1024): mean over panels (:252-253,:374-375)
        emb = _panel_mean_gpu(emb)
    lab = np.asarray(train['labels'][:], dtype=np.float32)
    return np.ascontiguousarray(emb), np.ascontiguousarray(lab[:, :2])


def _panel_mean_gpu(emb4: np.ndarray, device="cuda", rows_per_call: int = 1 << 20) -> np.ndarray:
    """`embeddings.mean(dim=1)` of a (N,4,1024) training bank on the GPU: pg_proto_build in its 4-panel mode with one
    single-member "prototype" per row (sum of the 4 panels in order, x 0.25 -- bit-identical to torch's CPU mean)."""
    n = emb4.shape[0]
    out = np.empty((n, emb4.shape[2]), dtype=np.float32)
    for s in range(0, n, rows_per_call):
        e = min(n, s + rows_per_call)
        ar = torch.arange(e - s + 1, device=device, dtype=torch.int64)
        out[s:e] = hip_ops.proto_build(torch.from_numpy(emb4[s:e]).to(device), ar, ar[:-1].contiguous()).cpu().numpy()
    return out


def _segmented_mean_gpu(train_emb, member_off, member_idx, device="cuda"):
    """pg_proto_build: one block per prototype streams its member rows (HBM-bound), same summation order."""
    if member_idx.size and (member_idx.min() < 0 or member_idx.max() >= train_emb.shape[0]):
        raise IndexError("prototype member index outside the training bank")
    out = hip_ops.proto_build(torch.from_numpy(train_emb).to(device), torch.from_numpy(member_off).to(device),
                              torch.from_numpy(member_idx).to(device))
    return out.cpu().numpy()


def build_bank(proto_path: str, dataset_path, verbose: bool = False) -> HostBank:
    """CSV + training embeddings -> CSR bank, reproducing the reference's prototype construction:
    rows of one geocell in CSV order (`proto_df.loc[cell]`, :299), a cell whose FIRST row has no indices is
    empty (:307-308), prototype embedding = fp32 mean of the member embeddings (:359-378), lng/lat/count
    columns become float32/int under the torch format (:312)."""
    import pandas as pd
    train_emb, train_lnglat = _training_arrays(dataset_path)
    df = pd.read_csv(proto_path)
    df['indices'] = df['indices'].apply(lambda s: _load_indices(s, verbose))
    df['geocell_idx'] = df['geocell_idx'].astype(int)
    num_cells = int(df['geocell_idx'].max()) + 1                  # :75
    order = np.argsort(df['geocell_idx'].values, kind='stable')   # keeps CSV order inside a cell
    cells = df['geocell_idx'].values[order]
    idx_lists = [df['indices'].values[i] for i in order]
    lng = df['lng'].values[order].astype(np.float32)
    lat = df['lat'].values[order].astype(np.float32)
    cnt = df['count'].values[order].astype(np.int32)
    keep = np.ones(len(order), dtype=bool)
    start = 0
    while start < len(order):                                     # drop cells whose first row is empty (:307-308)
        end = start
        while end < len(order) and cells[end] == cells[start]:
            end += 1
        if len(idx_lists[start]) == 0:
            keep[start:end] = False
        start = end
    cells, lng, lat, cnt = cells[keep], lng[keep], lat[keep], cnt[keep]
    idx_lists = [l for l, k in zip(idx_lists, keep) if k]
    P = len(idx_lists)
    cell_off = np.zeros(num_cells + 1, dtype=np.int64)
    np.add.at(cell_off, cells + 1, 1)
    cell_off = np.cumsum(cell_off)
    lens = np.array([len(l) for l in idx_lists], dtype=np.int64)
    member_off = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(lens, out=member_off[1:])
    member_idx = np.fromiter((i for l in idx_lists for i in l), dtype=np.int64, count=int(member_off[-1]))
    # prototype embedding = mean of the member embeddings (:359-378): pg_proto_build, torch's summation order (no host path)
    proto_emb = _segmented_mean_gpu(train_emb, member_off, member_idx) if P else np.zeros((0, train_emb.shape[1]), np.float32)
    return HostBank(proto_emb=proto_emb, cell_off=cell_off, proto_lnglat=np.stack([lng, lat], axis=1),
                    proto_count=cnt, member_off=member_off, member_idx=member_idx,
                    train_emb=train_emb, train_lnglat=train_lnglat)


def bank_from_protos(protos: List, dataset_path) -> HostBank:
    """Convert the reference's own `refiner.protos` (list of per-cell HF Datasets or None, the object
    evaluation/evaluate.py:66-75 pickles and reloads) into the CSR bank."""
    train_emb, train_lnglat = _training_arrays(dataset_path)
    embs, lnglat, cnt, idxs, cell_off = [], [], [], [], [0]
    for cell in protos:
        if cell is not None:
            c = cell.with_format('numpy')
            embs.append(np.asarray(c['embedding'][:], dtype=np.float32))
            lnglat.append(np.stack([np.asarray(c['lng'][:], np.float32), np.asarray(c['lat'][:], np.float32)], axis=1))
            cnt.append(np.asarray(c['count'][:], np.int32))
            idxs.extend([list(map(int, x)) for x in c['indices'][:]])
        cell_off.append(cell_off[-1] + (0 if cell is None else len(cell)))
    lens = np.array([len(l) for l in idxs], dtype=np.int64)
    member_off = np.zeros(len(idxs) + 1, dtype=np.int64)
    np.cumsum(lens, out=member_off[1:])
    return HostBank(proto_emb=np.concatenate(embs) if embs else np.zeros((0, 1024), np.float32),
                    cell_off=np.asarray(cell_off, np.int64),
                    proto_lnglat=np.concatenate(lnglat) if lnglat else np.zeros((0, 2), np.float32),
                    proto_count=np.concatenate(cnt) if cnt else np.zeros((0,), np.int32), member_off=member_off,
                    member_idx=np.fromiter((i for l in idxs for i in l), dtype=np.int64, count=int(member_off[-1])),
                    train_emb=train_emb, train_lnglat=train_lnglat)


def load_refiner_cache(path: str):
    """Read the refiner cache the reference writes with `torch.save(refiner, proto_model_path)` and reads back as
    `torch.load(proto_model_path).protos` (evaluation/evaluate.py:64-75) -- WITHOUT the reference's `models` package on the
    path: the pickle names `models.proto_refiner.ProtoRefiner` (and, for hedged refiners, `models.layers...`), which a custom
    unpickler maps onto an empty nn.Module shell that just receives the pickled attribute dict.  Also reads a refiner of THIS
    package saved the same way.  Returns `.protos`: the reference's list of per-cell HF Datasets / None (feed it to
    `ProtoRefiner(protos=...)` / `bank_from_protos`) or this package's HostBank.
    Raises FileNotFoundError like torch.load does (the reference catches exactly that, :68)."""
    import pickle
    import types

    class _Shell(nn.Module):
        """stands in for any class of the reference's `models` package found in the pickle"""

    # A pickle executes what it names.  This is an ALLOW-LIST, not a sandbox: it narrows what a cache file can reach to the data model a
    # refiner cache really holds -- the reference's own ProtoRefiner (-> _Shell), torch tensor / parameter reconstruction, `datasets`
    # tables and features, pyarrow / numpy / pandas array reconstruction, plain containers -- by exact module and, where a module also
    # exports callables that do more than build data, by exact name.  Dotted names (attribute chains such as `Dataset.load_from_disk`)
    # are refused, and so is everything else (os, subprocess, builtins.getattr / eval, types, functools, dill, multiprocess ...).
    # Only load caches you (or the reference on this machine) wrote; the packed `.npz` bank (`HostBank.save`) needs no pickle at all.
    _any = None
    allowed = {
        'collections': {'OrderedDict', 'defaultdict'},
        '_codecs': {'encode'},
        'copyreg': {'_reconstructor'}, 'copy_reg': {'_reconstructor'},      # (protocol 0 / 1 pickles use the Python 2 module name)
        'numpy': {'dtype', 'ndarray'},
        'numpy.core.multiarray': {'_reconstruct', 'scalar'}, 'numpy._core.multiarray': {'_reconstruct', 'scalar'},
        'numpy.core.numeric': {'_frombuffer'}, 'numpy._core.numeric': {'_frombuffer'},
        'torch._utils': {'_rebuild_tensor', '_rebuild_tensor_v2', '_rebuild_parameter', '_rebuild_parameter_with_state'},
        'torch': {'FloatStorage', 'DoubleStorage', 'HalfStorage', 'BFloat16Storage', 'LongStorage', 'IntStorage', 'ShortStorage',
                  'CharStorage', 'ByteStorage', 'BoolStorage', 'Size', 'device'},
        'torch.storage': {'UntypedStorage', 'TypedStorage'},      # `_load_from_bytes` is handled below: it re-enters THIS unpickler
        'torch.nn.parameter': {'Parameter'},
        'datasets.arrow_dataset': {'Dataset'}, 'datasets.dataset_dict': {'DatasetDict'},
        'datasets.features.features': _any, 'datasets.features': _any, 'datasets.info': _any, 'datasets.table': _any,
        'datasets.splits': _any, 'datasets.utils.version': _any, 'datasets.naming': _any,
        'pyarrow.lib': _any,
        'pandas.core.frame': {'DataFrame'}, 'pandas.core.series': {'Series'},
        'pandas._libs.internals': {'_unpickle_block'}, 'pandas.core.internals.managers': {'BlockManager', 'SingleBlockManager'},
        'pandas.core.internals.blocks': {'new_block'},
        'pandas.core.indexes.base': {'Index', '_new_Index'}, 'pandas.core.indexes.range': {'RangeIndex'},
        'pandas.core.indexes.numeric': {'Int64Index', 'Float64Index'},
    }
    # `object`: what copyreg._reconstructor (pickle protocols 0 / 1) names as the base of a plain class; it builds nothing by itself
    allowed_builtins = {'set', 'frozenset', 'list', 'dict', 'tuple', 'bytes', 'bytearray', 'str', 'int', 'float', 'bool', 'complex',
                        'slice', 'range', 'NoneType', 'object'}

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == 'models' or module.startswith('models.'):
                return _Shell
            if module == 'torch.storage' and name == '_load_from_bytes':
                # torch's own is `torch.load(io.BytesIO(b), weights_only=False)` with the STOCK unpickler: a cache carrying
                # `_load_from_bytes(<nested pickle>)` would resolve any global through it.  The nested payload goes through the same
                # allow-list instead.
                return _load_from_bytes_restricted
            ok = False
            if '.' not in name:
                if module in ('builtins', '__builtin__'):
                    ok = name in allowed_builtins
                elif module in allowed:
                    ok = allowed[module] is _any or name in allowed[module]
            if ok:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f'refiner cache {path!r} names {module}.{name}: not part of the data model a refiner cache holds '
                                         f'(the reference\'s models.*, torch / numpy / pandas / pyarrow / datasets array and table '
                                         f'reconstruction, plain containers)')

    import io
    pm = types.ModuleType('pigeon_amd._refiner_pickle')
    pm.Unpickler = _Unpickler
    pm.load = lambda f, **kw: _Unpickler(f, **kw).load()
    pm.loads = lambda b, **kw: _Unpickler(io.BytesIO(b), **kw).load()
    pm.__name__ = 'pickle'                     # torch.load only checks for the attributes it uses

    def _load_from_bytes_restricted(b):
        return torch.load(io.BytesIO(b), map_location='cpu', pickle_module=pm, weights_only=False)

    obj = torch.load(path, map_location='cpu', pickle_module=pm, weights_only=False)
    if not hasattr(obj, 'protos'):
        raise ValueError(f'{path}: the pickled object has no `.protos` (not a refiner cache)')
    return obj.protos


class ProtoRefiner(nn.Module):
    """Proto-Net refinement model (reference models/proto_refiner.py:17-90)."""

    def __init__(self, topk: int = 5, hedge: bool = False, max_refinement: int = 1000,
                 temperature: float = 1.6, proto_path: str = PROTO_PATH,
                 dataset_path: str = DATASET_PATH, protos: List = None,
                 verbose: bool = False, bank=None, device: str = 'cuda'):
        """Arguments as the reference.  Extra: `bank` (HostBank / SyntheticBank / path to a packed .npz) supplies
        the CSR bank directly; `protos` accepts the reference's pickled list of per-cell datasets."""
        super(ProtoRefiner, self).__init__()
        if hedge:
            raise NotImplementedError('hedge=True is out of scope: disabled in the final model (models/README.md:11)')
        self.topk = topk
        self.hedge = hedge
        self.max_refinement = max_refinement
        self.verbose = verbose
        if bank is not None:
            host = HostBank.load(bank) if isinstance(bank, str) else (HostBank(**bank) if isinstance(bank, dict) else bank)
        elif protos is not None:
            host = protos if isinstance(protos, HostBank) else bank_from_protos(protos, dataset_path)
        else:
            print('Initializing ProtoRefiner. This might take a while ...')
            host = build_bank(proto_path, dataset_path, verbose)
            print('Initialization of ProtoRefiner complete.')
        self.host_bank = host
        self.num_geocells = host.cell_off.shape[0] - 1
        self.protos = host                     # attribute name kept (evaluate.py:66-75 reads `ref.protos`)
        self.temperature = Parameter(torch.tensor(temperature), requires_grad=False)
        self.geo_scaling = Parameter(torch.tensor(20.), requires_grad=False)
        self._dbank = None
        self._device = device

    def _temperature_value(self) -> float:
        """float(temperature), read on EVERY call as the reference does (`self.temperature` enters the softmax of each
        forward, proto_refiner.py:187): an in-place edit through `.data` bumps no version counter, so nothing may be cached.
        The Parameter normally lives on the host (neither the reference nor evaluate() moves the refiner to the GPU), where
        this costs nothing; on a device it is one scalar D2H read per forward."""
        return float(self.temperature.data)

    def _device_bank(self, device) -> hip_ops.DeviceBank:
        if self._dbank is None:
            self._dbank = hip_ops.DeviceBank(self.host_bank, device=device)
        return self._dbank

    def __str__(self):
        rep = 'ProtoRefiner(\n'
        rep += f'\ttopk\t\t= {self.topk}\n'
        rep += f'\thedge\t\t= {self.hedge}\n'
        rep += f'\tmax_refinement\t= {self.max_refinement}\n'
        rep += f'\ttemperature\t= {self.temperature.data.item()}\n'
        rep += f'\tgeo_scaling\t= {self.geo_scaling.data.item()}\n'
        rep += ')'
        return rep

    def forward(self, embedding: Tensor = None, geo_tensor: Tensor = None, initial_preds: Tensor = None,
                candidate_cells: Tensor = None, candidate_probs: Tensor = None, cluster: Tensor = None,
                quiet: bool = False):
        """reference models/proto_refiner.py:121-231.  Returns (loss, preds_LLH (B,2) f32, preds_geocell (B,) i64)."""
        assert self.topk <= candidate_cells.size(1), \
            '"topk" parameter must be smaller or equal to the number of geocell candidates \
             passed into the forward function.'
        dev = embedding.device if embedding.is_cuda else torch.device(self._device)
        if dev.type != 'cuda':
            raise RuntimeError('pigeon_amd.ProtoRefiner runs on the GPU only (no CPU fallback)')
        with torch.no_grad():
            q = embedding.to(dev, torch.float32).contiguous()
            init = initial_preds.to(dev, torch.float64).contiguous()
            cand = candidate_cells.to(dev, torch.int64).contiguous()
            probs = None if candidate_probs is None else candidate_probs.to(dev, torch.float32).contiguous()
            preds_LLH, preds_geocell, guess_index, self.last_scratch = hip_ops.refine_forward(
                self._device_bank(dev), q, init, cand, probs, self.topk,
                self._temperature_value(), float(self.max_refinement), return_scratch=True)
            if not quiet:                                          # :224-227 (costs one D2H sync, like the reference)
                perc_changed = (guess_index != 0).sum() / guess_index.size(0)
                print(f'Changed geocell predictions of {perc_changed * 100:.1f} % of guesses.')
        loss = 0 if self.training else None
        return loss, preds_LLH, preds_geocell

    # candidates evaluated beyond `topk` by the certainty pass (could one of them enter the set and win?)
    EXTRA_EVAL = 4

    @torch.no_grad()
    def forward_certain(self, embedding: Tensor, initial_preds: Tensor, candidate_cells: Tensor, candidate_probs: Tensor,
                        head_weight: Tensor, wstats: Tensor, drift: Tensor = None):
        """`forward` plus the TOLERANCE of its discrete outputs against an embedding error (round 5; error model:
        pigeon_amd/certainty.py, kernels: pg_refine_forward_ex + pg_refine_certainty).  `candidate_cells` / `candidate_probs` may hold
        more than `topk` candidates (SuperGuessr computes `num_candidates + 4`): up to 4 of those beyond `topk` are evaluated too --
        never selected -- so that the pass can tell whether a cell just outside the set could get in and win; with none beyond
        `topk` that question stays open (`boundary_checked` False).  `head_weight` (C,1024) is the geocell head's weight matrix: the
        candidates' probabilities move with the embedding through it.
        Returns (preds_LLH (B,2) f32, preds_geocell (B,) i64, tol (B,) f32, code (B,) i32, boundary_checked)."""
        assert self.topk <= candidate_cells.size(1)
        dev = embedding.device
        if dev.type != 'cuda':
            raise RuntimeError('pigeon_amd.ProtoRefiner runs on the GPU only (no CPU fallback)')
        q = embedding.to(dev, torch.float32).contiguous()
        init = initial_preds.to(dev, torch.float64).contiguous()
        cand = candidate_cells.to(dev, torch.int64).contiguous()
        probs = None if candidate_probs is None else candidate_probs.to(dev, torch.float32).contiguous()
        n_eval = min(cand.shape[1], self.topk + self.EXTRA_EVAL)
        bank = self._device_bank(dev)
        T = self._temperature_value()
        llh, cell, choice, refined, scratch = hip_ops.refine_forward_ex(bank, q, init, cand, probs, self.topk, n_eval, T,
                                                                       float(self.max_refinement))
        tol, code = hip_ops.refine_certainty(bank, q, cand, probs, self.topk, scratch, head_weight, drift, wstats, T, refined, choice)
        self.last_scratch = scratch[:, :self.topk, :4]
        return llh, cell, tol, code, n_eval > self.topk
