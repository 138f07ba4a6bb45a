"""Seeded synthetic weights / inputs / geocells / prototype banks for the PIGEON hot path.

The reference ships no weights, geocells, prototypes or data (reference README.md:11), so every
parity test and the benchmark run on synthetic fixtures built here.

All generators are pure functions of their seed; they run identically in the authoring container and on
the GPU box, which is what lets small golden OUTPUT vectors (tests/golden/) stand in for the 1.2 GB of
ViT-L weights that cannot be committed.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Optional

import numpy as np
import torch

# ViT-L/14-336 geometry (reference config.py:6-7, `openai/clip-vit-large-patch14-336`)
IMAGE_SIZE = 336
PATCH = 14
GRID = IMAGE_SIZE // PATCH            # 24
TOKENS = GRID * GRID + 1              # 577
HIDDEN = 1024
HEADS = 16
HEAD_DIM = 64
MLP = 4096
LAYERS = 24
PATCH_K = 3 * PATCH * PATCH           # 588


def make_vit_weights(seed: int = 0, layers: int = LAYERS, affine_jitter: bool = False,
                     scale: float = 1.0, std_layers: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """State dict (transformers>=5 flat key layout) of a random-init CLIP ViT-L/14-336 vision tower.

    Distributions follow transformers' ``CLIPPreTrainedModel._init_weights`` (modeling_clip.py:404-428):
    class_embedding ~ N(0, D^-1/2), patch/position embedding ~ N(0, 0.02), q/k/v and fc2 ~
    N(0, D^-1/2 (2L)^-1/2), out_proj ~ N(0, D^-1/2), fc1 ~ N(0, (2D)^-1/2); biases 0, LayerNorm (1, 0).

    affine_jitter=True additionally randomises every bias and LayerNorm gamma/beta so that the bias /
    gamma / beta code paths of the kernels are actually exercised by parity tests (HF's default init
    leaves them at 0 / 1 / 0, which would hide a dropped bias).  ``scale`` multiplies the projection
    weights to push the network into a less benign numeric regime for stress tests.
    """
    g = torch.Generator().manual_seed(seed)
    D, F = HIDDEN, MLP
    L_std = layers if std_layers is None else std_layers

    def normal(shape, std):
        return torch.empty(shape, dtype=torch.float32).normal_(0.0, std, generator=g)

    in_std = D ** -0.5 * (2 * L_std) ** -0.5 * scale
    out_std = D ** -0.5 * scale
    fc_std = (2 * D) ** -0.5 * scale
    sd: Dict[str, torch.Tensor] = {}
    sd["embeddings.class_embedding"] = normal((D,), D ** -0.5)
    sd["embeddings.patch_embedding.weight"] = normal((D, 3, PATCH, PATCH), 0.02)
    sd["embeddings.position_embedding.weight"] = normal((TOKENS, D), 0.02)

    def ln(prefix):
        if affine_jitter:
            sd[prefix + ".weight"] = 1.0 + normal((D,), 0.1)
            sd[prefix + ".bias"] = normal((D,), 0.05)
        else:
            sd[prefix + ".weight"] = torch.ones(D)
            sd[prefix + ".bias"] = torch.zeros(D)

    def bias(n):
        return normal((n,), 0.02) if affine_jitter else torch.zeros(n)

    ln("pre_layrnorm")
    for i in range(layers):
        p = f"encoder.layers.{i}."
        sd[p + "self_attn.k_proj.weight"] = normal((D, D), in_std)
        sd[p + "self_attn.k_proj.bias"] = bias(D)
        sd[p + "self_attn.v_proj.weight"] = normal((D, D), in_std)
        sd[p + "self_attn.v_proj.bias"] = bias(D)
        sd[p + "self_attn.q_proj.weight"] = normal((D, D), in_std)
        sd[p + "self_attn.q_proj.bias"] = bias(D)
        sd[p + "self_attn.out_proj.weight"] = normal((D, D), out_std)
        sd[p + "self_attn.out_proj.bias"] = bias(D)
        ln(p + "layer_norm1")
        sd[p + "mlp.fc1.weight"] = normal((F, D), fc_std)
        sd[p + "mlp.fc1.bias"] = bias(F)
        sd[p + "mlp.fc2.weight"] = normal((D, F), in_std)
        sd[p + "mlp.fc2.bias"] = bias(D)
        ln(p + "layer_norm2")
    ln("post_layernorm")   # dead weight on this path but part of the state dict
    return sd


def make_vit_weights_trained_like(seed: int = 21, layers: int = LAYERS) -> Dict[str, torch.Tensor]:
    """Random weights pushed into the numeric regime of a TRAINED CLIP tower (stress fixture for the 16-bit operand
    chain; there are no public PIGEON weights, reference README.md:11):

      * massive activations -- a few (token, channel) entries of the residual stream two orders of magnitude above the
        rest, the well-known outlier features of trained ViTs: three tokens (CLS, 97, 333) get a 150x position-embedding
        spike on channels 7 and 519, which pre_layrnorm (gamma 10 on those channels) turns into |x| ~ 200;
      * rows whose mean dwarfs their spread -- pre_layrnorm.bias = 6 on every channel, so every residual row has
        |mean| / std >= 5 for the whole depth of the network (the regime in which rounding the UN-normalised row to
        16 bits costs digits that LayerNorm-then-round keeps);
      * LayerNorm gammas spanning 0.05 .. 1.6 (outlier channels damped, as trained models do) and non-zero betas/biases.
    """
    sd = make_vit_weights(seed=seed, layers=layers, affine_jitter=True)
    g = torch.Generator().manual_seed(seed + 1000)
    spikes_tok, spikes_ch = [0, 97, 333], [7, 519]
    pos = sd["embeddings.position_embedding.weight"]
    for t in spikes_tok:
        for c in spikes_ch:
            pos[t, c] += 3.0
    sd["pre_layrnorm.weight"][spikes_ch] = 10.0
    sd["pre_layrnorm.bias"] += 6.0
    for i in range(layers):
        for ln in ("layer_norm1", "layer_norm2"):
            w = sd[f"encoder.layers.{i}.{ln}.weight"]
            w.mul_(torch.empty(HIDDEN).uniform_(0.6, 1.5, generator=g))
            w[spikes_ch] = 0.05
            sd[f"encoder.layers.{i}.{ln}.bias"].add_(torch.empty(HIDDEN).normal_(0, 0.2, generator=g))
    return sd


def make_vit_weights_spread(seed: int = 31, layers: int = LAYERS, q_bias_std: float = 3.5, k_gain: float = 4.0,
                            v_gain: float = 4.0, heads_frac: float = 0.25,
                            base: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Random weights whose EMBEDDINGS spread the way a trained tower's do (round 4 fixture `pipeline24_spread`).

    A default-init tower maps every image to nearly the same token mean (pairwise cos-sim 0.94 .. 0.96): every per-token
    feature of iid pixels concentrates under the 577-token mean, and what is left is the input-independent part the MLPs
    write.  Trained towers escape that with GLOBAL, input-dependent attention: here a quarter of the heads of every layer
    (chosen per layer) get a large query BIAS (the query is then nearly the same for every token), a key projection with
    gain `k_gain` (score spread over the keys ~ q_bias_std * k_gain * 0.144 -> a softmax dominated by the few best-matching
    tokens of THIS image) and a value projection with gain `v_gain`: each such head writes the value of an input-selected
    token into every row of the residual stream, which survives the token mean at full strength.  No constant bias is
    added anywhere on the residual path; biases / LayerNorm parameters are jittered (affine_jitter) so that every term of
    the kernels is live.  Measured through the real reference on 512 seeded N(0,1) images (tests/golden/pipeline24_spread.npz):
    pairwise cos-sim of the 24-layer embeddings mean 0.62, min 0.38, max 0.75 (default init: 0.94 .. 0.96).  High-gain attention is
    also where 16-bit Q / K operands cost the most: tools/precision_sim.py predicts ~7e-4 relative embedding error for fp16
    operands on these weights (2.7e-4 default init).
    """
    # base: apply the attention modification to another state dict (e.g. make_vit_weights_trained_like: massive activations AND
    # spread embeddings in one tower); default = the jittered HF-init tower the committed fixture uses
    sd = make_vit_weights(seed=seed, layers=layers, affine_jitter=True) if base is None else {k: v.clone() for k, v in base.items()}
    g = torch.Generator().manual_seed(seed + 500)
    nh = max(1, int(HEADS * heads_frac))
    for i in range(layers):
        p = f"encoder.layers.{i}."
        for h in torch.randperm(HEADS, generator=g)[:nh].tolist():
            sl = slice(HEAD_DIM * h, HEAD_DIM * (h + 1))
            sd[p + "self_attn.q_proj.bias"][sl] = torch.empty(HEAD_DIM).normal_(0, q_bias_std, generator=g)
            sd[p + "self_attn.k_proj.weight"][sl] *= k_gain
            sd[p + "self_attn.v_proj.weight"][sl] *= v_gain
    return sd


def make_pixels(n_images: int, seed: int = 1234, panorama: bool = False) -> torch.Tensor:
    """Seeded N(0,1) pixels, the statistics of CLIP-normalised images.

    panorama=False -> (N,3,336,336); panorama=True -> (N/4, 12, 336, 336), the 4 panels concatenated on
    the channel axis exactly as reference preprocessing/dataset_preprocessing.py:199-200 does.
    """
    g = torch.Generator().manual_seed(seed)
    if panorama:
        assert n_images % 4 == 0
        return torch.randn((n_images // 4, 12, IMAGE_SIZE, IMAGE_SIZE), generator=g)
    return torch.randn((n_images, 3, IMAGE_SIZE, IMAGE_SIZE), generator=g)


def make_geocells(num_cells: int = 10000, seed: int = 0) -> np.ndarray:
    """(C,2) float64 [lng,lat] centroids, lng~U(-180,180), lat~U(-90,90)."""
    rng = np.random.default_rng(seed)
    lng = rng.uniform(-180.0, 180.0, size=num_cells)
    lat = rng.uniform(-90.0, 90.0, size=num_cells)
    return np.stack([lng, lat], axis=1)


def write_geocell_csv(path: str, centroids: np.ndarray) -> None:
    """CSV with the two columns the reference reads (models/super_guessr.py:171-172)."""
    import pandas as pd
    pd.DataFrame({"lng": centroids[:, 0], "lat": centroids[:, 1]}).to_csv(path, index=False)


def make_head_weights(num_cells: int, seed: int = 0, embed_dim: int = HIDDEN):
    """nn.Linear(embed_dim, C) default init: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
    g = torch.Generator().manual_seed(seed + 7)
    bound = 1.0 / math.sqrt(embed_dim)
    W = (torch.rand((num_cells, embed_dim), generator=g) * 2 - 1) * bound
    b = (torch.rand((num_cells,), generator=g) * 2 - 1) * bound
    return W, b


class SyntheticBank:
    """A CSR prototype bank + its training-embedding bank, in the arrays the kernels consume.

    proto_emb   (P,1024) f32   prototype embedding = mean of member embeddings (proto_refiner.py:359-378)
    cell_off    (C+1,)  i64    CSR offsets of each geocell's prototypes (empty cell -> zero-length)
    proto_lnglat(P,2)   f32    cluster centroid (`lng`,`lat` columns; float32 under torch format)
    proto_count (P,)    i32    `count` column
    member_off  (P+1,)  i64    CSR offsets into member_idx
    member_idx  (Nm,)   i64    `indices` column (rows of the training bank)
    train_emb   (Ntr,1024) f32 training embeddings (panel-averaged)
    train_lnglat(Ntr,2) f32    training `labels` [lng,lat]
    """

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_cells(self):
        return self.cell_off.shape[0] - 1


def make_bank(num_cells: int, protos_per_cell: int, seed: int = 2, dim: int = HIDDEN,
              empty_frac: float = 0.01, max_members: int = 8, exact_means: bool = True,
              jitter_protos: bool = True, center: Optional[np.ndarray] = None, radius: float = 1.0) -> SyntheticBank:
    """Synthetic prototype bank per SURVEY 8d: ~1% empty cells; 50% singleton clusters, 50% with 2..8
    members.  Training rows are laid out cluster by cluster so member lists are contiguous ranges (the
    kernels do not rely on that; member_idx is still an explicit index list).

    exact_means=True computes each prototype as the fp32 mean of its members exactly as the reference
    does (needed for parity against the reference's own prototype builder).  For the 1M-row perf bank
    exact_means=False draws prototypes directly (mean-of-members is irrelevant to kernel speed).

    center / radius (end-to-end fixtures): the training embeddings are placed around `center` (the mean query
    embedding, (dim,) f32) instead of the origin, every geocell at its own distance a_c in [0.3, 3] x radius
    (log-uniform) from it -- `radius` being the typical |query - center| -- so that the nearest-prototype distances
    of the candidate cells of one query differ by O(radius) and the refinement genuinely re-ranks candidates.
    """
    rng = np.random.default_rng(seed)
    empty = rng.random(num_cells) < empty_frac
    empty[-1] = False   # the reference sizes its proto list by max(geocell_idx)+1 (proto_refiner.py:75)
    n_per_cell = np.where(empty, 0, protos_per_cell).astype(np.int64)
    cell_off = np.zeros(num_cells + 1, dtype=np.int64)
    np.cumsum(n_per_cell, out=cell_off[1:])
    P = int(cell_off[-1])
    multi = rng.random(P) < 0.5
    count = np.where(multi, rng.integers(2, max_members + 1, size=P), 1).astype(np.int32)
    member_off = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(count.astype(np.int64), out=member_off[1:])
    Ntr = int(member_off[-1])
    member_idx = rng.permutation(Ntr).astype(np.int64)   # scattered rows: exercises the gather
    train_lnglat = np.stack([rng.uniform(-180, 180, Ntr), rng.uniform(-90, 90, Ntr)], axis=1).astype(np.float32)
    if exact_means:
        train_emb = rng.standard_normal((Ntr, dim), dtype=np.float32)
        if center is not None:
            a_cell = radius * np.exp(rng.uniform(np.log(0.3), np.log(3.0), size=num_cells))
            a = np.repeat(a_cell, n_per_cell).astype(np.float32)                   # (P,)
            centres = rng.standard_normal((P, dim), dtype=np.float32) * (a / np.sqrt(dim))[:, None]
            centres += np.asarray(center, dtype=np.float32)[None, :]
            rows = np.repeat(np.arange(P), count)
            train_emb[member_idx] = centres[rows] + np.float32(0.2 * radius / np.sqrt(dim)) * train_emb[member_idx]
        elif jitter_protos:
            # members of one cluster share a centre so that "nearest prototype" is meaningful
            centres = rng.standard_normal((P, dim), dtype=np.float32)
            rows = np.repeat(np.arange(P), count)
            train_emb[member_idx] = centres[rows] + 0.3 * train_emb[member_idx]
        te = torch.from_numpy(train_emb)
        proto_emb = np.empty((P, dim), dtype=np.float32)
        # torch mean over a cluster's members, as proto_refiner.py:378 -- all clusters of one size at once: (n_c, c, dim).mean(1) is
        # the same sequential sum over the members as (c, dim).mean(0) per cluster (bit-identical; the per-cluster loop took 16 - 200 s
        # of tiny torch calls for a 40 000-prototype bank on a loaded host)
        cnt = count.astype(np.int64)
        for c in np.unique(cnt):
            ps = np.nonzero(cnt == c)[0]
            for b in range(0, ps.size, 16384):                                        # (bounded: 16384 x c x dim floats per step)
                pb = ps[b:b + 16384]
                idx = member_idx[member_off[pb][:, None] + np.arange(int(c))[None, :]]  # (n, c) training rows
                proto_emb[pb] = te[torch.from_numpy(idx)].mean(dim=1).numpy()
    else:
        proto_emb = rng.standard_normal((P, dim), dtype=np.float32)
        train_emb = rng.standard_normal((Ntr, dim), dtype=np.float32)
    # cluster centroid = mean of member labels in float64 then stored float32 (torch format)
    proto_lnglat = np.empty((P, 2), dtype=np.float32)
    lab = train_lnglat.astype(np.float64)
    sums = np.add.reduceat(lab[member_idx], member_off[:-1], axis=0) if P > 0 else np.zeros((0, 2))
    proto_lnglat[:] = (sums / count[:, None]).astype(np.float32)
    return SyntheticBank(proto_emb=proto_emb, cell_off=cell_off, proto_lnglat=proto_lnglat,
                         proto_count=count, member_off=member_off, member_idx=member_idx,
                         train_emb=train_emb, train_lnglat=train_lnglat)


def write_bank_reference_files(bank: SyntheticBank, proto_csv: str, dataset_dir: str) -> None:
    """Write the bank in the two on-disk formats the REFERENCE consumes:
    the prototype CSV (columns geocell_idx,cluster,lng,lat,count,indices with `indices` a JSON list;
    dataset_creation/prototype/prototype.py:87-95, models/proto_refiner.py:70-76) and a HF DatasetDict
    with split `train`, columns `embedding` (1024,) f32 and `labels` (2,) [lng,lat]
    (models/proto_refiner.py:67,247-255,370-376)."""
    import pandas as pd
    import datasets
    rows = []
    C = bank.num_cells
    for c in range(C):
        for j, p in enumerate(range(int(bank.cell_off[c]), int(bank.cell_off[c + 1]))):
            idx = bank.member_idx[bank.member_off[p]:bank.member_off[p + 1]].tolist()
            rows.append(dict(geocell_idx=c, cluster=j, lng=float(bank.proto_lnglat[p, 0]),
                             lat=float(bank.proto_lnglat[p, 1]), count=int(bank.proto_count[p]),
                             indices=json.dumps(idx)))
    # the reference sizes its table by max(geocell_idx)+1 (proto_refiner.py:75): make sure the last cell exists
    pd.DataFrame(rows).to_csv(proto_csv, index=False)
    ds = datasets.Dataset.from_dict({
        "embedding": bank.train_emb,
        "labels": bank.train_lnglat,
    })
    ds.set_format("torch")
    datasets.DatasetDict(train=ds).save_to_disk(dataset_dir)


def env_cores() -> int:
    return os.cpu_count() or 1


def make_bank_device(num_cells: int, protos_per_cell: int, seed: int = 2, dim: int = HIDDEN, empty_frac: float = 0.01,
                     max_members: int = 8, device: str = "cuda"):
    """Perf-size bank (10 000 cells x 100 prototypes = 1M x 1024 fp32 = 4.1 GB, ~3M training rows =
    12 GB): the index arrays are built exactly like make_bank(); the two big embedding matrices are drawn
    directly in HBM (seeded torch generator) instead of 16 GB of host RNG + PCIe.  Returns a dict of device tensors
    in the pg_bank layout (usable as `ProtoRefiner(bank=...)` via hip_ops.DeviceBank)."""
    rng = np.random.default_rng(seed)
    empty = rng.random(num_cells) < empty_frac
    empty[-1] = False
    n_per_cell = np.where(empty, 0, protos_per_cell).astype(np.int64)
    cell_off = np.zeros(num_cells + 1, dtype=np.int64)
    np.cumsum(n_per_cell, out=cell_off[1:])
    P = int(cell_off[-1])
    multi = rng.random(P) < 0.5
    count = np.where(multi, rng.integers(2, max_members + 1, size=P), 1).astype(np.int32)
    member_off = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(count.astype(np.int64), out=member_off[1:])
    Ntr = int(member_off[-1])
    member_idx = rng.permutation(Ntr).astype(np.int64)
    g = torch.Generator(device=device).manual_seed(seed)
    proto_emb = torch.randn((P, dim), generator=g, device=device)
    train_emb = torch.randn((Ntr, dim), generator=g, device=device)
    lng = torch.rand((Ntr,), generator=g, device=device) * 360 - 180
    lat = torch.rand((Ntr,), generator=g, device=device) * 180 - 90
    train_lnglat = torch.stack([lng, lat], dim=1).contiguous()
    plng = torch.rand((P,), generator=g, device=device) * 360 - 180
    plat = torch.rand((P,), generator=g, device=device) * 180 - 90
    return dict(proto_emb=proto_emb, cell_off=torch.from_numpy(cell_off).to(device),
                proto_lnglat=torch.stack([plng, plat], dim=1).contiguous(),
                proto_count=torch.from_numpy(count).to(device), member_off=torch.from_numpy(member_off).to(device),
                member_idx=torch.from_numpy(member_idx).to(device), train_emb=train_emb, train_lnglat=train_lnglat)
