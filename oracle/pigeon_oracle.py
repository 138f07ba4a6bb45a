"""ORACLE -- CPU restatement of the PIGEON inference hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  Nothing under ``pigeon_amd/`` does, and the product path fails loudly without its HIP library.

What it restates (plain PyTorch fp32 / float64 on CPU, no HuggingFace modules, no reference imports):

  * ``vit_last_hidden_state`` -- transformers ``CLIPVisionModel.forward`` as called at reference
    models/clip_embedder.py:63 and models/super_guessr.py:395.  The arithmetic lives in third-party
    ``transformers`` (reference pins 4.23.1, env.yml:60; this image has 5.15.0, same math):
    modeling_clip.py ``CLIPVisionEmbeddings.forward`` (patch conv k=s=14 no bias, CLS concat, +position),
    ``pre_layrnorm``, 24 x ``CLIPEncoderLayer.forward`` (pre-LN; q/k/v/out Linear with bias; scale
    d_h^-1/2; fp32 softmax; ``quick_gelu`` x*sigmoid(1.702x); eps 1e-5).  ``last_hidden_state`` is taken
    BEFORE ``post_layernorm``.
  * ``clip_embedding`` -- models/clip_embedder.py:63-65: mean over all 577 tokens.
  * ``super_guessr_forward`` -- models/super_guessr.py:386-405,437,447-459 (inference branch).
  * ``proto_refiner_forward`` -- models/proto_refiner.py:121-231,233-255,332-357.
  * ``haversine`` -- preprocessing/geo_utils.py:40-55.

Pinning: the reference has no tests / golden vectors, so this restatement is pinned
against OUTPUTS OF THE REFERENCE ITSELF: ``oracle/make_golden.py`` runs the real reference modules
(through ``oracle/reference_loader.py``) on seeded synthetic inputs and commits the results under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against them.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Dict, Optional

import torch
import torch.nn.functional as F

TOKENS = 577
HIDDEN = 1024
HEADS = 16
HEAD_DIM = 64
LN_EPS = 1e-5

TopK = namedtuple("TopK", "values indices")


def _strip(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept both key layouts: transformers 4.23.1 ``vision_model.*`` and 5.x flat."""
    out = {}
    for k, v in sd.items():
        if k.startswith("vision_model."):
            k = k[len("vision_model."):]
        out[k] = v
    return out


def num_layers(sd) -> int:
    sd = _strip(sd)
    n = 0
    while f"encoder.layers.{n}.layer_norm1.weight" in sd:
        n += 1
    return n


@torch.no_grad()
def vit_last_hidden_state(sd: Dict[str, torch.Tensor], pixels: torch.Tensor,
                          collect: Optional[dict] = None) -> torch.Tensor:
    """(N,3,336,336) fp32 -> (N,577,1024) fp32, HF CLIPVisionModel.last_hidden_state."""
    sd = _strip(sd)
    x = pixels.to(torch.float32)
    N = x.shape[0]
    # CLIPVisionEmbeddings.forward (modeling_clip.py:202-218)
    pe = F.conv2d(x, sd["embeddings.patch_embedding.weight"], bias=None, stride=14)
    pe = pe.flatten(2).transpose(1, 2)                                   # (N,576,1024)
    cls = sd["embeddings.class_embedding"].expand(N, 1, -1)
    h = torch.cat([cls, pe], dim=1) + sd["embeddings.position_embedding.weight"][None]
    # CLIPVisionModel.forward: pre_layrnorm -> encoder
    h = F.layer_norm(h, (HIDDEN,), sd["pre_layrnorm.weight"], sd["pre_layrnorm.bias"], LN_EPS)
    if collect is not None:
        collect["pre_ln"] = h.clone()
    L = num_layers(sd)
    scale = HEAD_DIM ** -0.5
    for i in range(L):
        p = f"encoder.layers.{i}."
        # CLIPEncoderLayer.forward (modeling_clip.py:362-384)
        r = h
        y = F.layer_norm(h, (HIDDEN,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], LN_EPS)
        # CLIPAttention.forward (:298-335) + eager_attention_forward (:259-277)
        q = F.linear(y, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
        k = F.linear(y, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(y, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q = q.view(N, TOKENS, HEADS, HEAD_DIM).transpose(1, 2)
        k = k.view(N, TOKENS, HEADS, HEAD_DIM).transpose(1, 2)
        v = v.view(N, TOKENS, HEADS, HEAD_DIM).transpose(1, 2)
        w = torch.matmul(q, k.transpose(-1, -2)) * scale
        w = F.softmax(w, dim=-1, dtype=torch.float32)
        o = torch.matmul(w, v).transpose(1, 2).reshape(N, TOKENS, HIDDEN)
        o = F.linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = r + o
        r = h
        y = F.layer_norm(h, (HIDDEN,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], LN_EPS)
        # CLIPMLP.forward (:346-350) with quick_gelu
        y = F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        y = y * torch.sigmoid(1.702 * y)
        y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        h = r + y
        if collect is not None:
            collect[f"layer{i}"] = h.clone()
    return h


@torch.no_grad()
def clip_embedding(sd, pixels: torch.Tensor, chunk: int = 8) -> torch.Tensor:
    """models/clip_embedder.py:63-65 -- mean over the 577 tokens of last_hidden_state. (N,1024) fp32."""
    outs = []
    for s in range(0, pixels.shape[0], chunk):
        outs.append(vit_last_hidden_state(sd, pixels[s:s + chunk]).mean(dim=1))
    return torch.cat(outs, dim=0)


@torch.no_grad()
def super_guessr_forward(cell_weight: torch.Tensor, cell_bias: torch.Tensor, lla_geocells: torch.Tensor,
                         num_candidates: int, vit_sd=None, pixel_values: Optional[torch.Tensor] = None,
                         embedding: Optional[torch.Tensor] = None):
    """Inference branch of SuperGuessr.forward with panorama=True, hierarchical=False, heading=False,
    multi_task=False (the configuration evaluation/evaluate.py:42-44 builds).

    Returns dict(preds_LLH (B,2) f64, preds_geocell (B,) i64, topk TopK(values (B,k) f32, indices i64),
    embedding (B,4,1024) f32, logits (B,C) f32, probs (B,C) f32).
    """
    if pixel_values is not None:
        B = pixel_values.shape[0]
        px = pixel_values.reshape(B * 4, 3, 336, 336)                       # super_guessr.py:386-388
        emb = clip_embedding(vit_sd, px)                                     # :395-398
        embedding = emb.reshape(B, 4, -1)                                    # :404-405
    layer_input = embedding
    if layer_input.dim() == 3:
        output = layer_input.mean(dim=1)                                     # :437
    else:
        output = layer_input
    logits = F.linear(output, cell_weight, cell_bias)                        # :447
    probs = F.softmax(logits, dim=-1)                                        # :448
    preds = torch.argmax(probs, dim=-1)                                      # :454
    pred_llh = torch.index_select(lla_geocells, 0, preds)                    # :455
    topk = torch.topk(probs, num_candidates, dim=-1)                         # :459
    return dict(preds_LLH=pred_llh, preds_geocell=preds, topk=TopK(topk.values, topk.indices),
                embedding=embedding, logits=logits, probs=probs)


def haversine(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """preprocessing/geo_utils.py:40-55 -- great-circle km, inputs [lng,lat] degrees, R = 6378137 m."""
    rad = torch.tensor(6378137.0, dtype=torch.float64)
    x_rad, y_rad = torch.deg2rad(x), torch.deg2rad(y)
    delta = y_rad - x_rad
    a = torch.sin(delta[:, 1] / 2) ** 2 + torch.cos(x_rad[:, 1]) * torch.cos(y_rad[:, 1]) * torch.sin(delta[:, 0] / 2) ** 2
    c = 2 * torch.arcsin(torch.sqrt(a))
    return (rad * c) / 1000


@torch.no_grad()
def proto_refiner_forward(bank, embedding: torch.Tensor, initial_preds: torch.Tensor,
                          candidate_cells: torch.Tensor, candidate_probs: Optional[torch.Tensor],
                          topk: int = 5, temperature: float = 1.6, max_refinement: float = 1000,
                          return_debug: bool = False):
    """models/proto_refiner.py:121-231 restated over the CSR bank arrays (see SyntheticBank).

    Per sample i and candidate j < topk (loop :154-182):
      empty cell            -> score -100000, pred [0,0]                          (:168-174)
      logits = -cdist(protos_of_cell, emb)                                         (:176-177, :332-344)
      score  = max(logits); pred_id = argmax(logits)  (first max)                  (:180-181)
      count == 1 -> (lng,lat) of the prototype row                                 (:245-246)
      else       -> member with the LARGEST distance (argmax of +distance)         (:248-255)
    probs = exp(score/T)/sum(exp(score/T)) without max shift                       (:187-188,:355-357)
    final = c_probs[:topk]*probs; refined = argmax(final)                          (:191-193)
    veto: haversine(initial, refined_llh) > max_refinement -> final = c_probs[:topk] (:198-205)
    output row = top_preds[argmax(final)], cell = candidates[argmax(final)]        (:219-222)
    Returns (None, preds_LLH (B,2) f32, preds_geocell (B,) i64).
    """
    assert topk <= candidate_cells.size(1)
    if embedding.dim() == 3:
        embedding = embedding.mean(dim=1)                                          # :139-140
    if candidate_probs is None:
        candidate_probs = torch.zeros_like(candidate_cells)                        # :143-145
        candidate_probs[:, 0] = 1
    proto_emb = bank.proto_emb          # sliced first, converted after: the bank may be lazily materialised
    train_emb = bank.train_emb
    preds_llh, preds_cell, choices, dbg_scores = [], [], [], []
    T = torch.tensor(temperature, dtype=torch.float32)
    for i in range(embedding.shape[0]):
        emb = embedding[i]
        cands = candidate_cells[i]
        c_probs = candidate_probs[i]
        top_preds, top_dist = [], []
        for j in range(topk):
            cell = int(cands[j])
            s, e = int(bank.cell_off[cell]), int(bank.cell_off[cell + 1])
            if e == s:
                top_dist.append(-100000.0)
                top_preds.append([0.0, 0.0])
                continue
            logits = -torch.cdist(torch.as_tensor(proto_emb[s:e]), emb[None]).flatten()
            top_dist.append(torch.max(logits).item())
            pid = s + int(torch.argmax(logits))
            if int(bank.proto_count[pid]) == 1:
                lng, lat = float(bank.proto_lnglat[pid, 0]), float(bank.proto_lnglat[pid, 1])
            else:
                idx = torch.as_tensor(bank.member_idx[int(bank.member_off[pid]):int(bank.member_off[pid + 1])])
                d = torch.cdist(torch.as_tensor(train_emb[idx.numpy() if not torch.is_tensor(train_emb) else idx]), emb[None]).flatten()
                mi = int(idx[int(torch.argmax(d))])
                lng, lat = float(bank.train_lnglat[mi, 0]), float(bank.train_lnglat[mi, 1])
            top_preds.append([lng, lat])
        td = torch.tensor(top_dist)                                                # float32
        ex = torch.exp(td / T)
        probs = ex / torch.sum(ex, axis=0)
        final = c_probs[:topk] * probs
        refined = int(torch.argmax(final))
        refined_llh = torch.tensor(top_preds[refined]).unsqueeze(0)                # float32
        dist = haversine(initial_preds[i].unsqueeze(0), refined_llh)[0]
        if dist > max_refinement:
            final = c_probs[:topk]
        fid = int(torch.argmax(final))
        choices.append(fid)
        preds_llh.append(top_preds[fid])
        preds_cell.append(int(cands[fid]))
        dbg_scores.append(top_dist)
    out = (None, torch.tensor(preds_llh), torch.tensor(preds_cell))
    if return_debug:
        return out + (torch.tensor(choices), torch.tensor(dbg_scores))
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b||_2 / ||b||_2 over the whole tensor."""
    a = a.double(); b = b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel_err_rows(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double(); b = b.double()
    return float(((a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-30)).max())
