"""CLIP ViT-L/14-336 image embedder on the HIP kernels, behind the reference's call surface.

Mirrors reference models/clip_embedder.py (class CLIPEmbedding: same constructor arguments, `forward(image)`
accepts PIL image(s) or an (N,3,336,336) tensor and returns the (N,1024) mean-over-577-tokens embedding on
the device), with `transformers.CLIPVisionModel` replaced by `HipCLIPVisionModel`, a module whose forward
runs entirely inside libpigeon_hip.so.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Callable, Dict, Optional

import numpy as np
import torch
from torch import Tensor

from . import hip_ops, synthetic
from .config import CLIP_MODEL, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
from .utils import load_state_dict


class HipCLIPVisionModel(torch.nn.Module):
    """Stand-in for transformers.CLIPVisionModel on this path.

    * holds the fp32 parameters under the SAME state-dict names as transformers (flat 5.x layout; the 4.23.1
      `vision_model.` prefix is accepted on load), so the reference's name-wise checkpoint loading
      (models/utils.py:24-45, models/super_guessr.py:222-238) keeps working;
    * `forward(pixel_values=...)` returns an object with `.last_hidden_state` (N,577,1024), taken before
      post_layernorm exactly like the HF module the reference calls (models/clip_embedder.py:63);
    * `embed(pixel_values)` returns the token mean directly (what both reference call sites compute next,
      clip_embedder.py:64-65 / super_guessr.py:397-398) without materialising the hidden states.
    The packed bf16 device copy (hip_ops.VitEncoder) is built lazily and rebuilt after any weight load.
    """

    def __init__(self, state_dict: Optional[Dict[str, Tensor]] = None, layers: int = synthetic.LAYERS,
                 seed: Optional[int] = None, max_chunk: int = 0):
        super().__init__()
        if state_dict is None:
            if seed is None:
                raise ValueError("HipCLIPVisionModel needs a state_dict or an explicit seed for random init")
            state_dict = synthetic.make_vit_weights(seed=seed, layers=layers)
        sd = {}
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.is_floating_point():
                continue
            k = k[len("vision_model."):] if k.startswith("vision_model.") else k
            sd[k] = v.detach().to("cpu", torch.float32).clone()
        self._sd = sd
        self.config = SimpleNamespace(hidden_size=1024, _name_or_path=CLIP_MODEL, num_hidden_layers=layers)
        self._enc: Optional[hip_ops.VitEncoder] = None
        self._enc_device = None
        self._max_chunk = max_chunk
        # exact mode (`embed_precise`): the encoder additionally packs the split-fp16 weight copy (3x the 16-bit weight memory)
        self._precise = False                             # SuperGuessr(exact_top1) / enable_precise(True) switch it on
        self._dummy = torch.nn.Parameter(torch.zeros(1), requires_grad=False)   # lets .to()/is_cuda work

    # ---- nn.Module protocol over the plain weight dict ----
    def state_dict(self, *args, prefix: str = "", **kwargs):
        return {prefix + k: v for k, v in self._sd.items()}

    def load_state_dict(self, state_dict, strict: bool = True):
        load_state_dict(self, state_dict)
        return SimpleNamespace(missing_keys=[], unexpected_keys=[])

    def parameters(self, recurse: bool = True):
        yield self._dummy

    def _weights_changed(self):
        if self._enc is not None:
            self._enc.close()
        self._enc = None

    @property
    def base_model(self):
        return self                                      # HF: model.base_model is model for CLIPVisionModel

    def _encoder(self, device: torch.device) -> hip_ops.VitEncoder:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._enc is None or self._enc_device != idx:
            if self._enc is not None:
                self._enc.close()
            self._enc = hip_ops.VitEncoder(self._sd, device=idx, max_chunk=self._max_chunk, precise=self._precise)
            self._enc_device = idx
        return self._enc

    def enable_precise(self, on: bool = True):
        """Make `embed_precise` available (the packed encoder is rebuilt with the split-weight copy on next use)."""
        if bool(on) != self._precise:
            self._precise = bool(on)
            self._weights_changed()

    def embed_precise(self, pixel_values: Tensor, out: Optional[Tensor] = None) -> Tensor:
        """The exact mode of `embed`: near-fp32 arithmetic (pg_vit_forward_precise), ~3x the time per image at 40+ images, more below.
        `out` (N,1024) fp32: write there instead of a fresh tensor."""
        self.enable_precise(True)
        pixel_values = _to_device_pixels(pixel_values, self._dummy.device)
        return self._encoder(pixel_values.device).forward_precise(pixel_values, out=out)

    def embed(self, pixel_values: Tensor) -> Tensor:
        pixel_values = _to_device_pixels(pixel_values, self._dummy.device)
        return self._encoder(pixel_values.device).forward(pixel_values)

    def forward(self, pixel_values: Tensor = None, **kwargs):
        pixel_values = _to_device_pixels(pixel_values, self._dummy.device)
        emb, hid = self._encoder(pixel_values.device).forward(pixel_values, return_hidden=True)
        return SimpleNamespace(last_hidden_state=hid, pooler_output=None, token_mean=emb)


def load_pretrained_clip(name: Optional[str] = None) -> HipCLIPVisionModel:
    """`CLIPVisionModel.from_pretrained(CLIP_MODEL)` of the reference (models/clip_embedder.py:26, evaluation/evaluate.py:36), offline:
    `name` (default: env PIGEON_CLIP_MODEL, else config.CLIP_MODEL) is resolved by transformers with `local_files_only=True` -- a
    directory written by `save_pretrained`, or the hub id if it sits in the local HuggingFace cache -- and its vision-tower state
    dict is packed into a `HipCLIPVisionModel`.  A full CLIPModel checkpoint works as well (its `vision_model.*` keys are taken).
    Raises RuntimeError with the reason when neither is there (this container has no network and no cache)."""
    name = name or os.environ.get("PIGEON_CLIP_MODEL") or CLIP_MODEL
    try:
        from transformers import CLIPVisionModel
        hf = CLIPVisionModel.from_pretrained(name, local_files_only=True)
    except Exception as e:  # noqa  (OSError: not cached; ImportError: no transformers; ValueError: bad directory)
        raise RuntimeError(f"CLIPVisionModel.from_pretrained({name!r}, local_files_only=True) failed: {type(e).__name__}: {e}") from e
    cfg = hf.config
    geom = (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.image_size, cfg.patch_size)
    if geom != (1024, 4096, 16, 336, 14):
        raise RuntimeError(f"{name!r} is not a ViT-L/14-336 vision tower (hidden, mlp, heads, image, patch = {geom}); "
                           "the HIP encoder is built for that geometry only")
    m = HipCLIPVisionModel(hf.state_dict(), layers=cfg.num_hidden_layers)
    m.config._name_or_path = name
    return m


def _to_device_pixels(pixel_values: Tensor, device) -> Tensor:
    if not pixel_values.is_cuda:
        dev = device if (isinstance(device, torch.device) and device.type == "cuda") else torch.device("cuda")
        pixel_values = pixel_values.to(dev)
    if pixel_values.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        pixel_values = pixel_values.float()
    return pixel_values.contiguous()


def _resize_output_size(h: int, w: int, size: int = 336):
    """transformers 4.23.1 (reference env.yml:60) ImageFeatureExtractionMixin.resize with an int size and
    default_to_square=False: the shorter edge becomes `size`, the longer one int(size * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def clip_preprocess(images) -> Tensor:
    """CLIPProcessor restated for PIL inputs on the HOST (what `self.processor(images=...)` does at reference
    models/clip_embedder.py:52; this is DataLoader-worker work in the reference too): convert RGB, resize shorter edge
    to 336 (PIL BICUBIC), centre crop 336x336, float32 / 255.0, (x - mean) / std -> (N,3,336,336) fp32.
    `gpu_preprocess` below is the device path and produces bit-identical values."""
    from PIL import Image
    if not isinstance(images, (list, tuple)):
        images = [images]
    out = []
    mean = np.asarray(OPENAI_CLIP_MEAN).astype(np.float32)
    std = np.asarray(OPENAI_CLIP_STD).astype(np.float32)
    for im in images:
        im = im.convert("RGB")
        w, h = im.size
        nh, nw = _resize_output_size(h, w)
        if (nh, nw) != (h, w):
            im = im.resize((nw, nh), resample=Image.BICUBIC)
        left, top = (nw - 336) // 2, (nh - 336) // 2
        im = im.crop((left, top, left + 336, top + 336))
        a = np.asarray(im).astype(np.float32) / 255.0
        out.append(((a - mean) / std).transpose(2, 0, 1))
    return torch.from_numpy(np.ascontiguousarray(np.stack(out)))


# pg_prep handles by (in_h, in_w, device): each owns the resize coefficient tables of one input geometry.  Street-view panels come
# in a handful of sizes, photo collections (YFCC) in thousands: least-recently-used handles beyond the cap are destroyed.
_PREPROCESSORS: "OrderedDict" = None
_PREPROCESSORS_MAX = 64


def _preprocessor(in_h: int, in_w: int, device_index: int) -> "hip_ops.Preprocessor":
    global _PREPROCESSORS
    from collections import OrderedDict
    if _PREPROCESSORS is None:
        _PREPROCESSORS = OrderedDict()
    key = (int(in_h), int(in_w), int(device_index))
    p = _PREPROCESSORS.pop(key, None)
    if p is None:
        p = hip_ops.Preprocessor(key[0], key[1], device=key[2])
    _PREPROCESSORS[key] = p                                        # most recently used last
    while len(_PREPROCESSORS) > _PREPROCESSORS_MAX:
        old_key, old = _PREPROCESSORS.popitem(last=False)
        torch.cuda.synchronize(old_key[2])                         # ITS device: its tables may still be read by a queued kernel
        old.close()
    return p


# Pinned host staging buffers of gpu_preprocess's PIL / ndarray-list path, one per (count, image shape, device): [tensor, event of the
# last host-to-device copy that read it].  A buffer is rewritten only after that copy has run (the event is long past in practice).
_STAGING: Dict = {}
_STAGING_MAX = 8


def _staging(n: int, shape, dev) -> Tensor:
    key = (int(n), tuple(int(v) for v in shape), int(dev.index))
    ent = _STAGING.get(key)
    if ent is None:
        if len(_STAGING) >= _STAGING_MAX:
            _STAGING.pop(next(iter(_STAGING)))
        ent = _STAGING[key] = [torch.empty((key[0],) + key[1], dtype=torch.uint8).pin_memory(), None]
    elif ent[1] is not None:
        ent[1].synchronize()
    return ent[0]


def _staging_done(t: Tensor) -> None:
    for ent in _STAGING.values():
        if ent[0] is t:
            ent[1] = torch.cuda.Event()
            ent[1].record()
            return


def gpu_preprocess(images, device="cuda", out_dtype: torch.dtype = torch.float32) -> Tensor:
    """CLIP preprocessing on the GPU (pg_prep_forward): `images` is a uint8 RGB tensor / ndarray (N,H,W,3) or (H,W,3),
    or a list of PIL images / arrays (grouped by size; every size gets its own coefficient tables).  Returns the
    (N,3,336,336) pixel_values on `device`, bit-identical to `clip_preprocess` / the reference's CLIPProcessor."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if hasattr(images, "convert"):                                 # one PIL image, as `processor(images=img)` accepts it
        images = [images]
    if torch.is_tensor(images) or isinstance(images, np.ndarray):
        t = torch.as_tensor(images)
        if t.dim() == 3:
            t = t[None]
        groups = [(None, t)]
    else:
        # (a PIL image that is RGB already is not copied by `convert`: a serving request's four views are; round 6)
        arrs = [np.asarray(im if getattr(im, "mode", None) == "RGB" else im.convert("RGB")) if hasattr(im, "convert") else np.asarray(im)
                for im in images]
        by_size: Dict = {}
        for i, a in enumerate(arrs):
            by_size.setdefault(a.shape, []).append(i)
        groups = []
        for idx in by_size.values():
            a0 = arrs[idx[0]]
            if a0.dtype != np.uint8 or a0.ndim != 3:
                groups.append((idx, torch.from_numpy(np.stack([arrs[i] for i in idx]))))       # refused below with the usual message
                continue
            # the images of one size go straight into a PINNED staging buffer (one copy instead of np.stack + the driver's own staging
            # of a pageable source), from which the transfer below is asynchronous
            stage = _staging(len(idx), a0.shape, dev)
            view = stage.numpy()
            for j, i in enumerate(idx):
                view[j] = arrs[i]
            groups.append((idx, stage))
    n_total = sum(g[1].shape[0] for g in groups)
    out = torch.empty((n_total, 3, 336, 336), dtype=out_dtype, device=dev)
    for idx, t in groups:
        if t.dtype != torch.uint8 or t.shape[-1] != 3:
            raise ValueError(f"gpu_preprocess expects uint8 RGB (N,H,W,3), got {t.dtype} {tuple(t.shape)}")
        with torch.cuda.device(dev):
            on_dev = t.to(dev, non_blocking=True).contiguous()
            _staging_done(t)                                       # (a pinned staging buffer is reused only after this copy has run)
            px = _preprocessor(t.shape[1], t.shape[2], dev.index)(on_dev, out_dtype)
        if idx is None:
            out = px
        else:
            out[torch.as_tensor(idx, device=dev)] = px
    return out


class CLIPEmbedding(torch.nn.Module):
    def __init__(self, model_name: str, device: str = 'cuda', load_checkpoint: bool = False,
                 panorama: bool = False, state_dict: Optional[Dict[str, Tensor]] = None,
                 clip_model: Optional[HipCLIPVisionModel] = None, contract_guard: Optional[str] = None):
        """CLIP embedding model (not trainable) -- reference models/clip_embedder.py:11-40.

        Args follow the reference.  The reference pulls the base weights from the HuggingFace hub
        (`CLIPVisionModel.from_pretrained(CLIP_MODEL)`, :26); here the same call runs with `local_files_only=True`
        (`load_pretrained_clip`: the local HF cache, or a `save_pretrained` directory named by env PIGEON_CLIP_MODEL), and two extra
        keyword arguments can supply the weights instead: `state_dict` (transformers CLIPVisionModel names) or a ready
        `clip_model`.  With `load_checkpoint=True`, `model_name` is a torch checkpoint copied over the weights by
        name with the leading `base_model.` component stripped, exactly as :30-32 does.

        `contract_guard` (round 6; default env PIGEON_EMBED_GUARD, else 'exact'): what to do when THIS set of weights puts the 16-bit
        encoder outside the embedding contract (1e-3 relative per image against the fp32 reference).  The first forward sends up to 32
        of its images through the fast and the exact encoder (pg_vit_forward / pg_vit_forward_precise), prints the measured per-image
        error and -- above 0.85e-3 overall or 0.95e-3 on the worst image, the rule `SuperGuessr` applies -- 'exact': encodes everything
        in the exact mode from then on (~3x the time per image), 'raise': raises, 'off': no check.  The embeddings `run.py embed`
        writes are what prototype banks are built from: a tower with large attention logits must not write them out of tolerance
        silently.
        """
        super().__init__()
        self.device = device
        self.processor = clip_preprocess
        if clip_model is not None:
            self.clip_model = clip_model
        elif state_dict is not None:
            self.clip_model = HipCLIPVisionModel(state_dict)
        else:
            try:                                                     # the reference's way (:25-26), from local files only
                self.clip_model = load_pretrained_clip()
            except RuntimeError as why:
                if load_checkpoint and os.path.exists(model_name):   # no base weights here, but the checkpoint carries the tower
                    ckpt = torch.load(model_name, map_location='cpu')
                    ckpt = {('.'.join(k.split('.')[1:]) if 'base_model' in k else k): v for k, v in ckpt.items()}
                    self.clip_model = HipCLIPVisionModel(ckpt)
                    load_checkpoint = False
                else:
                    raise RuntimeError(
                        "CLIPEmbedding: no weights. The reference downloads openai/clip-vit-large-patch14-336 from the hub; offline, "
                        "point PIGEON_CLIP_MODEL at a directory written by save_pretrained (or have the hub id in the local HF cache), "
                        "pass state_dict=... / clip_model=..., or a checkpoint path with load_checkpoint=True.  " + str(why)) from why
        self.panorama = panorama
        self.contract_guard = (contract_guard or os.environ.get('PIGEON_EMBED_GUARD', 'exact')).lower()
        if self.contract_guard not in ('exact', 'raise', 'off'):
            raise ValueError(f"contract_guard must be 'exact', 'raise' or 'off', got {self.contract_guard!r}")
        self.guard_stats = None                  # what the first forward measured (None until then / with the guard off)
        self.force_exact = False
        self.debias = os.environ.get('PIGEON_DEBIAS', '1') not in ('', '0')
        self.bias = None                         # (1024,) fp32: the systematic part `_check_contract` measured, subtracted from every fast embedding

        if load_checkpoint:
            sd = torch.load(model_name, map_location='cpu')
            load_state_dict(self.clip_model.base_model, sd, embedder=True)
            print('Loaded embedder from checkpoint:', model_name)

        if type(device) == str:
            self.clip_model = self.clip_model.to(self.device)
        else:
            self.clip_model = self.clip_model.cuda(self.device)
        self.eval()

    def _get_embedding(self, image) -> Tensor:
        """reference models/clip_embedder.py:42-66"""
        with torch.no_grad():
            if isinstance(image, Tensor) == False or image.dtype == torch.uint8:
                # PIL image(s) / uint8 HWC arrays: resize + crop + normalise on the GPU (bit-identical to the
                # reference's host-side CLIPProcessor); 16-bit pixels when the encoder's MFMA operands are fp16
                dev = self.device if type(self.device) == str else f'cuda:{self.device}'
                pixel_values = gpu_preprocess(image, dev, self._pixel_dtype())
            else:
                pixel_values = image
            if type(self.device) == str:
                pixel_values = pixel_values.to(self.device)
            else:
                pixel_values = pixel_values.cuda(self.device)
            base = self.clip_model.base_model
            if self.contract_guard != 'off' and self.guard_stats is None and pixel_values.shape[0] > 0:
                self._check_contract(base, pixel_values)
            # last_hidden_state.mean(dim=1), fused in the library (token_mean kernel)
            if self.force_exact:
                return base.embed_precise(pixel_values)
            emb = base.embed(pixel_values)
            if self.bias is not None:
                # the 16-bit encoder's measured systematic error taken out of every image's embedding (pigeon_amd/certainty.py `debias`)
                hip_ops.embedding_debias(emb, self.bias if self.bias.device == emb.device else self.bias.to(emb.device))
            return emb

    def _check_contract(self, base, pixel_values: Tensor, max_images: int = 32, contract: float = 1e-3) -> dict:
        """Once per set of weights: the 16-bit encoder against the exact one on up to `max_images` images of the first batch.
        (A first batch of fewer than 8 images measures the error but fits no bias: nothing is subtracted for the lifetime of this
        object -- hand the embed path a full first batch, as `run.py embed` does, or set PIGEON_DEBIAS=0 on both entry points.)"""
        px = _to_device_pixels(pixel_values[:max_images], base._dummy.device)
        fast, exact = base.embed(px), base.embed_precise(px)
        err = (fast - exact).norm(dim=1) / exact.norm(dim=1).clamp_min(1e-30)
        st = {'images': int(px.shape[0]), 'image_rel_err': float((fast - exact).norm() / exact.norm().clamp_min(1e-30)),
              'worst_image_rel_err': float(err.max()), 'contract': contract}
        # one verdict for a data-parallel job (`run.py embed` under torchrun): every rank measures its own first batch; all ranks act on
        # the worst measurement, so that the files they write together do not mix encoders
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            every = [None] * dist.get_world_size()
            dist.all_gather_object(every, (st['image_rel_err'], st['worst_image_rel_err']))
            st['image_rel_err_all_ranks'] = max(e[0] for e in every)
            st['worst_image_rel_err_all_ranks'] = max(e[1] for e in every)
        # the part of that error that is the SAME for every image (relative to |e|): with PIGEON_DEBIAS (default on) it is subtracted from
        # every fast embedding this module returns (pg_embedding_debias), when a held-out half of the images confirms that what is left
        # is smaller.  The verdict below is taken on the RAW error (the conservative reading).
        bias = None
        if self.debias and px.shape[0] >= 8:
            rel = (fast - exact) / exact.norm(dim=1, keepdim=True).clamp_min(1e-30)
            from .certainty import Certainty
            held = Certainty.apply_bias(fast[1::2], rel[0::2].mean(dim=0))
            left = float(((held - exact[1::2]).norm(dim=1) / exact[1::2].norm(dim=1).clamp_min(1e-30)).pow(2).mean().sqrt())
            raw = float(err.pow(2).mean().sqrt())
            st['bias_norm'], st['residual_rms'] = float(rel.mean(dim=0).norm()), left
            if left < 0.9 * raw:
                bias = rel.mean(dim=0).contiguous()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            every = [None] * dist.get_world_size()                   # one vector for the whole job: the mean of the ranks', if all keep one
            dist.all_gather_object(every, None if bias is None else bias.cpu())
            bias = None if any(b is None for b in every) else torch.stack(every).mean(dim=0).to(fast.device)
        self.bias = bias
        st['debias'] = bias is not None
        overall = st.get('image_rel_err_all_ranks', st['image_rel_err'])
        worst = st.get('worst_image_rel_err_all_ranks', st['worst_image_rel_err'])
        st['outside'] = overall > 0.85 * contract or worst > 0.95 * contract
        self.guard_stats = st
        print(f"pigeon_amd.CLIPEmbedding: 16-bit encoder vs exact encoder on {st['images']} images of the first batch: "
              f"{st['image_rel_err']:.2e} relative overall, worst image {st['worst_image_rel_err']:.2e} (contract {contract:g})"
              + (f"; its systematic part ({st['bias_norm']:.2e}) is subtracted from every embedding, {st['residual_rms']:.2e} left" if bias is not None else ""))
        if st['outside']:
            if self.contract_guard == 'raise':
                raise RuntimeError('CLIPEmbedding: the 16-bit encoder is outside the 1e-3 embedding contract on these weights '
                                   f"({st['image_rel_err']:.2e} overall, worst image {st['worst_image_rel_err']:.2e}); construct with "
                                   "contract_guard='exact' to encode in the exact mode")
            self.force_exact = True
            print('pigeon_amd.CLIPEmbedding: outside the embedding contract on these weights -- every image will be encoded in the '
                  'exact mode (about 3x the time per image).')
        return st

    def _pixel_dtype(self) -> torch.dtype:
        mma = os.environ.get("PIGEON_MMA_DTYPE", "f16").lower()
        return torch.float16 if mma != "bf16" else torch.float32

    def _pre_embed_hook(self) -> Callable:
        def hook(model, input, output):
            self.pre_embed_outputs = output[0]
        return hook

    def forward(self, image) -> Tensor:
        """reference models/clip_embedder.py:79-89"""
        return self._get_embedding(image)
