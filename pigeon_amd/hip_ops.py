"""Torch-tensor front end of the C ABI: tensors are only containers (device memory + current stream);
every FLOP of the hot path happens inside libpigeon_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import Bank, VitCfg, check, load

TOKENS, HIDDEN, MLP, PATCHES, KPAD = 577, 1024, 4096, 576, 640


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_T16 = {torch.float16: _lib.PG_DTYPE_F16, torch.bfloat16: _lib.PG_DTYPE_BF16}
_PIXDT = {torch.float32: _lib.PG_DTYPE_F32, torch.float16: _lib.PG_DTYPE_F16, torch.bfloat16: _lib.PG_DTYPE_BF16}
_PG2T = {_lib.PG_DTYPE_F16: torch.float16, _lib.PG_DTYPE_BF16: torch.bfloat16, _lib.PG_DTYPE_F32: torch.float32}


def _dt16(t: torch.Tensor) -> int:
    if t.dtype not in _T16:
        raise _lib.PigeonHipError(f"expected a float16 or bfloat16 tensor, got {t.dtype}")
    return _T16[t.dtype]


def _dev(t: torch.Tensor, dtype=None) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.PigeonHipError("expected a device tensor")
    if dtype is not None and t.dtype != dtype:
        raise _lib.PigeonHipError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.PigeonHipError("expected a contiguous tensor")
    return t


def _shape(t: torch.Tensor, name: str, *dims):
    """The C ABI takes pointers plus a few sizes; every other extent is implied.  Refuse a tensor whose shape is not the implied
    one (None = any extent) instead of letting a kernel read past its end."""
    if t.dim() != len(dims) or any(d is not None and int(s) != int(d) for s, d in zip(t.shape, dims)):
        want = "(" + ",".join("*" if d is None else str(int(d)) for d in dims) + ")"
        raise _lib.PigeonHipError(f"{name} must have shape {want}, got {tuple(t.shape)}")
    return t


# ----------------------------------------------------------------------------------------- building blocks
def gemm16(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epi: int,
           qscale: float = 1.0, qcols: int = 0, aux: Optional[torch.Tensor] = None, variant: int = 0,
           M: Optional[int] = None):
    """out (epilogue-dependent) <- A[M,K] x W[N,K]^T, both fp16 or both bf16; see pg_op_gemm16."""
    for t in (A, W, out):                                  # row-strided views are fine (padded leading dimensions)
        if not t.is_cuda or t.stride(1) != 1:
            raise _lib.PigeonHipError("gemm16 expects device tensors with unit inner stride")
    if W.dtype != A.dtype:
        raise _lib.PigeonHipError("gemm16: A and W must have the same 16-bit dtype")
    M = A.shape[0] if M is None else M
    K = A.shape[1]
    N = W.shape[0]
    dst = out
    if epi == _lib.EPI_RESID:
        # the residual epilogue's last row tile reads up to 383 rows past row M of `out` (see gemm16_resid_stat / pigeon_hip.h)
        have = out.untyped_storage().nbytes() - out.storage_offset() * out.element_size()
        if have < (M + 384) * out.stride(0) * out.element_size():
            dst = torch.zeros((max(M, out.shape[0]) + 384, out.shape[1]), dtype=out.dtype, device=out.device)[:out.shape[0]]
            dst.copy_(out)
    check(load().pg_op_gemm16_ld(_dt16(A), _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(dst), dst.stride(0), M, N, K,
                                 epi, float(qscale), int(qcols), _p(aux), variant, _stream()), "pg_op_gemm16_ld")
    if dst is not out:
        out.copy_(dst)
    return out


def tune_gemm_tail_rows(rows: int) -> None:
    """pg_tune_gemm_tail_rows: most rows pg_gemm_launch hands to the small-tile tail kernel (0 = never split).  Timing only."""
    check(load().pg_tune_gemm_tail_rows(int(rows)), "pg_tune_gemm_tail_rows")


def tune_gemm_tail_shape(min_k: int, min_n: int) -> None:
    """pg_tune_gemm_tail_shape: the tail split is used for GEMMs with K >= min_k or N >= min_n ((0, 0) = all).  Timing only."""
    check(load().pg_tune_gemm_tail_shape(int(min_k), int(min_n)), "pg_tune_gemm_tail_shape")


def rowstat_cast(x: torch.Tensor, out_dtype: torch.dtype = torch.float16, eps: float = 1e-5):
    """x fp32 (rows,1024) -> (16-bit copy, rowstat (rows,2) = (rstd, mean*rstd))."""
    _dev(x, torch.float32)
    rows = x.numel() // HIDDEN
    x16 = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    rs = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    check(load().pg_op_rowstat_cast(_p(x), _p(x16), _dt16(x16), _p(rs), rows, float(eps), _stream()), "pg_op_rowstat_cast")
    return x16, rs


def gemm16_resid_stat(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, X: torch.Tensor, variant: int = 0):
    """X += A.W^T + bias in place; returns (x16 copy of the new X, statpart (N/64, M, 2)).

    The persistent kernels fetch the residual rows of a whole 256- / 384-row tile through a buffer descriptor whose row term rides in
    the SGPR offset, which the hardware bounds check does not cover: in the last (partial) tile they READ up to 383 rows past row M of
    X (never write there; the values are discarded).  Inside the encoder those rows are the workspace's next buffer.  Here X must own
    that slack -- if its storage ends earlier the GEMM runs on a padded copy (this wrapper serves tests and tools, not the hot path)."""
    M, K = A.shape
    N = W.shape[0]
    x16 = torch.empty((M, N), dtype=A.dtype, device=A.device)
    part = torch.empty((N // 64, M, 2), dtype=torch.float32, device=A.device)
    have = X.untyped_storage().nbytes() - X.storage_offset() * X.element_size()
    Xr = X
    if have < (M + 384) * X.stride(0) * X.element_size():
        Xr = torch.zeros((M + 384, N), dtype=X.dtype, device=X.device)[:M]
        Xr.copy_(X)
    check(load().pg_op_gemm16_resid_stat(_dt16(A), _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(Xr), Xr.stride(0), _p(x16),
                                         x16.stride(0), _p(part), M, N, K, variant, _stream()), "pg_op_gemm16_resid_stat")
    if Xr is not X:
        X.copy_(Xr)
    return x16, part


def rowstat_finalize(part: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    slots, rows = part.shape[0], part.shape[1]
    rs = torch.empty((rows, 2), dtype=torch.float32, device=part.device)
    check(load().pg_op_rowstat_finalize(_p(part), slots, _p(rs), rows, float(eps), _stream()), "pg_op_rowstat_finalize")
    return rs


def gemm16_ln(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, colsum: torch.Tensor, rowstat: torch.Tensor, epi: int,
              qscale: float = 1.0, qcols: int = 0, variant: int = 0) -> torch.Tensor:
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    check(load().pg_op_gemm16_ln(_dt16(A), _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(colsum), _p(rowstat), _p(out),
                                 out.stride(0), M, N, K, epi, float(qscale), int(qcols), variant, _stream()), "pg_op_gemm16_ln")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out_dtype: torch.dtype = torch.float16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(x, torch.float32)
    rows = x.numel() // HIDDEN
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    pg = _lib.PG_DTYPE_F32 if out.dtype == torch.float32 else _dt16(out)
    check(load().pg_op_layernorm(_p(x), _p(gamma), _p(beta), _p(out), pg, rows, float(eps), _stream()), "pg_op_layernorm")
    return out


def attention(qkv: torch.Tensor, n_images: int) -> torch.Tensor:
    _dev(qkv)
    out = torch.empty((n_images * TOKENS, HIDDEN), dtype=qkv.dtype, device=qkv.device)
    check(load().pg_op_attention(_dt16(qkv), _p(qkv), _p(out), n_images, _stream()), "pg_op_attention")
    return out


def im2col(pixels: torch.Tensor, out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    _dev(pixels)
    n = pixels.shape[0]
    dt = _PIXDT[pixels.dtype]
    out = torch.empty((n * PATCHES, KPAD), dtype=out_dtype, device=pixels.device)
    check(load().pg_op_im2col(_p(pixels), dt, _p(out), _dt16(out), n, _stream()), "pg_op_im2col")
    return out


def token_mean(x: torch.Tensor) -> torch.Tensor:
    _dev(x, torch.float32)
    n = x.shape[0]
    out = torch.empty((n, HIDDEN), dtype=torch.float32, device=x.device)
    check(load().pg_op_token_mean(_p(x), _p(out), n, _stream()), "pg_op_token_mean")
    return out


def cast_f32(x: torch.Tensor, out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    _dev(x, torch.float32)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(load().pg_op_cast_f32(_p(x), _p(y), _dt16(y), x.numel(), _stream()), "pg_op_cast_f32")
    return y


# ----------------------------------------------------------------------------------------- around the hot path
def proto_build(train_emb: torch.Tensor, member_off: torch.Tensor, member_idx: torch.Tensor) -> torch.Tensor:
    """(Ntr,1024) or (Ntr,4,1024) fp32 training embeddings + CSR member lists -> (P,1024) fp32 prototype means."""
    _dev(train_emb, torch.float32); _dev(member_off, torch.int64); _dev(member_idx, torch.int64)
    if train_emb.dim() not in (2, 3) or train_emb.shape[-1] != HIDDEN:
        raise _lib.PigeonHipError(f"proto_build: embeddings must be (Ntr,{HIDDEN}) or (Ntr,panels,{HIDDEN}), got {tuple(train_emb.shape)}")
    panels = 1 if train_emb.dim() == 2 else int(train_emb.shape[1])
    if member_off.dim() != 1 or member_off.numel() < 1 or member_idx.dim() != 1:
        raise _lib.PigeonHipError("proto_build: member_off (P+1,) and member_idx (n,) expected")
    P = member_off.numel() - 1
    # one-off bank construction: a host round trip for the CSR invariants is cheap, a member index past the training bank is a
    # read out of bounds in the kernel
    off = member_off.cpu()
    if int(off[0]) != 0 or int(off[-1]) != member_idx.numel() or bool((off[1:] < off[:-1]).any()):
        raise _lib.PigeonHipError("proto_build: member_off must rise from 0 to len(member_idx)")
    if member_idx.numel() and (int(member_idx.min()) < 0 or int(member_idx.max()) >= train_emb.shape[0]):
        raise _lib.PigeonHipError(f"proto_build: member index outside the {train_emb.shape[0]} training rows")
    out = torch.empty((P, HIDDEN), dtype=torch.float32, device=train_emb.device)
    check(load().pg_proto_build(_p(train_emb), panels, train_emb.shape[0], _p(member_off), _p(member_idx), P, _p(out),
                                _stream()), "pg_proto_build")
    return out


def haversine_matrix(x: torch.Tensor, y_rows: torch.Tensor) -> torch.Tensor:
    """x (N,2) fp32/fp64 [lng,lat], y_rows (M,2) fp64 -> (N,M) fp64 km."""
    _dev(x); _dev(y_rows, torch.float64)
    if x.dtype not in (torch.float32, torch.float64):
        raise _lib.PigeonHipError("haversine_matrix: x must be fp32 or fp64")
    _shape(x, "x", None, 2); _shape(y_rows, "y_rows", None, 2)
    N, M = x.shape[0], y_rows.shape[0]
    out = torch.empty((N, M), dtype=torch.float64, device=x.device)
    check(load().pg_haversine_matrix(_p(x), _lib.PG_DTYPE_F64 if x.dtype == torch.float64 else _lib.PG_DTYPE_F32, _p(y_rows),
                                     N, M, _p(out), _stream()), "pg_haversine_matrix")
    return out


def haversine_pairs(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x (N,2) fp64, y (N,2) fp32/fp64 [lng,lat] degrees -> (N,) fp64 km (row-paired)."""
    _dev(x, torch.float64); _dev(y)
    if y.dtype not in (torch.float32, torch.float64) or x.shape != y.shape or x.dim() != 2 or x.shape[1] != 2:
        raise _lib.PigeonHipError("haversine_pairs: x (N,2) fp64 and y (N,2) fp32/fp64 expected")
    out = torch.empty((x.shape[0],), dtype=torch.float64, device=x.device)
    check(load().pg_haversine_pairs(_p(x), _p(y), _lib.PG_DTYPE_F64 if y.dtype == torch.float64 else _lib.PG_DTYPE_F32,
                                    x.shape[0], _p(out), _stream()), "pg_haversine_pairs")
    return out


def smooth_labels(distances: torch.Tensor, constant: float) -> torch.Tensor:
    _dev(distances, torch.float64)
    _shape(distances, "distances", None, None)
    N, M = distances.shape
    out = torch.empty_like(distances)
    check(load().pg_smooth_labels(_p(distances), N, M, float(constant), _p(out), _stream()), "pg_smooth_labels")
    return out


# ----------------------------------------------------------------------------------------- image preprocessing
class Preprocessor:
    """CLIP preprocessing on the GPU for one input geometry: (N,H,W,3) uint8 RGB -> (N,3,336,336) pixel_values,
    bit-exact with `CLIPProcessor(images=pil)` (Pillow fixed-point bicubic + numpy float32 normalisation)."""

    def __init__(self, in_h: int, in_w: int, device: int = 0):
        _lib.require_gpu()
        self._h = C.c_void_p()
        check(load().pg_prep_create(C.byref(self._h), int(device), int(in_h), int(in_w)), "pg_prep_create")
        self.in_h, self.in_w, self.device = int(in_h), int(in_w), int(device)
        geo = (C.c_int32 * 6)()
        check(load().pg_prep_geometry(self._h, geo), "pg_prep_geometry")
        self.resized_h, self.resized_w, self.top, self.left, self.row0, self.nrows = [int(x) for x in geo]
        self._ws = None

    def forward(self, images_u8: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        _dev(images_u8, torch.uint8)
        if images_u8.dim() != 4 or tuple(images_u8.shape[1:]) != (self.in_h, self.in_w, 3):
            raise _lib.PigeonHipError(f"images must be (N,{self.in_h},{self.in_w},3) uint8, got {tuple(images_u8.shape)}")
        if out_dtype not in (torch.float32, torch.float16):
            raise _lib.PigeonHipError("preprocess output dtype must be float32 or float16")
        n = images_u8.shape[0]
        need = C.c_size_t()
        check(load().pg_prep_workspace_bytes(self._h, n, C.byref(need)), "pg_prep_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != images_u8.device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=images_u8.device)
        out = torch.empty((n, 3, 336, 336), dtype=out_dtype, device=images_u8.device)
        check(load().pg_prep_forward(self._h, _p(images_u8), n, _p(out), _PIXDT[out_dtype], _p(self._ws), self._ws.numel(),
                                     _stream()), "pg_prep_forward")
        return out

    __call__ = forward

    def close(self):
        if self._h:
            load().pg_prep_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------- ViT encoder handle
class VitEncoder:
    """Owns a pg_vit handle: bf16-packed weights resident in HBM, forward = ViT-L/14-336 + token mean.

    state_dict keys may be in either transformers layout (vision_model.* or flat); values are CPU or device
    tensors of any float dtype (converted to fp32 on the host, then packed by the library).
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor], device: int = 0, max_chunk: int = 0,
                 layers: Optional[int] = None, mma_dtype: Optional[str] = None, precise: bool = False):
        """mma_dtype: None (library default: fp16, or env PIGEON_MMA_DTYPE=bf16), 'f16' or 'bf16'.
        precise: also pack the split-fp16 weight copy of the exact mode (`forward_precise`; 3x the 16-bit weight memory)."""
        _lib.require_gpu()
        lib = load()
        keys = [k[len("vision_model."):] if k.startswith("vision_model.") else k for k in state_dict]
        if layers is None:
            layers = 0
            while f"encoder.layers.{layers}.layer_norm1.weight" in keys:
                layers += 1
        if layers < 1:
            raise _lib.PigeonHipError("state dict holds no encoder layers")
        self.layers = layers
        self.device = device
        pgdt = {None: 0, "f16": _lib.PG_DTYPE_F16, "fp16": _lib.PG_DTYPE_F16, "bf16": _lib.PG_DTYPE_BF16}[mma_dtype]
        cfg = VitCfg(layers, 336, 14, 1024, 16, 4096, 1e-5, max_chunk, pgdt, 1 if precise else 0)
        self.precise = bool(precise)
        self._ws_precise = None
        h = C.c_void_p()
        torch.cuda.set_device(device)
        check(lib.pg_vit_create(C.byref(h), device, C.byref(cfg)), "pg_vit_create")
        self._h = h
        for name, t in state_dict.items():
            if not torch.is_tensor(t) or not t.is_floating_point():
                continue                                   # e.g. embeddings.position_ids (int64 buffer)
            a = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(a.dim(), 1))(*(list(a.shape) or [1]))
            check(lib.pg_vit_load_weight(h, name.encode(), C.c_void_p(a.data_ptr()), _lib.PG_DTYPE_F32, shape,
                                         max(a.dim(), 1)), f"pg_vit_load_weight({name})")
        check(lib.pg_vit_finalize(h), "pg_vit_finalize")
        self.mma_dtype = {_lib.PG_DTYPE_F16: "f16", _lib.PG_DTYPE_BF16: "bf16"}[lib.pg_vit_mma_dtype(h)]
        self._ws = None
        self.max_chunk = max_chunk if max_chunk > 0 else 512

    def _workspace(self, n: int) -> torch.Tensor:
        need = C.c_size_t()
        check(load().pg_vit_workspace_bytes(self._h, n, C.byref(need)), "pg_vit_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value + 256, dtype=torch.uint8, device=f"cuda:{self.device}")
        return self._ws

    def forward(self, pixels: torch.Tensor, return_hidden: bool = False):
        """pixels (N,3,336,336) fp32/bf16 on the device -> (N,1024) fp32 [, (N,577,1024) fp32]."""
        _dev(pixels)
        if pixels.dim() != 4 or tuple(pixels.shape[1:]) != (3, 336, 336):
            raise _lib.PigeonHipError(f"pixels must be (N,3,336,336), got {tuple(pixels.shape)}")
        if pixels.dtype not in _PIXDT:
            raise _lib.PigeonHipError("pixels must be fp32, fp16 or bf16")
        n = pixels.shape[0]
        ws = self._workspace(n)
        off = (-ws.data_ptr()) % 256
        emb = torch.empty((n, HIDDEN), dtype=torch.float32, device=pixels.device)
        hid = torch.empty((n, TOKENS, HIDDEN), dtype=torch.float32, device=pixels.device) if return_hidden else None
        dt = _PIXDT[pixels.dtype]
        check(load().pg_vit_forward_hidden(self._h, _p(pixels), dt, n, _p(emb), _p(hid),
                                           C.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream()),
              "pg_vit_forward")
        return (emb, hid) if return_hidden else emb

    __call__ = forward

    def forward_precise(self, pixels: torch.Tensor, return_hidden: bool = False, out: Optional[torch.Tensor] = None):
        """The exact mode (pg_vit_forward_precise): same contract as `forward`, near-fp32 arithmetic (split-fp16 GEMM operands,
        fp32 attention / LayerNorm / QuickGELU), ~5x the time per image.  Needs `precise=True` at construction."""
        if not self.precise:
            raise _lib.PigeonHipError("forward_precise: the encoder was built without precise=True (no split-weight copy)")
        _dev(pixels)
        if pixels.dim() != 4 or tuple(pixels.shape[1:]) != (3, 336, 336):
            raise _lib.PigeonHipError(f"pixels must be (N,3,336,336), got {tuple(pixels.shape)}")
        if pixels.dtype not in _PIXDT:
            raise _lib.PigeonHipError("pixels must be fp32, fp16 or bf16")
        n = pixels.shape[0]
        need = C.c_size_t()
        check(load().pg_vit_precise_workspace_bytes(self._h, n, C.byref(need)), "pg_vit_precise_workspace_bytes")
        if self._ws_precise is None or self._ws_precise.numel() < need.value + 256:
            self._ws_precise = None
            self._ws_precise = torch.empty(need.value + 256, dtype=torch.uint8, device=f"cuda:{self.device}")
        ws = self._ws_precise
        off = (-ws.data_ptr()) % 256
        if out is not None:
            _dev(out, torch.float32); _shape(out, "out", n, HIDDEN)
        emb = out if out is not None else torch.empty((n, HIDDEN), dtype=torch.float32, device=pixels.device)
        hid = torch.empty((n, TOKENS, HIDDEN), dtype=torch.float32, device=pixels.device) if return_hidden else None
        check(load().pg_vit_forward_precise(self._h, _p(pixels), _PIXDT[pixels.dtype], n, _p(emb), _p(hid),
                                            C.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream()), "pg_vit_forward_precise")
        return (emb, hid) if return_hidden else emb

    def graph(self, on: Optional[bool] = None):
        """Switch the encoder's hipGraph replay on / off (None: query only) -> (replays, captures) since creation."""
        r, c = C.c_int64(), C.c_int64()
        check(load().pg_vit_graph(self._h, -1 if on is None else int(bool(on)), C.byref(r), C.byref(c)), "pg_vit_graph")
        return int(r.value), int(c.value)

    # ---- per-kernel-class timing for bench.py ----
    def profile_enable(self, on: bool = True, classes=None):
        """classes: None = every kernel class, else an iterable of class names (_lib.PROF_CLASSES) to bracket with events."""
        v = 1 if on else 0
        if on and classes is not None:
            v = 0
            for c in classes:
                v |= 1 << (_lib.PROF_CLASSES.index(c) + 1)
        check(load().pg_vit_profile_enable(self._h, v), "pg_vit_profile_enable")

    def profile_reset(self):
        check(load().pg_vit_profile_reset(self._h), "pg_vit_profile_reset")

    def profile_read(self):
        n = len(_lib.PROF_CLASSES)
        launches = (C.c_int64 * n)()
        ms = (C.c_double * n)()
        check(load().pg_vit_profile_read(self._h, launches, ms), "pg_vit_profile_read")
        return {name: (int(launches[i]), float(ms[i])) for i, name in enumerate(_lib.PROF_CLASSES)}

    # ---- fp16 saturation counter (debug) ----
    def saturation_check(self, on: bool = True):
        check(load().pg_vit_saturation_check(self._h, 1 if on else 0), "pg_vit_saturation_check")

    def saturation_read(self, reset: bool = True) -> int:
        """16-bit activations found sitting exactly on the fp16 limit +-65504 (= clamped conversions) since the last reset."""
        n = C.c_int64()
        check(load().pg_vit_saturation_read(self._h, C.byref(n), 1 if reset else 0), "pg_vit_saturation_read")
        return int(n.value)

    def range_alarm_read(self, reset: bool = True) -> int:
        """Rows of the residual stream whose sum of squares reached 65504^2 since the last reset (always-on, fp16 operands):
        0 means no 16-bit copy of a residual row can have been clamped.  Synchronises the device."""
        n = C.c_int64()
        check(load().pg_vit_range_alarm_read(self._h, C.byref(n), 1 if reset else 0), "pg_vit_range_alarm_read")
        return int(n.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().pg_vit_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------- head
def head_forward(emb: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, centroids: torch.Tensor, k: int):
    """emb (B,P,1024) or (B,1024) fp32; returns dict(logits, topk_values, topk_indices, preds_geocell, preds_LLH)."""
    _dev(emb, torch.float32); _dev(W, torch.float32); _dev(bias, torch.float32); _dev(centroids, torch.float64)
    if emb.dim() not in (2, 3):
        raise _lib.PigeonHipError(f"emb must be (B,{HIDDEN}) or (B,P,{HIDDEN}), got {tuple(emb.shape)}")
    B = emb.shape[0]
    P = emb.shape[1] if emb.dim() == 3 else 1
    _shape(emb, "emb", *((B, P, HIDDEN) if emb.dim() == 3 else (B, HIDDEN)))
    Cn = W.shape[0]
    _shape(W, "W", Cn, HIDDEN); _shape(bias, "bias", Cn); _shape(centroids, "centroids", Cn, 2)
    if not 1 <= int(k) <= Cn:
        raise _lib.PigeonHipError(f"head: k = {k} candidates of {Cn} geocells")
    dev = emb.device
    logits = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    tv = torch.empty((B, k), dtype=torch.float32, device=dev)
    ti = torch.empty((B, k), dtype=torch.int64, device=dev)
    am = torch.empty((B,), dtype=torch.int64, device=dev)
    llh = torch.empty((B, 2), dtype=torch.float64, device=dev)
    check(load().pg_head_forward(_p(emb), B, P, _p(W), _p(bias), _p(centroids), Cn, k, _p(logits), _p(tv), _p(ti),
                                 _p(am), _p(llh), _stream()), "pg_head_forward")
    return dict(logits=logits, topk_values=tv, topk_indices=ti, preds_geocell=am, preds_LLH=llh)


def head_margin(logits: torch.Tensor, emb: torch.Tensor, W: torch.Tensor):
    """Certainty inputs of the top-1 (pg_head_margin): logits (B,C) fp32 as written by head_forward, emb (B,P,1024) or (B,1024),
    W (C,1024).  Returns (margin (B,) fp32 = logit(top1) - logit(top2), sens (B,) fp32 = |mean_p emb| |W[top1] - W[top2]| / 32,
    top2 (B,) int64)."""
    _dev(logits, torch.float32); _dev(emb, torch.float32); _dev(W, torch.float32)
    B, Cn = logits.shape
    P = emb.shape[1] if emb.dim() == 3 else 1
    _shape(emb, "emb", *((B, P, HIDDEN) if emb.dim() == 3 else (B, HIDDEN)))
    _shape(W, "W", Cn, HIDDEN)
    margin = torch.empty((B,), dtype=torch.float32, device=logits.device)
    sens = torch.empty((B,), dtype=torch.float32, device=logits.device)
    top2 = torch.empty((B,), dtype=torch.int64, device=logits.device)
    check(load().pg_head_margin(_p(logits), B, Cn, _p(emb), P, _p(W), _p(margin), _p(sens), _p(top2), _stream()), "pg_head_margin")
    return margin, sens, top2


def head_certainty(logits: torch.Tensor, emb: torch.Tensor, W: torch.Tensor, topk_idx: torch.Tensor,
                   drift: Optional[torch.Tensor], wstats: torch.Tensor):
    """pg_head_certainty: tolerance of the top-1 against every cell.  logits (B,C) as written by head_forward, emb (B,P,1024) or
    (B,1024), topk_idx (B,kx) int64 from the same head_forward call, drift (1024,) fp32 or None, wstats (2,) fp32 = [largest row
    norm of W, max_c |W[c].drift| (0 without drift)].  Returns (tol (B,) f32, code (B,) i32, margin (B,) f32, sens (B,) f32)."""
    _dev(logits, torch.float32); _dev(emb, torch.float32); _dev(W, torch.float32); _dev(topk_idx, torch.int64)
    _dev(wstats, torch.float32)
    B, Cn = logits.shape
    P = emb.shape[1] if emb.dim() == 3 else 1
    _shape(emb, "emb", *((B, P, HIDDEN) if emb.dim() == 3 else (B, HIDDEN)))
    _shape(W, "W", Cn, HIDDEN); _shape(topk_idx, "topk_idx", B, None); _shape(wstats, "wstats", 2)
    kx = topk_idx.shape[1]
    if drift is not None:
        _dev(drift, torch.float32); _shape(drift, "drift", HIDDEN)
    dev = logits.device
    tol = torch.empty((B,), dtype=torch.float32, device=dev)
    code = torch.empty((B,), dtype=torch.int32, device=dev)
    margin = torch.empty((B,), dtype=torch.float32, device=dev)
    sens = torch.empty((B,), dtype=torch.float32, device=dev)
    check(load().pg_head_certainty(_p(logits), B, Cn, _p(emb), P, _p(W), _p(topk_idx), kx, _p(drift), _p(wstats), _p(tol), _p(code),
                                   _p(margin), _p(sens), _stream()), "pg_head_certainty")
    return tol, code, margin, sens


def embedding_debias(emb: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """pg_embedding_debias, in place: emb[r] -= |emb[r]| * bias for every row of the (n,1024) fp32 matrix `emb` (contiguous; any
    leading shape).  bias (1024,) fp32: the calibrated systematic part of the 16-bit encoder's error (pigeon_amd/certainty.py)."""
    _dev(emb, torch.float32); _dev(bias, torch.float32); _shape(bias, "bias", HIDDEN)
    if emb.dim() < 1 or emb.shape[-1] != HIDDEN or not emb.is_contiguous():
        raise ValueError(f"embedding_debias: emb must be contiguous (..., {HIDDEN}), got {tuple(emb.shape)}")
    n = emb.numel() // HIDDEN
    check(load().pg_embedding_debias(_p(emb), n, HIDDEN, _p(bias), _stream()), "pg_embedding_debias")
    return emb


# ----------------------------------------------------------------------------------------- deferred exact tier (csrc/requeue.hip)
def head_wstats(W: torch.Tensor, drift: Optional[torch.Tensor]) -> torch.Tensor:
    """pg_head_wstats: (2,) fp32 = [largest row norm of W, max over cells of |W[c].drift| (0 without drift)]."""
    _dev(W, torch.float32); _shape(W, "W", None, HIDDEN)
    if drift is not None:
        _dev(drift, torch.float32); _shape(drift, "drift", HIDDEN)
    out = torch.empty((2,), dtype=torch.float32, device=W.device)
    check(load().pg_head_wstats(_p(W), W.shape[0], _p(drift), _p(out), _stream()), "pg_head_wstats")
    return out


def requeue_append(head_tol: torch.Tensor, refine_tol: Optional[torch.Tensor], refine_code: Optional[torch.Tensor], thr: float,
                   force_all: bool = False, dst_base: int = 0, flushed: int = 0, cap: int = 0,
                   counters: Optional[torch.Tensor] = None, slot_dst: Optional[torch.Tensor] = None):
    """pg_requeue_append.  Returns (certain (B,) uint8, cause (B,) int32, row_slot (B,) int32 or None without a queue)."""
    _dev(head_tol, torch.float32)
    B = head_tol.numel()
    if refine_tol is not None:
        _dev(refine_tol, torch.float32); _shape(refine_tol, "refine_tol", B)
    if refine_code is not None:
        _dev(refine_code, torch.int32); _shape(refine_code, "refine_code", B)
    dev = head_tol.device
    certain = torch.empty((B,), dtype=torch.uint8, device=dev)
    cause = torch.empty((B,), dtype=torch.int32, device=dev)
    row_slot = None
    if cap > 0:
        _dev(counters, torch.int64); _shape(counters, "counters", 2)
        _dev(slot_dst, torch.int64); _shape(slot_dst, "slot_dst", cap)
        row_slot = torch.empty((B,), dtype=torch.int32, device=dev)
    check(load().pg_requeue_append(_p(head_tol), _p(refine_tol), _p(refine_code), B, float(thr), 1 if force_all else 0, int(dst_base),
                                   int(flushed), int(cap), _p(counters), _p(slot_dst), _p(row_slot), _p(certain), _p(cause), _stream()),
          "pg_requeue_append")
    return certain, cause, row_slot


def rows_to_slots(src: torch.Tensor, row_slot: torch.Tensor, dst: torch.Tensor) -> None:
    """pg_rows_to_slots: dst[row_slot[r]] = src[r] for the rows with a slot; src (B, ...) and dst (cap, ...) share the row shape."""
    _dev(src); _dev(dst); _dev(row_slot, torch.int32)
    if src.dtype != dst.dtype or tuple(src.shape[1:]) != tuple(dst.shape[1:]):
        raise _lib.PigeonHipError(f"rows_to_slots: rows of {src.dtype} {tuple(src.shape[1:])} into {dst.dtype} {tuple(dst.shape[1:])}")
    B = src.shape[0]
    _shape(row_slot, "row_slot", B)
    rb = (src.numel() // max(B, 1)) * src.element_size()
    check(load().pg_rows_to_slots(_p(src), rb, _p(row_slot), B, _p(dst), _stream()), "pg_rows_to_slots")


def requeue_take(slot_dst: torch.Tensor, head: int, n_valid: int, n_pad: int) -> torch.Tensor:
    """pg_requeue_take -> (n_pad,) int64 ring rows of the slots [head, head + n_pad) (mod cap), -1 beyond the first n_valid."""
    _dev(slot_dst, torch.int64)
    out = torch.empty((n_pad,), dtype=torch.int64, device=slot_dst.device)
    check(load().pg_requeue_take(_p(slot_dst), slot_dst.numel(), int(head), int(n_valid), int(n_pad), _p(out), _stream()), "pg_requeue_take")
    return out


def scatter_rows(src: torch.Tensor, dst_row: torch.Tensor, dst: torch.Tensor, remap=None) -> None:
    """pg_scatter_rows: dst[dst_row[i]] = src[i] (dst_row[i] < 0: skipped).  remap = (wb, b, off): dst_row addresses the gathered ring,
    dst is a local-only array (see include/pigeon_hip.h)."""
    _dev(src); _dev(dst); _dev(dst_row, torch.int64)
    n = src.shape[0]
    _shape(dst_row, "dst_row", n)
    if src.dtype != dst.dtype or tuple(src.shape[1:]) != tuple(dst.shape[1:]):
        raise _lib.PigeonHipError(f"scatter_rows: rows of {src.dtype} {tuple(src.shape[1:])} into {dst.dtype} {tuple(dst.shape[1:])}")
    rb = 1
    for d in src.shape[1:]:
        rb *= int(d)
    rb *= src.element_size()
    wb, b, off = remap if remap is not None else (0, 0, 0)
    check(load().pg_scatter_rows(_p(src), rb, _p(dst_row), n, _p(dst), dst.shape[0], int(wb), int(b), int(off), _stream()), "pg_scatter_rows")


# ----------------------------------------------------------------------------------------- exact-mode building blocks
def x3_split(x: torch.Tensor, gelu: bool = False) -> torch.Tensor:
    """fp32 (rows,cols) -> fp16 triple (rows, 3 cols) = [hi | lo | hi 2^-8]; gelu: through QuickGELU first."""
    _dev(x, torch.float32)
    rows, cols = x.shape
    y = torch.empty((rows, 3 * cols), dtype=torch.float16, device=x.device)
    check(load().pg_op_x3_split(_p(x), _p(y), rows, cols, 1 if gelu else 0, _stream()), "pg_op_x3_split")
    return y


def gemm16_parts(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], S: int) -> torch.Tensor:
    """pg_op_gemm16_parts: A (M, S*Kp), W (N, S*Kp) 16-bit -> (S, M, N) fp32, part p = A[:, p Kp:(p+1) Kp] x W[:, p Kp:(p+1) Kp]^T
    (+ bias for p = 0), all parts in one persistent launch."""
    _dev(A); _dev(W)
    if W.dtype != A.dtype or A.shape[1] != W.shape[1] or A.shape[1] % S:
        raise _lib.PigeonHipError("gemm16_parts: A and W must share dtype and a K' divisible by S")
    M, Kt = A.shape
    N = W.shape[0]
    out = torch.empty((S, M, N), dtype=torch.float32, device=A.device)
    check(load().pg_op_gemm16_parts(_dt16(A), _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(out), M, N, Kt // S, S, _stream()),
          "pg_op_gemm16_parts")
    return out


def x3_layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _dev(x, torch.float32)
    rows = x.numel() // HIDDEN
    y = torch.empty((rows, 3 * HIDDEN), dtype=torch.float16, device=x.device)
    check(load().pg_op_x3_layernorm(_p(x), _p(gamma), _p(beta), _p(y), rows, float(eps), _stream()), "pg_op_x3_layernorm")
    return y


def x3_pack_weight(W: torch.Tensor) -> torch.Tensor:
    """Host-side mirror of the library's weight triple (vit.hip pack_x3), for the building-block tests: W fp32 (N,K) ->
    fp16 (N,3K) = [Wh | Wh | (W - Wh) 2^8]."""
    Wh = W.to(torch.float16)
    Wl = ((W - Wh.float()) * 256.0).to(torch.float16)
    return torch.cat([Wh, Wh, Wl], dim=1).contiguous()


def attention_f32(qkv: torch.Tensor, n_images: int) -> torch.Tensor:
    _dev(qkv, torch.float32)
    _shape(qkv, "qkv", n_images * TOKENS, 3 * HIDDEN)
    out = torch.empty((n_images * TOKENS, HIDDEN), dtype=torch.float32, device=qkv.device)
    check(load().pg_op_attention_f32(_p(qkv), _p(out), n_images, _stream()), "pg_op_attention_f32")
    return out


# ----------------------------------------------------------------------------------------- refiner
class DeviceBank:
    """CSR prototype bank resident in HBM (see include/pigeon_hip.h pg_bank)."""

    FIELDS = [("proto_emb", torch.float32), ("cell_off", torch.int64), ("proto_lnglat", torch.float32),
              ("proto_count", torch.int32), ("member_off", torch.int64), ("member_idx", torch.int64),
              ("train_emb", torch.float32), ("train_lnglat", torch.float32)]

    def __init__(self, arrays, device="cuda"):
        self.t = {}
        for name, dt in self.FIELDS:
            a = getattr(arrays, name) if not isinstance(arrays, dict) else arrays[name]
            t = torch.as_tensor(a) if not torch.is_tensor(a) else a
            self.t[name] = t.to(device=device, dtype=dt).contiguous()
        self.num_cells = self.t["cell_off"].numel() - 1
        self.num_protos = self.t["proto_emb"].shape[0]
        self.num_train = self.t["train_emb"].shape[0]
        t = self.t                                            # the CSR arrays must fit each other: the kernels index by them
        _shape(t["proto_emb"], "proto_emb", self.num_protos, HIDDEN); _shape(t["proto_lnglat"], "proto_lnglat", self.num_protos, 2)
        _shape(t["proto_count"], "proto_count", self.num_protos); _shape(t["member_off"], "member_off", self.num_protos + 1)
        _shape(t["cell_off"], "cell_off", self.num_cells + 1); _shape(t["member_idx"], "member_idx", None)
        _shape(t["train_emb"], "train_emb", self.num_train, HIDDEN); _shape(t["train_lnglat"], "train_lnglat", self.num_train, 2)
        self.struct = Bank(*[C.c_void_p(self.t[n].data_ptr()) for n, _ in self.FIELDS],
                           self.num_cells, self.num_protos, self.num_train)


def refine_forward(bank: DeviceBank, q: torch.Tensor, init_llh: torch.Tensor, cand: torch.Tensor,
                   cand_prob: Optional[torch.Tensor], topk: int, temperature: float, max_refine_km: float,
                   return_scratch: bool = False):
    """Returns (preds_LLH (B,2) f32, preds_geocell (B,) i64, choice (B,) i32) [, scratch (B,topk,4) f32 =
    (score, lng, lat, bank rows streamed) per (query, candidate)]."""
    _dev(q, torch.float32); _dev(init_llh, torch.float64); _dev(cand, torch.int64)
    if cand_prob is not None:
        _dev(cand_prob, torch.float32)
    if q.dim() not in (2, 3) or cand.dim() != 2:
        raise _lib.PigeonHipError(f"refine: q must be (B,{HIDDEN}) or (B,P,{HIDDEN}) and cand (B,k), got {tuple(q.shape)} / {tuple(cand.shape)}")
    B = q.shape[0]
    P = q.shape[1] if q.dim() == 3 else 1
    k = cand.shape[1]
    _shape(q, "q", *((B, P, HIDDEN) if q.dim() == 3 else (B, HIDDEN)))
    _shape(init_llh, "init_llh", B, 2); _shape(cand, "cand", B, k)
    if cand_prob is not None:
        _shape(cand_prob, "cand_prob", B, k)
    if not 1 <= int(topk) <= k:
        raise _lib.PigeonHipError(f"refine: topk = {topk} of k = {k} candidates")
    dev = q.device
    scratch = torch.empty((B, topk, 4), dtype=torch.float32, device=dev)
    out_llh = torch.empty((B, 2), dtype=torch.float32, device=dev)
    out_cell = torch.empty((B,), dtype=torch.int64, device=dev)
    out_choice = torch.empty((B,), dtype=torch.int32, device=dev)
    check(load().pg_refine_forward(C.byref(bank.struct), _p(q), B, P, _p(init_llh), _p(cand), _p(cand_prob), k, topk,
                                   float(temperature), float(max_refine_km), _p(scratch), _p(out_llh), _p(out_cell),
                                   _p(out_choice), _stream()), "pg_refine_forward")
    if return_scratch:
        return out_llh, out_cell, out_choice, scratch
    return out_llh, out_cell, out_choice


def refine_forward_ex(bank: DeviceBank, q: torch.Tensor, init_llh: torch.Tensor, cand: torch.Tensor,
                      cand_prob: Optional[torch.Tensor], topk: int, n_eval: int, temperature: float, max_refine_km: float):
    """pg_refine_forward_ex: refine_forward's selection over the first `topk` candidates, with `n_eval` >= topk candidates evaluated and
    the 12-float records pg_refine_certainty needs.  Returns (preds_LLH, preds_geocell, choice, refined, scratch12 (B,n_eval,12))."""
    _dev(q, torch.float32); _dev(init_llh, torch.float64); _dev(cand, torch.int64)
    if cand_prob is not None:
        _dev(cand_prob, torch.float32)
    if q.dim() not in (2, 3) or cand.dim() != 2:
        raise _lib.PigeonHipError(f"refine: q must be (B,{HIDDEN}) or (B,P,{HIDDEN}) and cand (B,k), got {tuple(q.shape)} / {tuple(cand.shape)}")
    B = q.shape[0]
    P = q.shape[1] if q.dim() == 3 else 1
    k = cand.shape[1]
    _shape(q, "q", *((B, P, HIDDEN) if q.dim() == 3 else (B, HIDDEN)))
    _shape(init_llh, "init_llh", B, 2); _shape(cand, "cand", B, k)
    if cand_prob is not None:
        _shape(cand_prob, "cand_prob", B, k)
    if not 1 <= int(topk) <= int(n_eval) <= k:
        raise _lib.PigeonHipError(f"refine_ex: topk = {topk}, n_eval = {n_eval} of k = {k} candidates")
    dev = q.device
    scratch = torch.empty((B, n_eval, 12), dtype=torch.float32, device=dev)
    out_llh = torch.empty((B, 2), dtype=torch.float32, device=dev)
    out_cell = torch.empty((B,), dtype=torch.int64, device=dev)
    out_choice = torch.empty((B,), dtype=torch.int32, device=dev)
    out_refined = torch.empty((B,), dtype=torch.int32, device=dev)
    check(load().pg_refine_forward_ex(C.byref(bank.struct), _p(q), B, P, _p(init_llh), _p(cand), _p(cand_prob), k, int(topk), int(n_eval),
                                      float(temperature), float(max_refine_km), _p(scratch), _p(out_llh), _p(out_cell),
                                      _p(out_choice), _p(out_refined), _stream()), "pg_refine_forward_ex")
    return out_llh, out_cell, out_choice, out_refined, scratch


def refine_certainty(bank: DeviceBank, q: torch.Tensor, cand: torch.Tensor, cand_prob: Optional[torch.Tensor], topk: int,
                     scratch12: torch.Tensor, W: torch.Tensor, drift: Optional[torch.Tensor], wstats: torch.Tensor,
                     temperature: float, refined: torch.Tensor, choice: torch.Tensor):
    """pg_refine_certainty over the records refine_forward_ex left.  Returns (tol (B,) f32, code (B,) i32)."""
    _dev(q, torch.float32); _dev(cand, torch.int64); _dev(scratch12, torch.float32); _dev(W, torch.float32)
    _dev(wstats, torch.float32); _dev(refined, torch.int32); _dev(choice, torch.int32)
    B = q.shape[0]
    P = q.shape[1] if q.dim() == 3 else 1
    k = cand.shape[1]
    n_eval = scratch12.shape[1]
    _shape(q, "q", *((B, P, HIDDEN) if q.dim() == 3 else (B, HIDDEN)))
    _shape(cand, "cand", B, k); _shape(scratch12, "scratch12", B, n_eval, 12); _shape(W, "W", None, HIDDEN)
    _shape(refined, "refined", B); _shape(choice, "choice", B); _shape(wstats, "wstats", 2)
    if cand_prob is not None:
        _dev(cand_prob, torch.float32); _shape(cand_prob, "cand_prob", B, k)
    if drift is not None:
        _dev(drift, torch.float32); _shape(drift, "drift", HIDDEN)
    tol = torch.empty((B,), dtype=torch.float32, device=q.device)
    code = torch.empty((B,), dtype=torch.int32, device=q.device)
    check(load().pg_refine_certainty(C.byref(bank.struct), _p(q), B, P, _p(cand), _p(cand_prob), k, int(topk), int(n_eval), _p(scratch12),
                                     _p(W), W.shape[0], _p(drift), _p(wstats), float(temperature), _p(refined), _p(choice), _p(tol),
                                     _p(code), _stream()), "pg_refine_certainty")
    return tol, code


def tune_gemm_raster(gn: int) -> None:
    """pg_tune_gemm_raster: N tiles per raster group of the 384 x 256 GEMM (0 default, -1 all).  Timing only."""
    check(load().pg_tune_gemm_raster(int(gn)), "pg_tune_gemm_raster")


def tune_exact_fusion(on: bool) -> None:
    """pg_tune_exact_fusion: the exact pass's activation splits inside their producers (default on).  Bit-identical either way."""
    check(load().pg_tune_exact_fusion(1 if on else 0), "pg_tune_exact_fusion")


def tune_gemm_mid(on) -> None:
    """pg_tune_gemm_mid: the routing of GEMM launches of up to ~64 images between the 384 x 256 / 256 x 256 persistent kernels and the
    128 x 128 small-batch kernel (True / 1: on, the default; False / 0: a variant means its own kernel; 2: only the 128 x 128 kernel
    as an alternative -- the A/B arm).  Timing only: all kernels give the same bits for a row."""
    mode = 2 if (not isinstance(on, bool) and on == 2) else (1 if on else 0)
    check(load().pg_tune_gemm_mid(mode), "pg_tune_gemm_mid")
