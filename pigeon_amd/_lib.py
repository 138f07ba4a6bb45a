"""ctypes binding of libpigeon_hip.so (include/pigeon_hip.h).  The product path has NO fallback: if the HIP
library is missing or there is no GPU, importing/using the ops raises -- loudly (see `require_gpu`)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PIGEON_HIP_LIB lets a developer A/B two builds of the library in one process launch (tools/); default = in-tree build
LIB_PATH = os.environ.get("PIGEON_HIP_LIB") or os.path.join(_HERE, "libpigeon_hip.so")

PG_DTYPE_F32, PG_DTYPE_BF16, PG_DTYPE_F16, PG_DTYPE_F64 = 0, 1, 2, 3
EPI_QKV, EPI_GELU, EPI_RESID, EPI_PATCH, EPI_F32, EPI_RESID_STAT, EPI_QKV_LN, EPI_GELU_LN, EPI_GELU_X3 = 0, 1, 2, 3, 4, 5, 6, 7, 8
PROF_CLASSES = ["gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2", "gemm_patch", "attention", "layernorm",
                "im2col", "token_mean"]


class PigeonHipError(RuntimeError):
    pass


class VitCfg(C.Structure):
    _fields_ = [("layers", C.c_int32), ("image_size", C.c_int32), ("patch", C.c_int32), ("hidden", C.c_int32),
                ("heads", C.c_int32), ("mlp", C.c_int32), ("ln_eps", C.c_float), ("max_chunk", C.c_int32),
                ("mma_dtype", C.c_int32), ("precise", C.c_int32)]


class Bank(C.Structure):
    _fields_ = [("proto_emb", C.c_void_p), ("cell_off", C.c_void_p), ("proto_lnglat", C.c_void_p),
                ("proto_count", C.c_void_p), ("member_off", C.c_void_p), ("member_idx", C.c_void_p),
                ("train_emb", C.c_void_p), ("train_lnglat", C.c_void_p),
                ("num_cells", C.c_int64), ("num_protos", C.c_int64), ("num_train", C.c_int64)]


# name -> (restype, argtypes); must list EVERY symbol declared in include/pigeon_hip.h (tests check this)
_P, _I, _I64, _F, _D, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
SIGNATURES = {
    "pg_last_error": (C.c_char_p, []),
    "pg_abi_version": (_I, []),
    "pg_device_count": (_I, []),
    "pg_vit_create": (_I, [C.POINTER(_P), _I, C.POINTER(VitCfg)]),
    "pg_vit_load_weight": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_I64), _I]),
    "pg_vit_finalize": (_I, [_P]),
    "pg_vit_workspace_bytes": (_I, [_P, _I, C.POINTER(_SZ)]),
    "pg_vit_forward": (_I, [_P, _P, _I, _I, _P, _P, _SZ, _P]),
    "pg_vit_forward_hidden": (_I, [_P, _P, _I, _I, _P, _P, _P, _SZ, _P]),
    "pg_vit_precise_workspace_bytes": (_I, [_P, _I, C.POINTER(_SZ)]),
    "pg_vit_forward_precise": (_I, [_P, _P, _I, _I, _P, _P, _P, _SZ, _P]),
    "pg_vit_graph": (_I, [_P, _I, C.POINTER(_I64), C.POINTER(_I64)]),
    "pg_vit_destroy": (_I, [_P]),
    "pg_vit_mma_dtype": (_I, [_P]),
    "pg_vit_profile_enable": (_I, [_P, _I]),
    "pg_vit_profile_read": (_I, [_P, C.POINTER(_I64), C.POINTER(_D)]),
    "pg_vit_profile_reset": (_I, [_P]),
    "pg_tune_gemm_stagger": (_I, [_F]),
    "pg_tune_gemm_tail_rows": (_I, [_I]),
    "pg_tune_gemm_tail_shape": (_I, [_I, _I]),
    "pg_tune_gemm_raster": (_I, [_I]),
    "pg_tune_gemm_mid": (_I, [_I]),
    "pg_tune_exact_attention": (_I, [_I]),
    "pg_tune_exact_products": (_I, [_I]),
    "pg_tune_exact_fusion": (_I, [_I]),
    "pg_gemm_route": (_I, [_I, _I, _I, _I, _I, _P]),
    "pg_vit_saturation_check": (_I, [_P, _I]),
    "pg_vit_saturation_read": (_I, [_P, C.POINTER(_I64), _I]),
    "pg_vit_range_alarm_read": (_I, [_P, C.POINTER(_I64), _I]),
    "pg_comm_unique_id": (_I, [_P]),
    "pg_comm_init_rank": (_I, [C.POINTER(_P), _I, _P, _I]),
    "pg_comm_count": (_I, [_P, C.POINTER(_I)]),
    "pg_allgather": (_I, [_P, _P, _P, _SZ, _P]),
    "pg_allgather_many": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_SZ), _P]),
    "pg_comm_destroy": (_I, [_P]),
    "pg_comm_rccl_version": (_I, []),
    "pg_prep_create": (_I, [C.POINTER(_P), _I, _I, _I]),
    "pg_prep_destroy": (_I, [_P]),
    "pg_prep_geometry": (_I, [_P, C.POINTER(C.c_int32)]),
    "pg_prep_workspace_bytes": (_I, [_P, _I, C.POINTER(_SZ)]),
    "pg_prep_forward": (_I, [_P, _P, _I, _P, _I, _P, _SZ, _P]),
    "pg_proto_build": (_I, [_P, _I, _I64, _P, _P, _I64, _P, _P]),
    "pg_haversine_matrix": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "pg_haversine_pairs": (_I, [_P, _P, _I, _I64, _P, _P]),
    "pg_smooth_labels": (_I, [_P, _I, _I, _D, _P, _P]),
    "pg_head_forward": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "pg_head_margin": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "pg_head_certainty": (_I, [_P, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "pg_refine_forward": (_I, [C.POINTER(Bank), _P, _I, _I, _P, _P, _P, _I, _I, _F, _D, _P, _P, _P, _P, _P]),
    "pg_refine_forward_ex": (_I, [C.POINTER(Bank), _P, _I, _I, _P, _P, _P, _I, _I, _I, _F, _D, _P, _P, _P, _P, _P, _P]),
    "pg_refine_certainty": (_I, [C.POINTER(Bank), _P, _I, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _F, _P, _P, _P, _P, _P]),
    "pg_requeue_append": (_I, [_P, _P, _P, _I, _F, _I, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "pg_rows_to_slots": (_I, [_P, _I64, _P, _I, _P, _P]),
    "pg_requeue_take": (_I, [_P, _I64, _I64, _I, _I, _P, _P]),
    "pg_scatter_rows": (_I, [_P, _I64, _P, _I, _P, _I64, _I64, _I64, _I64, _P]),
    "pg_head_wstats": (_I, [_P, _I, _P, _P, _P]),
    "pg_embedding_debias": (_I, [_P, _I64, _I, _P, _P]),
    "pg_op_gemm16": (_I, [_I, _P, _I64, _P, _P, _P, _I64, _I, _I, _I, _I, _F, _I, _P, _I, _P]),
    "pg_op_gemm16_ld": (_I, [_I, _P, _I64, _P, _I64, _P, _P, _I64, _I, _I, _I, _I, _F, _I, _P, _I, _P]),
    "pg_op_rowstat_cast": (_I, [_P, _P, _I, _P, _I64, _F, _P]),
    "pg_op_gemm16_resid_stat": (_I, [_I, _P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _P, _I, _I, _I, _I, _P]),
    "pg_op_rowstat_finalize": (_I, [_P, _I, _P, _I64, _F, _P]),
    "pg_op_gemm16_ln": (_I, [_I, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _I, _I, _I, _I, _F, _I, _I, _P]),
    "pg_op_layernorm": (_I, [_P, _P, _P, _P, _I, _I64, _F, _P]),
    "pg_op_attention": (_I, [_I, _P, _P, _I, _P]),
    "pg_op_im2col": (_I, [_P, _I, _P, _I, _I, _P]),
    "pg_op_token_mean": (_I, [_P, _P, _I, _P]),
    "pg_op_cast_f32": (_I, [_P, _P, _I, _I64, _P]),
    "pg_op_gemm16_parts": (_I, [_I, _P, _I64, _P, _I64, _P, _P, _I, _I, _I, _I, _P]),
    "pg_op_x3_split": (_I, [_P, _P, _I64, _I, _I, _P]),
    "pg_op_x3_layernorm": (_I, [_P, _P, _P, _P, _I64, _F, _P]),
    "pg_op_attention_f32": (_I, [_P, _P, _I, _P]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PigeonHipError(
            f"{LIB_PATH} not found: build it with `python -m pigeon_amd.build` (hipcc, gfx950). "
            "pigeon_amd has no CPU/PyTorch fallback for its hot path.")
    # One HIP runtime per process: the library receives torch's streams and device pointers, so it must bind to the libamdhip64
    # torch has loaded (torch ships its own copy).  Loaded FIRST, libpigeon_hip.so would pull /opt/rocm's runtime in under the same
    # soname and torch would then find "No HIP GPUs" (seen when build() and smoke() ran in one process).  Importing torch here makes
    # the order irrelevant; it is the tensor container of every caller of this module anyway.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().pg_last_error()
        raise PigeonHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def require_gpu():
    """Fail loudly when the HIP path cannot run (missing library or no device)."""
    lib = load()
    n = lib.pg_device_count()
    if n <= 0:
        raise PigeonHipError("pigeon_amd needs an AMD GPU (gfx950) visible to HIP; found none. "
                             "There is no CPU fallback for the hot path.")
    return n
