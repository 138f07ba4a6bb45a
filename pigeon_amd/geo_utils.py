"""Great-circle helpers with the reference's names and conventions (preprocessing/geo_utils.py, preprocessing/utils.py).
Inputs are [lng, lat] in degrees, R = 6378137 m, results in km.

`haversine`, `haversine_matrix` and `smooth_labels` run in libpigeon_hip.so (csrc/geo_proto.hip) on device tensors and
RAISE on host tensors: this package has no CPU/PyTorch fallback (the host restatements used to check the kernels live
in oracle/geo_oracle.py, pinned to outputs of the reference's own functions).  `haversine_np` is the reference's numpy
helper for the evaluation metrics, which the reference computes on the host from the collected predictions
(evaluation/metrics.py:150) -- it is not on the device path.
"""
import numpy as np
import torch
from torch import Tensor

from ._lib import PigeonHipError

rad_np = np.float64(6378137.0)


def haversine_np(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """reference preprocessing/geo_utils.py:23-38 (host metric helper)"""
    x_rad, y_rad = map(np.radians, [x, y])
    delta = y_rad - x_rad
    a = np.sin(delta[:, 1] / 2) ** 2 + np.cos(x_rad[:, 1]) * np.cos(y_rad[:, 1]) * np.sin(delta[:, 0] / 2) ** 2
    c = 2 * np.arcsin(np.sqrt(a))
    return (rad_np * c) / 1000


def _need_cuda(name: str, *ts: Tensor):
    for t in ts:
        if not t.is_cuda:
            raise PigeonHipError(f'pigeon_amd.{name} runs on the GPU only (pg_{name}); got a {t.device} tensor. '
                                 'There is no CPU fallback.')


def haversine(x: Tensor, y: Tensor) -> Tensor:
    """reference preprocessing/geo_utils.py:40-55: row-paired distances, x (N,2) float64, y (N,2) float32/float64."""
    _need_cuda('haversine_pairs', x, y)
    from . import hip_ops
    return hip_ops.haversine_pairs(x.to(torch.float64).contiguous(), y.contiguous())


def haversine_matrix(x: Tensor, y: Tensor) -> Tensor:
    """reference preprocessing/geo_utils.py:58-74: x (N,2), y (2,M) float64 -> (N,M) km (the SuperGuessr soft-label
    call, models/super_guessr.py:470)."""
    _need_cuda('haversine_matrix', x, y)
    if y.dtype != torch.float64 or x.dtype not in (torch.float32, torch.float64) or x.dim() != 2:
        raise PigeonHipError('haversine_matrix: x (N,2) float32/float64 and y (2,M) float64 expected')
    from . import hip_ops
    return hip_ops.haversine_matrix(x.contiguous(), y.t().contiguous())


def smooth_labels(distances: Tensor, constant: float = None) -> Tensor:
    """reference preprocessing/utils.py:7-19: exp(-(d - rowmin d) / LABEL_SMOOTHING_CONSTANT), NaN/inf -> 0."""
    from .config import LABEL_SMOOTHING_CONSTANT
    constant = LABEL_SMOOTHING_CONSTANT if constant is None else constant
    _need_cuda('smooth_labels', distances)
    if distances.dtype != torch.float64 or distances.dim() != 2:
        raise PigeonHipError('smooth_labels: (N,M) float64 distances expected')
    from . import hip_ops
    return hip_ops.smooth_labels(distances.contiguous(), constant)
