"""Great-circle helpers with the reference's names and conventions (preprocessing/geo_utils.py).
Inputs are [lng, lat] in degrees, R = 6378137 m, result in km.  The float64 veto inside the refinement
kernel (csrc/refine.hip haversine_km) implements the same formula on the device."""
import numpy as np
import torch
from torch import Tensor

rad_np = np.float64(6378137.0)
rad_torch = torch.tensor(6378137.0, dtype=torch.float64)


def haversine_np(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """reference preprocessing/geo_utils.py:23-38"""
    x_rad, y_rad = map(np.radians, [x, y])
    delta = y_rad - x_rad
    a = np.sin(delta[:, 1] / 2) ** 2 + np.cos(x_rad[:, 1]) * np.cos(y_rad[:, 1]) * np.sin(delta[:, 0] / 2) ** 2
    c = 2 * np.arcsin(np.sqrt(a))
    return (rad_np * c) / 1000


def haversine(x: Tensor, y: Tensor) -> Tensor:
    """reference preprocessing/geo_utils.py:40-55"""
    x_rad, y_rad = torch.deg2rad(x), torch.deg2rad(y)
    delta = y_rad - x_rad
    a = torch.sin(delta[:, 1] / 2) ** 2 + torch.cos(x_rad[:, 1]) * torch.cos(y_rad[:, 1]) * torch.sin(delta[:, 0] / 2) ** 2
    c = 2 * torch.arcsin(torch.sqrt(a))
    return (rad_torch.to(c.device) * c) / 1000


def haversine_matrix(x: Tensor, y: Tensor) -> Tensor:
    """reference preprocessing/geo_utils.py:58-74: x (N,2), y (2,M) -> (N,M) km.  Device tensors with a float64 `y`
    (the SuperGuessr call, models/super_guessr.py:470) run in pg_haversine_matrix; host tensors use the torch
    expression below (this function is not on the hot path)."""
    if x.is_cuda and y.is_cuda and y.dtype == torch.float64 and x.dtype in (torch.float32, torch.float64) and x.dim() == 2:
        from . import hip_ops
        return hip_ops.haversine_matrix(x.contiguous(), y.t().contiguous())
    x_rad, y_rad = torch.deg2rad(x), torch.deg2rad(y)
    delta = x_rad.unsqueeze(2) - y_rad
    p = torch.cos(x_rad[:, 1]).unsqueeze(1) * torch.cos(y_rad[1, :]).unsqueeze(0)
    a = torch.sin(delta[:, 1, :] / 2) ** 2 + p * torch.sin(delta[:, 0, :] / 2) ** 2
    c = 2 * torch.arcsin(torch.sqrt(a))
    return (rad_torch.to(c.device) * c) / 1000


def smooth_labels(distances: Tensor, constant: float = None) -> Tensor:
    """reference preprocessing/utils.py:7-19: exp(-(d - rowmin d) / LABEL_SMOOTHING_CONSTANT), NaN/inf -> 0."""
    from .config import LABEL_SMOOTHING_CONSTANT
    constant = LABEL_SMOOTHING_CONSTANT if constant is None else constant
    if distances.is_cuda and distances.dtype == torch.float64 and distances.dim() == 2:
        from . import hip_ops
        return hip_ops.smooth_labels(distances.contiguous(), constant)
    adj_distances = distances - distances.min(dim=-1, keepdim=True)[0]
    smoothed_labels = torch.exp(-adj_distances / constant)
    return torch.nan_to_num(smoothed_labels, nan=0.0, posinf=0.0, neginf=0.0)
