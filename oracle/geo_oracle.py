"""ORACLE -- host restatement of the reference's great-circle / label-smoothing / metric helpers.
TEST INFRASTRUCTURE, NOT PRODUCT: only tests/ may import this (pigeon_amd/ never does).

Restates, in plain torch / numpy with torch's own dtype promotion (so fp32 inputs behave as in the reference):
  haversine          preprocessing/geo_utils.py:40-55
  haversine_matrix   preprocessing/geo_utils.py:58-74
  smooth_labels      preprocessing/utils.py:7-19 (LABEL_SMOOTHING_CONSTANT = 65, config.py:51)
  geoguessr_metrics  evaluation/metrics.py:89-181 minus country accuracy (geopandas) and the multi-task heads
Pinned to outputs of the reference's OWN functions on seeded inputs: tests/golden/geo.npz, written by
oracle/make_golden.py --only geo (which imports them through oracle/reference_loader.py); tests/test_oracle_golden.py.
"""
import numpy as np
import torch

RAD = torch.tensor(6378137.0, dtype=torch.float64)


def haversine(x, y):
    x_rad, y_rad = torch.deg2rad(x), torch.deg2rad(y)
    delta = y_rad - x_rad
    a = torch.sin(delta[:, 1] / 2) ** 2 + torch.cos(x_rad[:, 1]) * torch.cos(y_rad[:, 1]) * torch.sin(delta[:, 0] / 2) ** 2
    c = 2 * torch.arcsin(torch.sqrt(a))
    return (RAD * c) / 1000


def haversine_matrix(x, y):
    x_rad, y_rad = torch.deg2rad(x), torch.deg2rad(y)
    delta = x_rad.unsqueeze(2) - y_rad
    p = torch.cos(x_rad[:, 1]).unsqueeze(1) * torch.cos(y_rad[1, :]).unsqueeze(0)
    a = torch.sin(delta[:, 1, :] / 2) ** 2 + p * torch.sin(delta[:, 0, :] / 2) ** 2
    c = 2 * torch.arcsin(torch.sqrt(a))
    return (RAD * c) / 1000


def smooth_labels(distances, constant=65):
    adj = distances - distances.min(dim=-1, keepdim=True)[0]
    return torch.nan_to_num(torch.exp(-adj / constant), nan=0.0, posinf=0.0, neginf=0.0)


def haversine_np(x, y):
    x_rad, y_rad = np.radians(x), np.radians(y)
    delta = y_rad - x_rad
    a = np.sin(delta[:, 1] / 2) ** 2 + np.cos(x_rad[:, 1]) * np.cos(y_rad[:, 1]) * np.sin(delta[:, 0] / 2) ** 2
    return (np.float64(6378137.0) * (2 * np.arcsin(np.sqrt(a)))) / 1000


def geoguessr_metrics(predictions, labels, cell_preds, cell_labels, top5_geocells):
    """evaluation/metrics.py:138-166: the distance / geocell entries of compute_geoguessr_metrics."""
    d = haversine_np(predictions, labels)
    out = {'Mean_km_error': np.mean(d), 'Median_km_error': np.median(d)}
    for km in (1, 5, 10, 25, 50, 100, 200, 750, 1000, 2500):
        out[f'Under_{km}_km'] = (d < km).sum() / len(d)                                   # :89-100
    out['Geoguessr_score'] = np.mean(np.round(5000 * np.exp(-d / 1492.7)))               # :102-114, DECAY_CONSTANT config.py:50
    out['Geocell_accuracy'] = float(np.mean(np.asarray(cell_labels) == np.asarray(cell_preds)))
    out['Geocell_top5_accuracy'] = sum(int(l in t) for l, t in zip(cell_labels, top5_geocells)) / len(cell_labels)   # :116-136
    return out
