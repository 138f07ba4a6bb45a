"""Offline look at the STRUCTURE of the fast path's embedding error (round 5), on the dumps of tools/dump_fast_exact.py:
is there anything beyond one constant systematic vector that a calibration could predict?
   python tools/error_model_offline.py gpurun_out/r05/fast_exact_default.npz [...]
Fits on the even samples, judges on the odd ones: the constant drift; a per-sample scale of it; a ridge-regularised affine map of
the (normalised) embedding; and how much of the held-out residual lies in the leading principal directions of the training residual."""
import sys
import numpy as np

for path in sys.argv[1:]:
    z = np.load(path)
    f, e = z["fast"].astype(np.float64), z["exact"].astype(np.float64)
    n = len(e)
    en = np.linalg.norm(e, axis=1, keepdims=True)
    rel = (f - e) / en
    rms = lambda a: float(np.sqrt((np.linalg.norm(a, axis=1) ** 2).mean()))
    tr, te = np.arange(0, n, 2), np.arange(1, n, 2)
    beta = rel[tr].mean(0)
    res = rel[te] - beta
    print(f"{path}: {n} samples; relative error RMS {rms(rel):.3e}; after the constant systematic vector (held out) {rms(res):.3e}")
    s = (rel @ beta) / (beta @ beta)
    print(f"   per-sample scale of the systematic vector: mean {s.mean():.3f}, std {s.std():.3f}; residual if that scale were known {rms(rel - np.outer(s, beta)):.3e}")
    u = e / en
    u0 = u[tr].mean(0)
    X, Y = u[tr] - u0, rel[tr] - beta
    K = X @ X.T
    for lam in (1e-4, 1e-2, 1e-1, 1.0):
        A = np.linalg.solve(K + lam * np.trace(K) / len(K) * np.eye(len(K)), Y)
        print(f"   affine in the embedding, ridge {lam:g}: held-out residual {rms(res - (u[te] - u0) @ X.T @ A):.3e}")
    _, _, Vt = np.linalg.svd(rel[tr] - beta, full_matrices=False)
    print("   share of the held-out residual in the top k principal directions of the training residual: " +
          ", ".join(f"k={k}: {float(((res @ Vt[:k].T) ** 2).sum() / (res ** 2).sum()):.3f}" for k in (1, 4, 16, 64)))
