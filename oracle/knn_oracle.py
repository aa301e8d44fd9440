"""TEST INFRASTRUCTURE - CPU restatement of the reference's distCUDA2 (submodules/simple-knn/simple_knn.cu:129-221):
the mean of the squared distances from every point to its three nearest OTHER points (duplicates at distance 0 count;
a missing neighbour counts as FLT_MAX, simple_knn.cu:155,193).  The reference's box pruning is exact, so the answer
is the plain 3-NN answer: brute force in fp32 for small P (the arithmetic of `updateKBest`, simple_knn.cu:113-127),
a k-d tree (scipy, float64 search + fp32 distances) for large P.  Only tests may import this module."""
from __future__ import annotations

import numpy as np

FLT_MAX = np.float32(np.finfo(np.float32).max)


def mean_dist2_bruteforce(points: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(points, np.float32)
    P = p.shape[0]
    out = np.empty(P, np.float32)
    for i in range(P):
        d = p - p[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32)
        d2[i] = FLT_MAX
        best = np.sort(d2)[:3] if P > 1 else np.array([], np.float32)
        best = np.concatenate([best[best < FLT_MAX] if P <= 3 else best, np.full(3, FLT_MAX, np.float32)])[:3]
        with np.errstate(over="ignore"):
            out[i] = (best[0] + best[1] + best[2]) / np.float32(3.0)
    return out


def mean_dist2_kdtree(points: np.ndarray) -> np.ndarray:
    from scipy.spatial import cKDTree
    p = np.ascontiguousarray(points, np.float32)
    _, idx = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=4)
    out = np.empty(p.shape[0], np.float32)
    # the tree may return the query itself anywhere among equal-distance candidates: drop one self index per row
    for i in range(p.shape[0]):
        nb = [j for j in idx[i] if j != i][:3] if i in idx[i] else list(idx[i][:3])
        d = p[nb] - p[i]
        d2 = np.sort((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32))
        out[i] = (d2[0] + d2[1] + d2[2]) / np.float32(3.0)
    return out
