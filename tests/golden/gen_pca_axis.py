#!/usr/bin/env python3
"""sklearn.decomposition.PCA(n_components=3).fit(X).components_[0] — the axis get_unary orients every local frame by
(shape_context.py:162-165) — for 40 clouds of 4 .. 1 500 points in the layouts the reference meets: the transposed view of a
C-ordered 3 x N array (what get_unary passes when transposed=False), a C-ordered N x 3 array, integer-valued (voxel) coordinates,
and a 4 x N array whose 4th row is dropped by a slice.  Both solver branches of svd_solver='auto' occur (n >= 30: covariance_eigh;
below: full).  Also checks, over 3 500 further random clouds that are NOT stored, that the NumPy restatement used by the oracle and by
the product (pca_axis_host) reproduces sklearn bit for bit.  Data only; build container only (scikit-learn is not needed at test time).
Usage: python tests/golden/gen_pca_axis.py"""
import os
import sys

import numpy as np
from sklearn.decomposition import PCA

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

rng = np.random.default_rng(4242)
out = {}
k = 0
for t in range(40):
    n = int(rng.integers(4, 40)) if t % 4 == 0 else int(rng.integers(30, 1501))
    A = rng.normal(size=(3, n)) * rng.uniform(1, 80, size=(3, 1)) + rng.uniform(-300, 300, size=(3, 1))
    kind = t % 4
    if kind == 1:
        A = np.round(A)
    if kind == 3:
        A4 = np.vstack([A, rng.uniform(1, 9, size=(1, n))])           # 4 x N: the reference drops the last column of the transpose
        X = A4.transpose()[:, :3]
        out["c%02d_cloud" % k] = A4
    elif kind == 2:
        X = np.ascontiguousarray(A.T)                                  # N x 3, C order (transposed=True callers)
        out["c%02d_cloud" % k] = X
    else:
        X = A.transpose()
        out["c%02d_cloud" % k] = A
    out["c%02d_kind" % k] = np.array([kind])
    out["c%02d_axis" % k] = PCA(n_components=3).fit(X).components_[0].copy()
    assert np.array_equal(oracle.pca_axis(X), out["c%02d_axis" % k]), (k, n, kind)
    k += 1
out["cases"] = np.array([k])
bad = 0
for t in range(3500):
    n = int(rng.integers(4, 3000)) if t % 3 else int(rng.integers(4, 60))
    A = rng.normal(size=(3, n)) * rng.uniform(1, 80, size=(3, 1)) + rng.uniform(-300, 300, size=(3, 1))
    if t % 5 == 4:
        A = np.round(A)
    X = A.transpose() if t % 2 else np.ascontiguousarray(A.T)
    bad += not np.array_equal(PCA(n_components=3).fit(X).components_[0], oracle.pca_axis(X))
print("restatement against sklearn on 3500 unstored clouds: %d differ" % bad)
assert bad == 0
np.savez_compressed(os.path.join(HERE, "pca_axis.npz"), **out)
print("wrote", os.path.getsize(os.path.join(HERE, "pca_axis.npz")), "bytes")
