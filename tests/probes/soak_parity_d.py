#!/usr/bin/env python3
"""The soak's fourth family, SURVEY §8(f): label image -> centroids (random volumes, dtypes, label gaps, anisotropy), cdist
(random shapes), the evaluation metrics (random keypoints / detections / transforms), PCA-only alignment — against the oracle.
Usage: python tests/probes/soak_parity_d.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import oracle  # noqa: E402
from platymatch_amd import _native as nat  # noqa: E402
from platymatch_amd.evaluate_metrics import calculate_metrics, cdist  # noqa: E402
from platymatch_amd.label_image import label_centroids, ransac_error_from_sizes  # noqa: E402
from platymatch_amd.pipeline import pca_alignment  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle.build(); nat.load()
counts = dict(cases=0)
fails = []
t_end = time.perf_counter() + budget
seed = seed0
while time.perf_counter() < t_end:
    rng = np.random.default_rng(31337 * seed + 1)
    tag = "seed %d" % seed
    # ---- label image
    shape = tuple(int(v) for v in rng.integers(1, [40, 70, 140]))
    dtype = [np.uint8, np.int16, np.uint16, np.int32, np.int64, np.uint32][seed % 6]
    kind = seed % 4
    if kind == 0:                                           # blobs on a coarse grid, with holes
        zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
        lab = ((zz // 4) * 36 + (yy // 9) * 6 + xx // 25 + 1) % (200 if dtype is np.uint8 else 5000)
        lab[rng.random(shape) < 0.3] = 0
    elif kind == 1:                                         # salt and pepper, sparse label ids
        lab = rng.integers(0, 250 if dtype is np.uint8 else 3000, size=shape) * (rng.random(shape) < 0.2)
    elif kind == 2:                                         # one big label and a few voxels of others
        lab = np.full(shape, 3)
        lab[rng.random(shape) < 0.01] = 77
    else:                                                   # empty but for (maybe) one voxel
        lab = np.zeros(shape, dtype=np.int64)
        if rng.random() < 0.7:
            lab[tuple(int(rng.integers(0, s)) for s in shape)] = 5
    lab = lab.astype(dtype)
    aniso = float(rng.choice([1.0, 2.0, 0.5, 3.3]))
    try:
        oc, osz, oid = oracle.label_centroids(lab, aniso)
        gc, gsz, gid = label_centroids(lab, aniso)
        if oid.size == 0:
            ok = gid.size == 0 and gsz.size == 0
        else:
            ok = np.array_equal(gc, oc) and np.array_equal(gsz, osz) and np.array_equal(np.asarray(gid, dtype=np.int64), np.asarray(oid, dtype=np.int64))
        if not ok:
            fails.append(tag + ": label_centroids differ (shape %s, %s, kind %d)" % (shape, dtype.__name__, kind))
        elif oid.size and ransac_error_from_sizes(gsz, gsz) != oracle.ransac_error_from_sizes(osz, osz):
            fails.append(tag + ": ransac_error_from_sizes differs")
    except Exception as e:
        fails.append(tag + ": label probe error %r (shape %s, %s, kind %d)" % (e, shape, dtype.__name__, kind))
    # ---- cdist + metrics
    try:
        n, m, k = int(rng.integers(1, 400)), int(rng.integers(1, 400)), int(rng.integers(1, 12))
        a, b = rng.normal(size=(3, n)) * 50 + 100, rng.normal(size=(3, m)) * 50 + 100
        if seed % 3 == 0:
            a, b = np.round(a), np.round(b)
        if not np.array_equal(cdist(a, b), oracle.cdist(a, b)):
            fails.append(tag + ": cdist bits differ (%d x %d)" % (n, m))
        k = min(k, n, m)
        T1 = np.eye(4); T1[:3, :3] += rng.normal(scale=0.05, size=(3, 3)); T1[:3, 3] = rng.normal(scale=5, size=3)
        T2 = np.eye(4); T2[:3, 3] = rng.normal(scale=1, size=3)
        mid, fid = rng.permutation(10 * n)[:n], rng.permutation(10 * m)[:m]
        mk_idx, fk_idx = rng.permutation(n)[:k], rng.permutation(m)[:k]
        kp_ids = rng.permutation(100)[:k]
        mk = a[:, mk_idx] + rng.normal(scale=0.3, size=(3, k))
        fk = b[:, fk_idx] + rng.normal(scale=0.3, size=(3, k))
        args = (mk, kp_ids, a, mid, fk, kp_ids[rng.permutation(k)] if seed % 2 else kp_ids, b, fid, T1, T2)
        o = oracle.calculate_metrics(*args)
        g = calculate_metrics(*args)
        if not (o[0] == g[0] and abs(o[1] - g[1]) <= 1e-11 * max(abs(o[1]), 1e-300)):
            fails.append(tag + ": calculate_metrics %r vs %r" % (g, o))
    except Exception as e:
        fails.append(tag + ": metrics probe error %r" % (e,))
    # ---- PCA-only alignment
    try:
        p, q = int(rng.integers(4, 2000)), int(rng.integers(4, 2000))
        ca = rng.normal(size=(3, p)) * rng.uniform(5, 60, size=(3, 1)) + 50
        cb = rng.normal(size=(3, q)) * rng.uniform(5, 60, size=(3, 1)) - 20
        mt, ftm = pca_alignment(ca, cb)
        for got, cloud in ((mt, ca), (ftm, cb)):
            want = oracle.pca_components(cloud.T)
            sv = np.linalg.svd(cloud - cloud.mean(1, keepdims=True), compute_uv=False)
            gap = min(sv[0] - sv[1], sv[1] - sv[2]) / sv[0]
            if gap > 1e-3 and not np.abs(np.asarray(got) - want).max() < 1e-9 / gap:
                fails.append(tag + ": pca components differ by %.1e (gap %.1e)" % (np.abs(np.asarray(got) - want).max(), gap))
    except Exception as e:
        fails.append(tag + ": pca probe error %r" % (e,))
    counts["cases"] += 1
    seed += 1
print("soak D: seeds %d..%d: %s" % (seed0, seed - 1, counts))
print("mismatches: %d" % len(fails))
for f in fails[:40]:
    print("  " + f)
sys.exit(1 if fails else 0)
