#!/usr/bin/env python3
"""The tile kernel's edge-guard counters against a NumPy restatement of the rule (pair counts, not just "any"), on generic, voxel
and lattice clouds: ring = neighbours within the noise of a ring radius (+ duplicates of the query), angular = within it of a
sector plane or polar cone (+ duplicates).  Usage: python tests/probes/guard_consistency.py [clouds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oracle  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

oracle.build(); nat.load()
dev = torch.device("cuda:0")
R_EDGES = np.logspace(np.log10(1 / 8), np.log10(2), 5)


def restated(cloud, c, md, x0):
    n = cloud.shape[1]
    ring_hits = ang_hits = 0
    for i in range(n):
        d = cloud[:, i]
        z = (d - c) / np.linalg.norm(d - c)
        x = x0 - z * np.dot(x0, z)
        sin_a = np.linalg.norm(x)
        x /= sin_a
        y = np.cross(z, x)
        v = np.delete(cloud, i, 1) - d[:, None]
        lx, ly, lz = np.abs(x @ v), np.abs(y @ v), np.abs(z @ v)
        r = np.sqrt(lx * lx + ly * ly + lz * lz)
        pl = np.sqrt(lx * lx + ly * ly)
        noise = 1.6e-13 * np.abs(d).sum()
        dup = r == 0
        rho = md * R_EDGES[:4]
        ring = np.any(np.abs(r[:, None] - rho[None, :]) <= np.maximum(4e-14 * rho[None, :], noise), axis=1) & ~dup
        sector = np.minimum(np.minimum(lx, ly), np.minimum(np.abs(ly - np.tan(np.pi / 6) * lx) * np.cos(np.pi / 6), np.abs(ly - np.tan(np.pi / 3) * lx) * 0.5))
        cone = np.minimum(lz, np.minimum(np.abs(lz * 0.5 - pl * np.cos(np.pi / 6)), np.abs(lz * np.cos(np.pi / 6) - pl * 0.5)))
        ang = (~(sector > np.maximum(1e-12 / max(sin_a, 1e-6) * pl, noise)) | ~(cone > noise)) & ~dup
        ring_hits += int(ring.sum() + dup.sum())
        ang_hits += int(ang.sum() + dup.sum())
    return ring_hits, ang_hits


count = int(sys.argv[1]) if len(sys.argv) > 1 else 90
bad = 0
tot = [0, 0]
for seed in range(count):
    rng = np.random.default_rng(99991 * seed + 3)
    n = int(rng.integers(5, 120))
    kind = ["generic", "voxel", "lattice"][seed % 3]
    x = rng.normal(size=(3, n)) * rng.uniform(5, 40, size=(3, 1)) + rng.uniform(0, 300, size=(3, 1))
    if kind == "voxel":
        x = np.round(x)
    elif kind == "lattice":
        x = np.round(x * 0.2) * 5.0
    c, md, x0 = np.ravel(oracle.get_centroid(x, False)), oracle.get_mean_distance(x, False), oracle.pca_axis(x.T)
    if not np.isfinite(md) or md == 0 or np.any(np.all(x == c[:, None], axis=0)):
        continue
    g = K.shape_context(nat.to_dev(np.ascontiguousarray(x), dev=dev), nat.to_dev(c, dev=dev), nat.to_dev(x0, dev=dev),
                        nat.to_dev(np.array([md]), dev=dev), 4)["guard"].cpu().tolist()
    want = restated(x, c, md, x0)
    ok = tuple(g) == want
    bad += not ok
    tot[0] += g[0]; tot[1] += g[1]
    if not ok or seed < 6:
        print("seed %d %-8s N=%3d: kernel %s, restated %s%s" % (seed, kind, n, g, list(want), "" if ok else "   <-- differ"))
print("clouds: %d, kernel hits in total (ring, angular): %s, disagreements: %d" % (count, tot, bad))
sys.exit(1 if bad else 0)
