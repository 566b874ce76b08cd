#!/usr/bin/env python3
"""Term-table kernel against the computed kernel on random clouds of random sizes and shapes (bit equality of all eight
matrices), including clustered clouds (large counts in few bins) and tiny ones.  Tools only; the test-suite holds fixed cases."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = torch.device("cuda:0")
tabled_total = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n, m = int(rng.integers(3, 7000)), int(rng.integers(3, 7000))
    kind = it % 4
    def cloud(k):
        if kind == 0:
            return rng.normal(size=(3, k)) * np.array([[60.0], [40.0], [25.0]])
        if kind == 1:
            return rng.random((3, k)) * 300.0
        if kind == 2:                                                   # clusters: some bins hold most of the neighbours
            c = rng.normal(size=(3, 5)) * 100.0
            return c[:, rng.integers(0, 5, k)] + rng.normal(size=(3, k)) * 3.0
        return np.round(rng.random((3, k)) * 20.0)                      # a lattice: duplicates and exact ties
    x, y = nat.to_dev(cloud(n), dev=dev), nat.to_dev(cloud(m), dev=dev)
    hm = K.shape_context(x, K.centroid(x), K.pca_axis(x), K.mean_distance(x), 2)["hist"]
    hf = K.shape_context(y, K.centroid(y), K.pca_axis(y), K.mean_distance(y), 4)["hist"]
    if not K.chi2_symmetric(hm, hf):
        print("case %d (%d x %d, kind %d): frames are not permutations of frame 1 here, skipped" % (it, n, m, kind))
        continue
    info = {}
    a = K.chi2_cost8_frame1(hm[0], hf[0], info=info)
    b = K.chi2_cost8(hm, hf, path="symmetric-computed")
    same = bool(torch.equal(a, b)) or bool(torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0)))
    tabled_total += sum(info["tabled_shells"])
    print("case %d (%d x %d, kind %d): identical %s, tabled shells %d" % (it, n, m, kind, same, sum(info["tabled_shells"])), flush=True)
    assert same
print("all identical; tabled shells in total:", tabled_total)
