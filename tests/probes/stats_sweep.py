#!/usr/bin/env python3
"""Centroid (both layouts) and mean pairwise distance against the oracle, bit for bit, for EVERY cloud size 2..600 (the pair list
crosses NumPy's 8 192-element pieces at N = 129, 182, 223, ...; leaves of 128; remainders of every length) and a few large ones.
Usage: python tests/probes/stats_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import oracle  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

oracle.build(); nat.load()
dev = torch.device("cuda:0")
bad = []
sizes = list(range(2, 601)) + [1000, 1449, 2048, 4097, 8191, 8192, 8193, 12345, 16385, 20011]
rng = np.random.default_rng(5)
for n in sizes:
    x = rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + rng.uniform(-500, 500, size=(3, 1))
    if n % 7 == 0:
        x = np.round(x)
    xd = nat.to_dev(np.ascontiguousarray(x), dev=dev)
    if K.mean_distance(xd).item() != oracle.get_mean_distance(x, False):
        bad.append("mean distance, N=%d" % n)
    if not np.array_equal(K.centroid(xd).cpu().numpy(), np.ravel(oracle.get_centroid(x, False))):
        bad.append("centroid 3 x N, N=%d" % n)
    if not np.array_equal(K.centroid(xd, sequential=True).cpu().numpy(), np.ravel(oracle.get_centroid(np.ascontiguousarray(x.T), True))):
        bad.append("centroid N x 3, N=%d" % n)
print("sizes checked: %d; mismatches: %d" % (len(sizes), len(bad)))
for b in bad[:30]:
    print("  " + b)
sys.exit(1 if bad else 0)
