#!/usr/bin/env python3
"""Host-side cost of the NumPy-exact RANSAC index draws (pm_legacy_choice): seconds per 8 x 8000 trials.  Tools only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platymatch_amd.estimate_transform.shape_context import draw_ransac_samples  # noqa: E402

draw_ransac_samples(1000, 4, 10, rng=np.random.RandomState(0))
for n in [int(a) for a in sys.argv[1:]] or [5000, 20000, 50000]:
    best = 1e9
    for rep in range(3):
        rs = np.random.RandomState(rep)
        t = time.perf_counter()
        draw_ransac_samples(n, 4, 64000, rng=rs)
        best = min(best, time.perf_counter() - t)
    print("n = %6d: 8 x 8000 trials in %.3f s  (%.3f ns per shuffled element)" % (n, best, best / (64000.0 * n) * 1e9), flush=True)
