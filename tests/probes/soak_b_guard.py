#!/usr/bin/env python3
"""Planar cases of soak family B whose assignments differ from the oracle's: does the edge guard report them?
Usage: python tests/probes/soak_b_guard.py SEED [SEED ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import platymatch_amd  # noqa: E402
from platymatch_amd import _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402
from soak_cases import make_case_b  # noqa: E402

oracle.build(); nat.load(); pi.VERBOSE = False
for seed in [int(a) for a in sys.argv[1:]]:
    c = make_case_b(seed, 400)
    det, odet = {}, {}
    ref = oracle.estimate_transform(c["mv"], c["fx"], details=odet, **c["kwargs"])
    got = platymatch_amd.register(c["mv"], c["fx"], details=det, **c["kwargs"])
    differ = [h for h in range(8) if not np.array_equal(det["lsa"][h][1], odet["lsa"][h][1])]
    zs = [float(np.ptp(c[k][2])) for k in ("mv", "fx")]
    cen = [oracle.get_centroid(c[k], False).ravel()[2] - c[k][2, 0] for k in ("mv", "fx")]
    print("seed %d kind %d: hypotheses whose assignment differs %s; edge guard %s; z extent %s; centroid off the plane by %s"
          % (seed, c["kind"], differ, det.get("edge_guard"), zs, cen))
