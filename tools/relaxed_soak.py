#!/usr/bin/env python3
"""cost_mode='relaxed' against the exact mode on a stream of random registrations for a time budget: generic clouds, half-integer
lattices (tied distances, duplicates), the same cloud twice (zero-cost matches, exact ties), and clouds with NEAR-duplicates
(copies of nuclei displaced by 1e-9..1e-5 of the cloud's size: cost rows that differ in the 9th to 14th digit — alternatives inside,
at and above the certificate's margins).  Every case must return the exact mode's assignment vectors, inlier counts and 4x4
matrices bit for bit, whether its hypotheses were certified on the relaxed build or rebuilt exactly (both are counted).
Usage: python tools/relaxed_soak.py [seconds] [max_points] [first_seed] [relaxed|filter]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from soak_cases import make_case  # noqa: E402
from platymatch_amd import pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
max_points = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mode = sys.argv[4] if len(sys.argv) > 4 else "relaxed"        # or "filter" (its size threshold is lowered so that every case runs through it)
if mode == "filter":
    P.FILTER_MIN_POINTS = P.RELAXED_MIN_POINTS
pi.VERBOSE = False
lo = P.RELAXED_MIN_POINTS


def case(seed):
    """make_case's recipe at sizes the relaxed mode accepts (both clouds >= RELAXED_MIN_POINTS), plus two families of its own."""
    rng = np.random.default_rng(99991 * seed + 3)
    mv, fx, lattice, transform, rs = make_case(seed, max_points)
    while min(mv.shape[1], fx.shape[1]) < lo:                  # (sizes are drawn from 6..max_points: redraw until both are large enough)
        seed += 100003
        mv, fx, lattice, transform, rs = make_case(seed, max_points)
    kind = "lattice" if lattice else "generic"
    pick = int(rng.integers(0, 4))
    if not lattice and pick == 0:                               # the same cloud twice, permuted
        fx = np.ascontiguousarray(mv[:, rng.permutation(mv.shape[1])])
        kind = "same cloud twice"
    elif not lattice and pick == 1:                             # near-duplicates
        size = float(np.abs(mv).max())
        k = int(rng.integers(3, 40))
        for cloud in (mv, fx):
            src = rng.choice(cloud.shape[1], size=k, replace=False)
            dst = rng.choice(cloud.shape[1], size=k, replace=False)
            cloud[:, dst] = cloud[:, src] + rng.normal(size=(3, k)) * size * 10.0 ** rng.uniform(-9, -5, size=(1, k))
        kind = "near-duplicates"
    return mv, fx, kind, transform, rs


counts, fails = {}, []
t_end = time.perf_counter() + budget
t_print = time.perf_counter() + 30.0
seed = seed0
warnings.simplefilter("ignore", P.EdgeGuardWarning)
while time.perf_counter() < t_end:
    mv, fx, kind, transform, rs = case(seed)
    tag = "seed %d (%s, N=%d, M=%d, %s)" % (seed, kind, mv.shape[1], fx.shape[1], transform)
    kw = dict(transform=transform, ransac_trials=120, ransac_error=25.0 * (np.abs(mv).max() / 300.0 + 1e-9), icp_iterations=4, seed=rs)
    try:
        de, dr = {}, {}
        a = P.estimate_transform(mv, fx, details=de, **kw)
        b = P.estimate_transform(mv, fx, details=dr, cost_mode=mode, **kw)
        same = all(np.array_equal(de["lsa"][h][0], dr["lsa"][h][0]) and np.array_equal(de["lsa"][h][1], dr["lsa"][h][1]) for h in range(8))
        same = same and np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)
        if not same:
            fails.append(tag + ": relaxed mode's result differs from the exact mode's")
        modes = [d.get("cost_mode", "") for d in dr.get("assignment", {}).get("details", [])]
        c = counts.setdefault(kind, dict(cases=0, certified_on_relaxed=0, rebuilt_exactly=0))
        c["cases"] += 1
        c["certified_on_relaxed"] += sum(m.startswith(mode) for m in modes)
        c["rebuilt_exactly"] += sum(m.startswith("exact") for m in modes)
    except Exception as e:      # noqa: BLE001 — a probe: report and go on
        fails.append(tag + ": %s: %s" % (type(e).__name__, str(e)[:200]))
    seed += 1
    if time.perf_counter() > t_print:              # a progress line every half minute (a silent run is taken to be hung)
        print("... %d cases, %d mismatches" % (sum(c["cases"] for c in counts.values()), len(fails)), flush=True)
        t_print = time.perf_counter() + 30.0
print("%s soak: seeds %d..%d, %d..%d points" % (mode, seed0, seed - 1, lo, max_points))
for kind, c in sorted(counts.items()):
    print("  %-18s %4d cases: %5d hypotheses settled without an exact matrix, %5d built exactly" % (kind, c["cases"], c["certified_on_relaxed"], c["rebuilt_exactly"]))
print("mismatches: %d" % len(fails))
for f in fails[:40]:
    print("  " + f)
