#!/usr/bin/env python3
"""Concurrency soak: the random cases of tests/soak_cases.py registered as BATCHES (several worker threads, one HIP stream each,
shared allocator / kept cost buffers / per-device locks) must equal their stand-alone registrations bit for bit — and, seeded,
the oracle's assignments.  Usage: python tests/probes/soak_batch.py [rounds] [pairs_per_batch] [workers]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
from platymatch_amd import _native as nat, pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402
from soak_cases import make_case  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
nat.load(); pi.VERBOSE = False
fails, total = [], 0
t0 = time.perf_counter()
for rd in range(rounds):
    for transform in ("Affine", "Similar"):
        cases = [make_case(9000 + rd * 1000 + k, int(os.environ.get("PM_SOAK_MAX_POINTS", "1800"))) for k in range(per)]
        pairs = [(c[0], c[1]) for c in cases]
        seeds = [c[4] for c in cases]
        kw = dict(transform=transform, ransac_trials=120, ransac_error=20.0, icp_iterations=6)
        outcomes = []
        try:
            batch = P.estimate_transform_batch(pairs, workers=workers, seeds=seeds, **kw)
        except Exception as e:
            batch = None
            batch_exc = e
        for k, (mv, fx) in enumerate(pairs):
            try:
                alone = P.estimate_transform(mv, fx, seed=seeds[k], options={"private_rng": True}, **kw)
            except Exception as e:
                alone = e
            outcomes.append(alone)
        if batch is None:
            if not any(isinstance(o, Exception) for o in outcomes):
                fails.append("round %d %s: the batch raised %r, no stand-alone call did" % (rd, transform, batch_exc))
            continue
        for k, alone in enumerate(outcomes):
            total += 1
            if isinstance(alone, Exception):
                fails.append("round %d %s pair %d: stand-alone raised %r, the batch returned" % (rd, transform, k, alone))
                continue
            same = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(batch[k], alone))
            if not same:
                fails.append("round %d %s pair %d (N=%d, M=%d): batch != stand-alone" % (rd, transform, k, pairs[k][0].shape[1], pairs[k][1].shape[1]))
    print("... round %d done, %d pairs, %d mismatches, %.0f s" % (rd, total, len(fails), time.perf_counter() - t0), flush=True)
print("batch soak: %d pairs in batches of %d on %d workers: mismatches %d" % (total, per, workers, len(fails)))
for f in fails[:30]:
    print("  " + f)
sys.exit(1 if fails else 0)
