#!/usr/bin/env python3
"""One large batch under a watchdog: stacks of all threads every 60 s (is a slow batch slow or stuck?).
Usage: python tests/probes/soak_batch_trace.py [max_points] [pairs] [workers]"""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from platymatch_amd import _native as nat, pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402
from soak_cases import make_case  # noqa: E402

maxp = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 32
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nat.load(); pi.VERBOSE = False
faulthandler.dump_traceback_later(60, repeat=True, file=sys.stderr)
cases = [make_case(9000 + k, maxp) for k in range(per)]
for k, c in enumerate(cases):
    print("pair %d: N=%d M=%d lattice=%s" % (k, c[0].shape[1], c[1].shape[1], c[2]), flush=True)
kw = dict(transform=os.environ.get("PM_SOAK_TRANSFORM", "Affine"), ransac_trials=120, ransac_error=20.0, icp_iterations=6)
t0 = time.perf_counter()
timings, reports = {}, {}
out = P.estimate_transform_batch([(c[0], c[1]) for c in cases], workers=workers, seeds=[c[4] for c in cases], timings=timings, reports=reports, **kw)
print("batch done in %.1f s" % (time.perf_counter() - t0), flush=True)
for k in sorted(timings):
    print(k, {a: round(b, 2) for a, b in timings[k].items()}, reports.get(k, {}).get("routes"), flush=True)
