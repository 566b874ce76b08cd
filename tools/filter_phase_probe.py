#!/usr/bin/env python3
"""A complete registration in the default cost mode (the filter route from 8 192 nuclei), three calls, and where each pairing's
solver thread spent its wall clock in the last one: every query of the filtered matrix (selection on the float32 matrix + exact
evaluation of the selected entries + read-back), the host core's auction / shortest paths / repricing, the listed certificate.
The four pairings run side by side on four host threads, so a pairing's dense passes also wait for the other three's.
Usage: python tools/filter_phase_probe.py [N] [--alone]      (--alone: the four pairings one after the other, for the contention-free split)"""
import os
import sys
import threading
import time
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import lsap as L, pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

pi.VERBOSE = False
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 50000
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))


def timed(cls, name, label=None):
    f = getattr(cls, name)
    label = label or "%s.%s" % (cls.__name__, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            who = threading.current_thread().name
            acc[who][label] += time.perf_counter() - t
            calls[who][label] += 1
    setattr(cls, name, g)


for nm in ("row_select", "threshold_select", "certificate", "col_min", "diagonal"):
    timed(L.FilteredMatrix, nm)
for nm in ("add", "init_duals", "auction", "solve", "reprice", "get"):
    timed(L._Core, nm)
_certify_listed = L.certify_listed


def certify_listed(*a, **k):
    t = time.perf_counter()
    try:
        return _certify_listed(*a, **k)
    finally:
        who = threading.current_thread().name
        acc[who]["certify_listed (incl. its listing pass)"] += time.perf_counter() - t
        calls[who]["certify_listed (incl. its listing pass)"] += 1


L.certify_listed = certify_listed
if "--alone" in sys.argv:
    import concurrent.futures as cf
    _TPE = cf.ThreadPoolExecutor

    class Serial(_TPE):
        def __init__(self, max_workers=None, **kw):
            super().__init__(max_workers=1, **kw)
    L.ThreadPoolExecutor = Serial

mv, fx, _ = synth_pair(n, 42)
for rep in range(3):
    acc.clear()
    calls.clear()
    det = {"timing": True}
    t = time.perf_counter()
    P.estimate_transform(mv, fx, ransac_trials=8000, ransac_error=16, icp_iterations=50, details=det)
    torch.cuda.synchronize()
    print("N = M = %d, registration %d: wall %.1f ms, stages %s" % (n, rep + 1, (time.perf_counter() - t) * 1e3, {k: round(v, 3) for k, v in det["timing"].items()}), flush=True)
print("cost mode %s; per pairing (solver threads%s):" % (det["assignment"].get("cost_mode"), " one after the other" if "--alone" in sys.argv else " side by side"))
for h in range(4):
    d = det["assignment"]["details"][h]
    print("  pairing %d: %s" % (h, {k: (round(d[k], 4) if isinstance(d[k], float) else d[k]) for k in
                                   ("rounds", "steps", "augmentations", "auction_bids", "auction_seconds", "core_seconds", "polish_seconds", "polish_list_shape", "polish_listed",
                                    "polish_violated", "violated_per_round", "auction_violated", "listed", "tight_within_eps") if k in d}))
for who in sorted(acc):
    tot = sum(acc[who].values())
    print("  thread %s: %.1f ms in timed calls" % (who, tot * 1e3))
    for k in sorted(acc[who], key=lambda k: -acc[who][k]):
        print("      %-44s %3d calls %8.2f ms" % (k, calls[who][k], acc[who][k] * 1e3))
