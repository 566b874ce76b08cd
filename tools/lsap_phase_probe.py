#!/usr/bin/env python3
"""Where one hypothesis's assignment spends its wall clock, phase by phase (device passes incl. their read-back, the host core's
auction / shortest paths / repricing, the certificate, the uniqueness check), for a right and a wrong hypothesis.
Usage: python tools/lsap_phase_probe.py N [SEED]"""
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import lsap as L, pipeline as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 42
mv, fx, _ = synth_pair(n, seed)
be = P.GpuBackend()
U, _ = P.build_costs(be, be.cloud(mv), be.cloud(fx))
torch.cuda.synchronize()
acc, calls = defaultdict(float), defaultdict(int)


def timed(cls, name, label=None):
    f = getattr(cls, name)
    label = label or "%s.%s" % (cls.__name__, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
            calls[label] += 1
    setattr(cls, name, g)


for nm in ("row_select", "diagonal", "col_min", "certificate", "entries"):
    timed(L.DeviceMatrix, nm)
for nm in ("add", "init_duals", "auction", "solve", "reprice", "get"):
    timed(L._Core, nm)
lib = L.nat.load()
for rep in range(2):
    for h in (0, 1):
        acc.clear(); calls.clear()
        info = {}
        t0 = time.perf_counter()
        W = L.DeviceMatrix(U[h])
        sol = L.solve_core(W, info)
        t1 = time.perf_counter()
        ok = sol is not None and L.certify(W, *sol, info=info)
        t2 = time.perf_counter()
        if rep == 0:
            continue                      # (first pass warms the allocator and the library)
        print("n = %d, hypothesis %d: core %.1f ms + certify %.1f ms = %.1f ms; certified unique: %s; pricing rounds %s, bids %s, steps %s"
              % (n, h, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, ok, info.get("rounds"), info.get("auction_bids"), info.get("steps")))
        tot = 0.0
        for k in sorted(acc, key=lambda k: -acc[k]):
            print("    %-28s %3d calls %8.2f ms" % (k, calls[k], acc[k] * 1e3))
            tot += acc[k]
        print("    %-28s           %8.2f ms" % ("(python / numpy between)", (t2 - t0 - tot) * 1e3))
t0 = time.perf_counter()
out = L.solve_eight_on_device(U)
print("all eight through solve_eight_on_device: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
