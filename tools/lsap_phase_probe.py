#!/usr/bin/env python3
"""Where one hypothesis's assignment spends its wall clock, phase by phase (device passes incl. their read-back, the host core's
auction / shortest paths / repricing, the certificate, the uniqueness check), for a right and a wrong hypothesis.
Both drivers: the Python one (lsap.solve_core / certify, ~60 small calls per hypothesis, timed call by call) and the native one
(pm_lsap_solve_resident / pm_lsap_certify_resident, one foreign call each, its own phase clock), then all eight hypotheses on
four threads with either.  Usage: python tools/lsap_phase_probe.py N [N ...]"""
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

sizes = [int(a) for a in sys.argv[1:]] or [5000]
seed = 42
be = P.GpuBackend()
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
for n in sizes:
    mv, fx, _ = synth_pair(n, seed)
    U, _ = P.build_costs(be, be.cloud(mv), be.cloud(fx))
    torch.cuda.synchronize()
    for native in (False, True):
        L.NATIVE_DRIVER = native
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
                print("n = %d, hypothesis %d, %s driver: core %.1f ms + certify %.1f ms = %.1f ms; certified unique: %s; pricing rounds %s, bids %s, steps %s"
                      % (n, h, "native" if native else "Python", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, ok, info.get("rounds"),
                         info.get("auction_bids"), info.get("steps")))
                if native:
                    print("    auction (host) %.2f ms, shortest paths (host) %.2f ms, dense passes + copies (device) %.2f ms, certificate %.2f ms, rest %.2f ms"
                          % (info["auction_seconds"] * 1e3, info["core_seconds"] * 1e3, info["device_seconds"] * 1e3, info["certify_seconds"] * 1e3,
                             (t2 - t0 - info["auction_seconds"] - info["core_seconds"] - info["device_seconds"] - info["certify_seconds"]) * 1e3))
                    continue
                tot = 0.0
                for k in sorted(acc, key=lambda k: -acc[k]):
                    print("    %-28s %3d calls %8.2f ms" % (k, calls[k], acc[k] * 1e3))
                    tot += acc[k]
                print("    %-28s           %8.2f ms" % ("(python / numpy between)", (t2 - t0 - tot) * 1e3))
        for rep in range(3):
            t0 = time.perf_counter()
            out = L.solve_eight_on_device(U)
            print("n = %d, all eight through solve_eight_on_device, %s driver: %.1f ms" % (n, "native" if native else "Python", (time.perf_counter() - t0) * 1e3), flush=True)
    del U
    torch.cuda.empty_cache()
