#!/usr/bin/env python3
"""One pairing's filtered solve at N x N nuclei with a line per host-core call (auction, solve, reprice: seconds, bids / steps,
violations): does the pricing loop converge, and how fast?  Usage: python tools/solve_trace.py N [pairing] [seconds_budget]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import _kernels as K, lsap as L, pipeline as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
t = int(sys.argv[2]) if len(sys.argv) > 2 else 0
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
t_start = time.perf_counter()


def traced(cls, name, show):
    f = getattr(cls, name)

    def g(self, *a, **k):
        t0 = time.perf_counter()
        out = f(self, *a, **k)
        print("[%6.1f s] %-10s %7.3f s  %s" % (time.perf_counter() - t_start, name, time.perf_counter() - t0, show(self, out)), flush=True)
        if time.perf_counter() - t_start > budget:
            raise SystemExit("budget of %.0f s used up" % budget)
        return out
    setattr(cls, name, g)


traced(L._Core, "auction", lambda c, out: "bids %s" % out)
traced(L._Core, "solve", lambda c, out: "stats (edges, steps, augmentations, dummy scans) %s" % (c.get()[3],))
traced(L._Core, "reprice", lambda c, out: "violated rows %s" % out)
traced(L._Core, "add", lambda c, out: "")
mv, fx, _ = synth_pair(n, 42)
be = P.GpuBackend()
sc_m, sc_f, _ = P.build_descriptors(be, be.cloud(mv), be.cloud(fx))
a1, b1 = sc_m[0], sc_f[0]
F = K.chi2_filter_pair(a1, b1, t, dtype=torch.float32)
torch.cuda.synchronize()
print("N = M = %d, pairing %d: filter matrix built (%.1f GB), %.1f s" % (n, t, F.numel() * 4 / 1e9, time.perf_counter() - t_start), flush=True)
M = L.FilteredMatrix(L.DeviceMatrix(F), lambda r, c: tuple(x.cpu().numpy() for x in K.chi2_entries(a1, b1, t, r, c)), K.chi2_filter_delta() + 1e-13,
                     lambda r, c: K.chi2_entries(a1, b1, t, r, c, trusted=True))
info = {}
sol = L.solve_core(M, info)
print("solve_core done in %.1f s: %s" % (time.perf_counter() - t_start, {k: info[k] for k in ("rounds", "steps", "augmentations", "violated_per_round", "auction_violated", "polish_violated") if k in info}))
