#!/usr/bin/env python3
"""Cost build + eight assignments, back to back against pipelined: does starting a pairing's assignment while the later pairings
are still being built pay once the solver's streams have PRIORITY over the cost kernel's remaining workgroups?
Usage: python tools/pipelined_assign_probe.py [N] [repeats]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import _kernels as K  # noqa: E402
from platymatch_amd import lsap  # noqa: E402
from platymatch_amd import pipeline as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mv, fx, _ = synth_pair(n, 42)
be = P.GpuBackend()
mov, fix = be.cloud(mv), be.cloud(fx)
sc_m, sc_f, bn = P.build_descriptors(be, mov, fix)
sc_m1, sc_f1 = sc_m[0].contiguous(), sc_f[0].contiguous()
U = torch.empty((8, sc_m1.shape[0], sc_f1.shape[0]), dtype=torch.float64, device=mov.device)
torch.cuda.synchronize()


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "back to back":
        K.chi2_cost8_frame1(sc_m1, sc_f1, out=U)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = lsap.solve_eight_on_device(U)
    else:
        lsap.PIPELINED_PRIORITY = -1 if mode == "pipelined, priority" else 0
        _, ready = K.chi2_cost8_frame1_by_pairings(sc_m1, sc_f1, out=U)
        t1 = time.perf_counter()
        out = lsap.solve_eight_on_device(U, ready=ready)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3, out


ref = None
for mode in ("back to back", "pipelined", "pipelined, priority", "back to back"):
    for r in range(reps):
        a, b, out = run(mode)
        if ref is None:
            ref = out
        same = all(np.array_equal(x[1], y[1]) for x, y in zip(out, ref))
        print("%-22s build (or its launches) %8.1f ms   build + 8 assignments %8.1f ms   same answers: %s" % (mode, a, b, same), flush=True)
