#!/usr/bin/env python3
"""Where the first 50k cost build's wall time goes in a fresh process: descriptors, the 160 GB allocation, the launch.  Tools only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import _kernels as K, pipeline as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
mv, fx, _ = synth_pair(n, 42)
be = P.GpuBackend()
mov, fix = be.cloud(mv), be.cloud(fx)
P.build_costs(be, mov[:, :256].contiguous(), fix[:, :256].contiguous())
torch.cuda.synchronize()


def t(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-50s call %8.1f ms   + wait %8.1f ms" % (label, (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3), flush=True)
    return out


for rep in range(2):
    sc_m, sc_f, bn = t("descriptors", lambda: P.build_descriptors(be, mov, fix))
    sym = t("symmetry check", lambda: K.chi2_symmetric(sc_m, sc_f))
    U = t("torch.empty [8, n, n]", lambda: torch.empty((8, n, n), dtype=torch.float64, device=mov.device))
    t("chi2_cost8_frame1", lambda: K.chi2_cost8_frame1(sc_m[0], sc_f[0], out=U))
    t("chi2_cost8_frame1 again", lambda: K.chi2_cost8_frame1(sc_m[0], sc_f[0], out=U))
    del U, sc_m, sc_f
    print("--", flush=True)
