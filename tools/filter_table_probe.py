#!/usr/bin/env python3
"""The float32 filter build with and without its term table: launch time by HIP events at N = M points of
the bench's clouds, the largest deviation from the exact costs on a row block, and the packed-float32 / HBM-write view of the launch.
Usage: python tools/filter_table_probe.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat, pipeline as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
mv, fx, _ = bench.synth(n)
be = P.GpuBackend()
sc_m, sc_f, _ = P.build_descriptors(be, be.cloud(mv), be.cloud(fx))
assert K.chi2_symmetric(sc_m, sc_f)
a1, b1 = sc_m[0], sc_f[0]
out = torch.empty((4, n, n), dtype=torch.float32, device=a1.device)
res = {}
ref = None
for variant, name in ((0, "every shell computed (round 4)"), (1, "term table for sparsely filled shells (round 5)"),
                      (2, "table kernel at <= 128 registers"), (3, "the same, table ruled out")):
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.chi2_filter4(a1, b1, out=out, variant=variant)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    rows = min(n, 2048)
    exact = K.chi2_cost8(sc_m[:, :rows].contiguous(), sc_f)
    worst = max(float((out[t, :rows].double() - exact[k]).abs().max()) for t, pair in enumerate(K.PAIRINGS) for k in pair)
    ms = float(np.median(ts[1:]))
    res[variant] = ms
    if variant == 0:
        ref = out[:, :64].clone()
    else:
        name += " [differs from variant 0 on %.0f %% of the first 64 rows' entries]" % (100.0 * float((out[:, :64] != ref).double().mean()))
    flop = 5.0 * 4 * 360 * float(n) * n          # per (pair, bin, pairing): add, multiply, reciprocal, fused multiply-add
    print("variant %d, %-48s launches incl. pre-passes %s ms -> median %.1f ms; largest |filter - exact| on %d rows %.2e (bound %.1e); "
          "%.1f TFLOP/s of the algorithmic 5 flop per term (%.1f %% of the 157 TFLOP/s packed-float32 peak); %.0f GB/s of the %.0f GB written"
          % (variant, name + ":", [round(t, 1) for t in ts], ms, rows, worst, K.chi2_filter_delta(), flop / ms / 1e9, flop / ms / 1e9 / 157.0 * 100,
             16.0 * n * n / ms / 1e6, 16.0 * n * n / 1e9), flush=True)
print("N = M = %d: table %.1f ms against %.1f ms = %.2fx" % (n, res[1], res[0], res[0] / res[1]))
