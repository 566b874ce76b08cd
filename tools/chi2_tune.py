#!/usr/bin/env python3
"""Time the launch-shape variants of the half-cost chi-square kernel against each other (interleaved rounds, one
process, random data) — tools only, not part of the product.  Usage: python tools/chi2_tune.py [N] [variants...]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variants = [int(v) for v in sys.argv[2:]] or [0, 1, 2, 3, 4, 5, 6]
lib = nat.load()
fn = lib.pm_chi2_cost8_sym_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
               ctypes.c_int, ctypes.c_void_p]
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")
mv = nat.to_dev(rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0, dev=dev)
fx = nat.to_dev(rng.normal(size=(3, n)) * np.array([[50.0], [45.0], [30.0]]) + 100.0, dev=dev)
hm = K.shape_context(mv, K.centroid(mv), K.pca_axis(mv), K.mean_distance(mv), 2)["hist"]
hf = K.shape_context(fx, K.centroid(fx), K.pca_axis(fx), K.mean_distance(fx), 4)["hist"]
out = torch.empty((8, n, n), dtype=torch.float64, device=dev)
ref = None
times = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(hm[0].data_ptr(), n, hf[0].data_ptr(), n, out.data_ptr(), n, n * n, v, 0)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        if rnd:
            times[v].append(e0.elapsed_time(e1))
        if ref is None:
            ref = out.clone()
        elif rnd == 0:
            assert torch.equal(out, ref), "variant %d differs" % v
for v in variants:
    t = np.array(times[v])
    print("variant %d: median %.2f ms  min %.2f ms   (%.3e pair-matrices/s)" % (v, np.median(t), t.min(), 8.0 * n * n / (np.median(t) * 1e-3)))
