#!/usr/bin/env python3
"""The term-table variants of the half-cost chi-square kernel (pm_chi2_cost8_sym_ws_variant) against the plain kernel:
bit equality and time, on a row block of an N-point synthetic pair (the counts per shell depend on N, the time on the rows).
Tools only.  Usage: python tools/chi2_table_probe.py [N] [rows] [variants...]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
variants = [int(v) for v in sys.argv[3:]] or [2, 0, 1, 3, 4]
lib = nat.load()
fn = lib.pm_chi2_cost8_sym_ws_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
               ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")
kind = os.environ.get("PM_CLOUD", "normal")
if kind == "uniform":
    mvh, fxh = rng.random((3, n)) * 400.0, rng.random((3, n)) * 400.0
else:
    mvh = rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0
    fxh = rng.normal(size=(3, n)) * np.array([[50.0], [45.0], [30.0]]) + 100.0
mv, fx = nat.to_dev(mvh, dev=dev), nat.to_dev(fxh, dev=dev)
hm = K.shape_context(mv, K.centroid(mv), K.pca_axis(mv), K.mean_distance(mv), 2)["hist"]
hf = K.shape_context(fx, K.centroid(fx), K.pca_axis(fx), K.mean_distance(fx), 4)["hist"]
cm = torch.round(hm[0] * (n - 1)).reshape(n, 30, 12).amax(dim=(0, 2)).cpu().numpy().astype(int)
cf = torch.round(hf[0] * (n - 1)).reshape(n, 30, 12).amax(dim=(0, 2)).cpu().numpy().astype(int)
print("largest count per (r, theta) shell, moving:", cm.tolist())
print("largest count per (r, theta) shell, fixed: ", cf.tolist())
for tl in (48, 64, 88):
    print("shells with all counts < %d: %d of 30" % (tl, int(((cm < tl) & (cf < tl)).sum())))
ws_bytes = lib.pm_chi2_sym_workspace_bytes(rows, n)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
out = torch.empty((8, rows, n), dtype=torch.float64, device=dev)
a = hm[0][:rows].contiguous()
ref = K.chi2_cost8_frame1(a, hf[0])
times = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        out.fill_(-1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(a.data_ptr(), rows, hf[0].data_ptr(), n, out.data_ptr(), n, rows * n, ws.data_ptr(), ws_bytes, v, 0)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        if rnd:
            times[v].append(e0.elapsed_time(e1))
        else:
            assert torch.equal(out, ref), "variant %d differs from the plain kernel" % v
            meta = ws[:512].cpu().numpy()
            print("variant %d: identical bits; meta: tot %s bad %d" % (v, np.frombuffer(meta[16:32].tobytes(), dtype=np.float64).tolist(),
                                                                       int(np.frombuffer(meta[32:36].tobytes(), dtype=np.int32)[0])), flush=True)
for v in variants:
    t = np.array(times[v])
    print("variant %d: median %.2f ms  min %.2f ms   (%.3e point-pairs/s)" % (v, np.median(t), t.min(), rows * n / (np.median(t) * 1e-3)), flush=True)
