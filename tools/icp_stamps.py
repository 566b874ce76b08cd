#!/usr/bin/env python3
"""Where an ICP iteration's microseconds go: a DIAGNOSTIC build of the library (-DPM_ICP_STAMPS: lane 0 of every workgroup
stamps the 100 MHz clock at the phase boundaries of icp_iter_kernel) and the stamps of the last iteration of a run.
Usage: python tools/icp_stamps.py [n] [iters]      (the product library is untouched; the diagnostic one lands in tools/microbench/_build)"""
import ctypes
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platymatch_amd import build as B  # noqa: E402

out_dir = os.path.join(ROOT, "tools", "microbench", "_build")
os.makedirs(out_dir, exist_ok=True)
B.build_native()
hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
obj = os.path.join(out_dir, "pm_icp_grid_stamps.o")
extra = os.environ.get("PM_EXTRA_DEFINES", "").split()
print("diagnostic build with", ["-DPM_ICP_STAMPS"] + extra, flush=True)
subprocess.check_call([hipcc] + B.FLAGS + ["-DPM_ICP_STAMPS"] + extra + ["-c", os.path.join(B.CSRC, "pm_icp_grid.hip"), "-o", obj])
objs = [os.path.join(B.OBJ, s.replace(".hip", ".o").replace(".cpp", ".o")) for s in B.SOURCES if s != "pm_icp_grid.hip"] + [obj]
lib_path = os.path.join(out_dir, "libplatymatch_stamps.so")
subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", lib_path] + objs)

import torch  # noqa: E402
import bench  # noqa: E402
from platymatch_amd import _native as nat  # noqa: E402
nat.LIB_PATH = lib_path
from platymatch_amd import _kernels as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = nat.load()
lib.pm_debug_set_icp_stamps.restype = ctypes.c_int
lib.pm_debug_set_icp_stamps.argtypes = [ctypes.c_void_p]
mv, fx, start = bench.synth(n)
dev = torch.device("cuda:0")
fix, st = nat.to_dev(fx, dev=dev), nat.to_dev(start, dev=dev)
ws = nat.workspace(lib.pm_icp_workspace(n, n), dev)
one = os.environ.get("PM_STAMPS_LOOP", "1") == "1"
K.icp(st.clone(), fix, iters, ws=ws, one_launch=one)
torch.cuda.synchronize()
rows = n // 8 + 2
stamps = torch.zeros((rows, 8), dtype=torch.int64, device=dev)
assert lib.pm_debug_set_icp_stamps(stamps.data_ptr()) == 0
K.icp(st.clone(), fix, iters, ws=ws, one_launch=one)
torch.cuda.synchronize()
assert lib.pm_debug_set_icp_stamps(None) == 0
best = 1e9
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w_ = st.clone()
    e0.record()
    K.icp(w_, fix, 200, ws=ws, one_launch=one)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("TIMING n=%d defines=%s: %.1f us per iteration (200 iterations, stamps off)" % (n, " ".join(extra) or "-", best * 1e3 / 200), flush=True)
few_from = 32768
for d in extra:
    if d.startswith("-DPM_GR_FEW_FROM="):
        few_from = int(d.split("=")[1])
lanes = 4 if n >= few_from else 8
for d in extra:
    if d.startswith("-DPM_GR_TRY_LANES=") and n >= few_from:
        lanes = int(d.split("=")[1])
blocks = (n + 256 // lanes - 1) // (256 // lanes)
t = stamps.cpu().numpy()[:blocks].astype(np.float64)
t0 = t[:, 0].min()
loop = os.environ.get("PM_STAMPS_LOOP", "1") == "1" and iters >= 3 and blocks <= 1024
if loop:
    names = ["round starts (last iteration of the one-launch loop)", "-", "search done", "leaf partials stored + drained", "arrival add returned",
             "group finishers: tail done (last of all: solved + published)", "generation seen, barrier passed", "transform reloaded (round ends)"]
else:
  names = ["entry", "point, transform, previous match loaded", "search done", "leaf partials stored + drained", "arrival add returned",
         "group: partials fetched, summed, stored + drained", "group arrival add returned", "total fetched, solved, stored (end)"]
print("n = %d, %d lanes per point, %d workgroups; last of %d iterations; microseconds after the first workgroup's entry" % (n, lanes, blocks, iters))
print("%-52s %6s %8s %8s %8s" % ("stamp", "WGs", "min", "median", "max"))
for k, name in enumerate(names):
    v = t[:, k]
    v = (v[v >= t0] - t0) / 100.0            # 100 MHz -> us
    if len(v):
        print("%-52s %6d %8.2f %8.2f %8.2f" % (name, len(v), v.min(), np.median(v), v.max()))
