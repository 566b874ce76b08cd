#!/usr/bin/env python3
"""The FIRST registration of a fresh process against the following ones (the reference's user clicks "Run"
once — the cold number is their number).  One process measures one cold start, so run it once per mode:
    python tools/cold_start.py N [auto|exact|relaxed|filter] [--reserve]
--reserve: platymatch_amd.reserve(N, N, mode) first (timed separately) — what a caller who knows the sizes can do ahead of time.
Prints the wall clock of import + library load, of each of four registrations (unseeded, 8 x 8 000 RANSAC trials, 50 ICP
iterations) with its stage split, and the kept buffer's size."""
import os
import sys
import time

t_proc = time.perf_counter()
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
mode = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "auto"
from conftest import synth_pair  # noqa: E402
mv, fx, A_gt = synth_pair(n, 42)
t0 = time.perf_counter()
import torch  # noqa: E402
import platymatch_amd  # noqa: E402
from platymatch_amd import _native as nat, pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402
nat.load()
pi.VERBOSE = False
print("N = M = %d, cost_mode=%r; import torch + platymatch_amd + library load: %.2f s" % (n, mode, time.perf_counter() - t0), flush=True)
if "--reserve" in sys.argv:
    t0 = time.perf_counter()
    kept = platymatch_amd.reserve(n, n, mode)
    torch.cuda.synchronize()
    print("platymatch_amd.reserve(%d, %d, %r): %.2f s, %.1f GB kept" % (n, n, mode, time.perf_counter() - t0, kept / 1e9), flush=True)
if "--torch-warm" in sys.argv:
    # hypothesis test: how much of the first assignment stage is the first use of torch's own kernels (selection, sorting, indexing)?
    t0 = time.perf_counter()
    d = torch.device("cuda", torch.cuda.current_device())
    x = torch.arange(4096, dtype=torch.float64, device=d).view(64, 64)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    o = torch.argsort(torch.where(x > 3, x, torch.full_like(x, float("inf"))), dim=1, stable=True)
    g = torch.gather(x, 1, o)
    c = torch.bincount(o.reshape(-1), minlength=64)
    s_ = torch.cumsum(c, 0) - c
    z = torch.cat([g.reshape(-1), x.reshape(-1)])[x.reshape(-1).repeat(2) > 1]
    f = torch.stack([(~torch.isfinite(z)).any().to(torch.int32), (s_ > 0).any().to(torch.int32)]).cpu()
    y = torch.full((8, 8), -1, dtype=torch.int32, device=d)
    y[o[:8, 0].long().clamp(max=7), o[:8, 1].long().clamp(max=7)] = 1
    torch.cuda.synchronize()
    print("torch warm-up: first tensor %.3f s, first synchronize %.3f s, the op set %.3f s" % (t1 - t0, t2 - t1, time.perf_counter() - t2), flush=True)
for rep in range(4):
    det = {"timing": True}
    t0 = time.perf_counter()
    A_sc, A_icp, inl = P.estimate_transform(mv, fx, ransac_trials=8000, ransac_error=16, icp_iterations=50, details=det, cost_mode=mode)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    err = np.linalg.norm(A_icp @ A_sc - A_gt) / np.linalg.norm(A_gt)
    print("registration %d (%s): %7.3f s   rel. error %.1e  stages %s  kept buffer %.1f GB"
          % (rep + 1, "COLD: first call of the process" if rep == 0 else "warm", wall, err,
             {k: round(v, 3) for k, v in det["timing"].items()}, P.kept_cost_bytes(torch.device("cuda", torch.cuda.current_device())) / 1e9), flush=True)
print("process total %.1f s" % (time.perf_counter() - t_proc))
