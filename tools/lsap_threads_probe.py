#!/usr/bin/env python3
"""Do the four pairings' core solves slow each other down when they run side by side?  The native driver's own phase clock
(pm_lsap_solve_resident's report) for hypotheses 0..3 solved alone, then on 2 and on 4 threads at once.
Usage: python tools/lsap_threads_probe.py N [N ...]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import lsap as L, pipeline as P, _native as nat  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [5000]
be = P.GpuBackend()
L.NATIVE_DRIVER = True
print("host cpus visible: %d (affinity %d)" % (os.cpu_count(), len(os.sched_getaffinity(0))))


def one(U, h):
    stream = nat.side_stream(U.device, ("probe", h))
    with torch.cuda.device(U.device), torch.cuda.stream(stream):
        info = {}
        t0 = time.perf_counter()
        W = L.DeviceMatrix(U[h])
        sol = L.solve_core(W, info)
        ok = sol is not None and L.certify(W, *sol, info=info)
        stream.synchronize()
        return h, (time.perf_counter() - t0) * 1e3, info, ok


for n in sizes:
    mv, fx, _ = synth_pair(n, 42)
    U, _ = P.build_costs(be, be.cloud(mv), be.cloud(fx))
    torch.cuda.synchronize()
    for threads in (1, 1, 2, 4):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            res = list(ex.map(lambda h: one(U, h), range(4)))
        wall = (time.perf_counter() - t0) * 1e3
        print("n = %d, hypotheses 0..3 on %d thread(s): wall %.1f ms" % (n, threads, wall))
        for h, ms, info, ok in res:
            print("    h%d %.1f ms: auction %.2f, shortest paths %.2f, device passes %.2f, certificate %.2f; bids %s, steps %s, ok %s"
                  % (h, ms, info["auction_seconds"] * 1e3, info["core_seconds"] * 1e3, info["device_seconds"] * 1e3,
                     info["certify_seconds"] * 1e3, info.get("auction_bids"), info.get("steps"), ok), flush=True)
    del U
    torch.cuda.empty_cache()
