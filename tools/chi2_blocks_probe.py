#!/usr/bin/env python3
"""Is the 50k cost launch linear in its rows, or does its 160 GB footprint / its second-long duration cost something (TLB reach,
clocks)?  (A first rank-share measurement suggested 13x for an eighth of the rows; it was an artefact of that tool — wrong ring
radii, hence emptier shells and more of them tabled.)  The same build as ONE launch and as B launches of N/B rows each, (a) into row slices of the one
[8, N, M] buffer (the big launch's footprint and matrix spacing, short launches) and (b) into a compact [8, N/B, M] buffer reused
by every block (small footprint).  HIP events per launch.  Usage: python tools/chi2_blocks_probe.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat, pipeline as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda:0")
mv_h, fx_h, _ = bench.synth(n)
mov, fix = nat.to_dev(mv_h, dev=dev), nat.to_dev(fx_h, dev=dev)
be = P.GpuBackend(dev)
sc_m, sc_f, _ = P.build_descriptors(be, mov, fix)
a, b = sc_m[0].contiguous(), sc_f[0].contiguous()
U = torch.empty((8, n, n), dtype=torch.float64, device=dev)


def run(blocks, compact):
    rows = (n + blocks - 1) // blocks
    small = torch.empty((8, rows, n), dtype=torch.float64, device=dev) if compact else None
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for k in range(blocks):
        r0, r1 = k * rows, min(n, (k + 1) * rows)
        out = small[:, :r1 - r0] if compact else U[:, r0:r1]
        K.chi2_cost8_frame1(a[r0:r1], b, out=out)
        evs[k + 1].record()
    torch.cuda.synchronize()
    per = [evs[k].elapsed_time(evs[k + 1]) for k in range(blocks)]
    return sum(per), per


run(1, False)
for blocks in (1, 2, 4, 8, 16, 32):
    for compact in (False, True):
        if blocks == 1 and compact:
            continue
        tot, per = run(blocks, compact)
        print("%2d launch(es) of %5d rows into %s: total %7.1f ms; per launch first %.1f, median %.1f, last %.1f"
              % (blocks, (n + blocks - 1) // blocks, "a compact buffer      " if compact else "slices of the full one", tot, per[0], sorted(per)[len(per) // 2], per[-1]), flush=True)
