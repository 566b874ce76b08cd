#!/usr/bin/env python3
"""One rank's share of the multi-GPU runs, measured on ONE GPU (the 8-GPU node is the driver's: VERDICT r03 next #4b), so that
the driver's SCALE curve has a prediction to be compared with.

  strong scaling of the bench step (python bench.py --gpus G, N = M = 50 000): for G = 1, 2, 4, 8 what rank 0 executes —
    its pieces of both clouds' mean-distance sums + the ordered finish, centroid and axis (replicated), descriptors of its
    N/G moving and M/G fixed rows, the symmetry flag, the eight cost matrices of its N/G rows against all M columns, the 200
    ICP iterations (replicated below pipeline.ICP_SHARD_MIN_POINTS) — each timed with HIP events, no collective executed;
    the two exchanges are added from a model: all-reduce of the piece sums (2 x 1.2 MB) and all-gather of the fixed cloud's
    frame-1 descriptors (144 MB in all), at XGMI_GBS per link, ring schedule (G - 1 steps of one block) and, beside it, the
    direct one-shot schedule a full mesh allows (every block on its own link);
  config 4 (200 000 x 200 000 on 8 GPUs): one rank's 25 000 x 200 000 x 8 cost rows in slabs + the row arg-mins, with the
    statistics share and its descriptor rows; exchange: 576 MB of frame-1 fixed descriptors.

Writes profiles/r04_rank_share.json (or argv[1]).  Usage: python tools/rank_share.py [out.json] [--no-c4]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat, pipeline as P  # noqa: E402

XGMI_GBS = 153.0          # per link and direction (MI355X_MICROARCH.md: 7 links per GPU)
LATENCY_US = 20.0         # per collective step (launch + handshake), order of magnitude
dev = torch.device("cuda:0")
nat.load()
out_path = next((a for a in sys.argv[1:] if not a.startswith("--")), os.path.join(ROOT, "profiles", "r04_rank_share.json"))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def exchange_ms(total_bytes, G, steps_latency=True):
    """all-gather of total_bytes (every rank contributes total / G): ring = (G - 1) steps of one block over one link;
    direct = every peer's block on its own link at once (G - 1 <= 7 links)."""
    if G == 1:
        return 0.0, 0.0
    block = total_bytes / G
    ring = (G - 1) * (block / (XGMI_GBS * 1e9) * 1e3 + LATENCY_US * 1e-3)
    direct = block / (XGMI_GBS * 1e9) * 1e3 + LATENCY_US * 1e-3
    return ring, direct


def bench_share(n, G, icp_iters=200):
    mv_h, fx_h, start_h = bench.synth(n)
    mov, fix, start = (nat.to_dev(x, dev=dev) for x in (mv_h, fx_h, start_h))
    be = P.GpuBackend(dev)
    rows = P.shard_bounds(n, G)[1]
    U = torch.empty((8, rows, n), dtype=torch.float64, device=dev)
    icp_ws = nat.workspace(nat.load().pm_icp_workspace(n, n), dev)
    st = {}

    def stats():
        for x in (mov, fix):
            c, a = be.centroid_and_axis(x)
            part = be.mean_distance_partials(x, 0, G)
            st[id(x)] = (c, be.mean_distance_finish(part, n), a)
    t_stats = timed(stats)
    # (what was timed holds only this rank's pieces of the pair sums — the all-reduce is not executed here —, so the values the
    # later stages work with are taken from the one-GPU kernels: identical to what the all-reduce + finish would deliver)
    (cm, mdm, x0m), (cf, mdf, x0f) = be.stats(mov), be.stats(fix)
    sc = {}

    def desc():
        sc["m"] = be.shape_context(mov, cm, mdm, x0m, 2, 0, rows)
        sc["f"] = be.shape_context(fix, cf, mdf, x0f, 4, 0, rows)
        sc["flag"] = be.symmetry_flag(sc["m"], sc["f"])
    t_desc = timed(desc)
    f1 = be.shape_context(fix, cf, mdf, x0f, 4, 0, n)[:1].contiguous()           # what the all-gather delivers
    t_cost = timed(lambda: K.chi2_cost8_frame1(sc["m"][0], f1[0], out=U))
    t_icp = timed(lambda: K.icp(start.clone(), fix, icp_iters, ws=icp_ws))
    ar_ring, ar_direct = exchange_ms(2 * 2 * (n * (n - 1) // 2 // 8192 + 1) * 8, G)      # two all-reduces ~ reduce-scatter + all-gather
    ag_ring, ag_direct = exchange_ms(n * 360 * 8, G)
    compute = t_stats + t_desc + t_cost + t_icp
    del U
    torch.cuda.empty_cache()
    return dict(G=G, rows=rows, statistics_ms=t_stats, descriptors_ms=t_desc, cost_ms=t_cost, icp_ms=t_icp, compute_ms=compute,
                exchange_ring_ms=ar_ring + ag_ring, exchange_direct_ms=ar_direct + ag_direct,
                step_ring_ms=compute + ar_ring + ag_ring, step_direct_ms=compute + ar_direct + ag_direct)


def c4_share(n=200_000, G=8, slab=4096):
    mv, fx, _ = bench.synth(n, seed=4)
    mov, fix = nat.to_dev(mv, dev=dev), nat.to_dev(fx, dev=dev)
    be = P.GpuBackend(dev)
    rows = n // G
    st = {}

    def stats():
        for x in (mov, fix):
            c, a = be.centroid_and_axis(x)
            st[id(x)] = (c, be.mean_distance_finish(be.mean_distance_partials(x, 0, G), n), a)
    t_stats = timed(stats, reps=1)
    (cm, mdm, x0m), (cf, mdf, x0f) = be.stats(mov), be.stats(fix)          # (the true values: see bench_share)
    sc = {}

    def desc():
        sc["m"] = be.shape_context(mov, cm, mdm, x0m, 2, 0, rows)
        sc["f"] = be.shape_context(fix, cf, mdf, x0f, 4, 0, rows)
    t_desc = timed(desc, reps=1)
    f1 = be.shape_context(fix, cf, mdf, x0f, 4, 0, n)[:1].contiguous()
    buf = torch.empty((8, slab, n), dtype=torch.float64, device=dev)
    idx = torch.empty((8, rows), dtype=torch.int32, device=dev)

    def cost():
        for s0 in range(0, rows, slab):
            s1 = min(rows, s0 + slab)
            Ub = K.chi2_cost8_frame1(sc["m"][0, s0:s1].contiguous(), f1[0], out=buf[:, :s1 - s0])
            idx[:, s0:s1] = K.row_argmin(Ub)
    t_cost = timed(cost, reps=1)
    ag_ring, ag_direct = exchange_ms(n * 360 * 8, G)
    return dict(n=n, G=G, rows=rows, slab_rows=slab, statistics_ms=t_stats, descriptors_ms=t_desc, cost_rows_and_argmin_ms=t_cost,
                point_pairs_per_s_per_rank=rows * n / (t_cost * 1e-3), exchange_ring_ms=ag_ring, exchange_direct_ms=ag_direct,
                rank_total_ms=t_stats + t_desc + t_cost + ag_ring,
                aggregate_point_pairs_per_s=n * float(n) / ((t_stats + t_desc + t_cost + ag_ring) * 1e-3))


res = {"xgmi_gb_per_s_per_link": XGMI_GBS, "latency_us_per_step": LATENCY_US, "bench_50k": []}
for G in (1, 2, 4, 8):
    r = bench_share(50000, G)
    res["bench_50k"].append(r)
    print(json.dumps(r), flush=True)
one = res["bench_50k"][0]["step_ring_ms"]
for r in res["bench_50k"]:
    r["predicted_speedup_ring"] = one / r["step_ring_ms"]
    r["predicted_speedup_direct"] = one / r["step_direct_ms"]
    print("G = %d: predicted speed-up %.2f (ring) / %.2f (direct); replicated part (statistics finish + ICP) %.1f ms of %.1f"
          % (r["G"], r["predicted_speedup_ring"], r["predicted_speedup_direct"], r["icp_ms"], r["step_ring_ms"]), flush=True)
if "--no-c4" not in sys.argv:
    res["config4_200k_on_8"] = c4_share()
    print(json.dumps(res["config4_200k_on_8"]), flush=True)
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
print("wrote", out_path)
