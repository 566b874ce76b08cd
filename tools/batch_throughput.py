#!/usr/bin/env python3
"""Aggregate registrations/s of a batch of independent specimen pairs (BASELINE config 5: 64 pairs of mixed 2k-20k nuclei,
"replicas only" -- pairs never exchange data).  Each pair is a complete unsupervised registration as the widget runs it
(_dock_widget.py:526-718): statistics, descriptors, eight cost matrices, eight Hungarian solves, 8 x 8000 RANSAC trials,
50 ICP iterations.

One GPU:     python tools/batch_throughput.py [--pairs 64] [--workers 4] [--json profiles/r02_batch64.json]
Several:     python -m torch.distributed.run --nproc-per-node G ... tools/batch_throughput.py ...   (pairs dealt to ranks
             largest first, pipeline.estimate_transform_batch(group=...); one rank per GPU over RCCL)
Sizes: rng.integers(2000, 20001) per pair, default_rng(5); --max-points caps them for a short run."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--min-points", type=int, default=2000)
    ap.add_argument("--max-points", type=int, default=20000)
    ap.add_argument("--trials", type=int, default=8000)
    ap.add_argument("--icp", type=int, default=50)
    ap.add_argument("--json", default=None)
    ap.add_argument("--sequential-too", action="store_true", help="also time workers=1 (one GPU only)")
    ap.add_argument("--cost-mode", default="auto", choices=("auto", "exact", "relaxed", "filter"), help="estimate_transform(cost_mode=...)")
    ap.add_argument("--unseeded", action="store_true", help="no RANSAC seeds: index sets drawn on the device (the default for callers who do not seed)")
    args = ap.parse_args()
    pi.VERBOSE = False
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    group = None
    if world > 1:
        import torch.distributed as dist
        # rehearsal switches (as in bench.py): PM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, PM_BENCH_BACKEND=gloo replaces RCCL
        torch.cuda.set_device(0 if os.environ.get("PM_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("PM_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        group = dist.group.WORLD
    sizes = [int(x) for x in np.random.default_rng(5).integers(args.min_points, args.max_points + 1, size=args.pairs)]
    pairs, truth = [], []
    for k, n in enumerate(sizes):
        mv, fx, A = synth_pair(n, 100 + k)
        pairs.append((mv, fx))
        truth.append(A)
    P.estimate_transform(pairs[0][0][:, :300], pairs[0][1][:, :300], ransac_trials=100, icp_iterations=2)      # warm-up
    runs = []
    import threading
    t_start = time.perf_counter()
    stop = threading.Event()

    def heartbeat():                 # a long batch must show signs of life (the GPU box kills silent commands)
        while not stop.wait(60.0):
            print("[rank %d] %.0f s elapsed" % (rank, time.perf_counter() - t_start), flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    for w in ([1] if (args.sequential_too and world == 1) else []) + [args.workers]:
        torch.cuda.synchronize()
        timings = {}
        t = time.perf_counter()
        out = P.estimate_transform_batch(pairs, workers=w, seeds=None if args.unseeded else list(range(len(pairs))), group=group, timings=timings,
                                         ransac_trials=args.trials, ransac_error=16, icp_iterations=args.icp, cost_mode=args.cost_mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        err = max(np.linalg.norm(np.asarray(o[1]) @ np.asarray(o[0]) - A) / np.linalg.norm(A) for o, A in zip(out, truth))
        split = {}
        for tm in timings.values():
            for name, v in tm.items():
                split[name] = split.get(name, 0.0) + v
        if group is not None:                 # every rank must hold every result: compare with rank 0's copy
            import torch.distributed as dist
            flat = torch.as_tensor(np.concatenate([np.concatenate([np.asarray(o[0]).ravel(), np.asarray(o[1]).ravel(), np.asarray(o[2], dtype=np.float64)])
                                                   for o in out]))
            ref = flat.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(flat, ref), "ranks disagree on the batch results"
        run = {"cost_mode": args.cost_mode, "workers_per_gpu": w, "n_gpus": world, "ransac_index_sets": "device sampler (unseeded)" if args.unseeded else "NumPy stream (seeded)", "pairs": len(pairs), "seconds": dt, "registrations_per_s": len(pairs) / dt,
               "worst_rel_error_vs_ground_truth": err,
               "stage_seconds_summed_over_this_ranks_pairs": split,
               "per_pair": [{"pair": k, "n": sizes[k], **{a: round(b, 4) for a, b in timings[k].items()}} for k in sorted(timings)]}
        runs.append(run)
        if rank == 0:
            print("gpus=%d workers=%d: %d pairs (%d..%d points) in %.2f s -> %.3f registrations/s; worst rel. error vs ground truth %.1e; "
                  "stage seconds (this rank) %s" % (world, w, len(pairs), min(sizes), max(sizes), dt, len(pairs) / dt, err,
                                                    {a: round(b, 2) for a, b in split.items()}), flush=True)
    stop.set()
    if rank == 0 and args.json:
        with open(args.json, "w") as f:
            json.dump({"workload": "BASELINE configs[4]: %d pairs, sizes default_rng(5).integers(%d, %d), complete unsupervised "
                                   "registration each (8 x %d RANSAC trials, %d ICP iterations)"
                                   % (args.pairs, args.min_points, args.max_points + 1, args.trials, args.icp),
                       "sizes": sizes, "host_cores": os.cpu_count(), "runs": runs}, f, indent=1)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
