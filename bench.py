#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X estimate_transform hot path.

Workload (BASELINE.json metric: point-pairs/s, shape-context + chi-square cost + ICP, N = 50k): one
"step" is one pass of the path over a synthetic pair of 50 000-point clouds already resident in HBM:
  cloud statistics -> shape-context descriptors (2 + 4 frames) -> the eight N x M chi-square cost
  matrices (float64, written to HBM) -> 200 iterations of affine ICP.
value = N*M point pairs / step time (whole job, all ranks).  With --gpus G > 1 the SAME problem is
row-sharded (strong scaling): descriptors and cost rows by block with one all-gather of the fixed
descriptors over RCCL; the ICP refinement (1 % of the step with the grid search) is run whole by every rank.  The Hungarian solve is not
part of the step: at 50k it needs hours and 20 GB of host memory per matrix.

Launch:  python bench.py [--gpus N] [--steps K] [--warmup W]     (N > 1 and no WORLD_SIZE in the environment: the script starts its
                                                                  own N ranks, one per GPU, before anything touches a GPU)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector float64 spec peak
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X vector float32 spec peak (packed: two lanes' worth per instruction)
FILTER_CYCLES_PER_TERM = (192 * 8 + 373 * 4) / 192.0   # filter4_kernel's issue cycles per wave64 term (ISA census of its stage loop)

A_GT = np.array([[9.08173020e-01, -2.58092254e-01, 2.21387350e-01, 4.98532315e+00],
                 [-2.85490902e-02, 5.66865806e-01, 7.60292965e-01, -2.13218259e+02],
                 [-2.53059848e-01, -7.49475117e-01, 4.48778146e-01, 5.56203489e+02],
                 [0.0, 0.0, 0.0, 1.0]])


def synth(n, seed=0):
    """anisotropic blob; fixed = A_gt . moving + unit jitter, columns permuted.
    ICP start = the aligned cloud disturbed by 0.05 rad, 2 % scale and a few units of shift."""
    rng = np.random.default_rng(seed)
    mv = rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0
    aligned = A_GT[:3, :3] @ mv + A_GT[:3, 3:4]
    fx = aligned + rng.normal(scale=1.0, size=(3, n))
    fx = np.ascontiguousarray(fx[:, rng.permutation(n)])
    th = 0.05
    R = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    c = aligned.mean(1, keepdims=True)
    start = 1.02 * R @ (aligned - c) + c + np.array([[3.0], [-2.0], [4.0]])
    return np.ascontiguousarray(mv), fx, np.ascontiguousarray(start)


def _cpu_per_pair(oracle, mv, fx, start, threads, sub, rows, nn_rows, real=None):
    """Per-point-pair costs (seconds) of the oracle's stages with `threads` threads on a bounded sample.
    real = (um [2, r, 360], uf [4, M, 360]): the chi-square leg runs on these descriptors of the REAL clouds (r rows against all M
    columns, eight matrices) instead of on the subsample's."""
    m = fx.shape[1]
    oracle.set_threads(threads)
    cm, cf = oracle.get_centroid(mv, False), oracle.get_centroid(fx, False)
    x0m, x0f = oracle.pca_axis(mv.T), oracle.pca_axis(fx.T)
    t = time.perf_counter()
    mdm = oracle.get_mean_distance(mv[:, :sub], False)
    t_md = (time.perf_counter() - t) / (sub * (sub - 1) / 2)
    t = time.perf_counter()
    cnt_m, tot_m = oracle.shape_context_counts(cm, mdm, mv[:, :sub], "moving", x0=x0m)
    cnt_f, tot_f = oracle.shape_context_counts(cf, mdm, fx[:, :sub], "fixed", x0=x0f)
    t_sc = (time.perf_counter() - t) / (6.0 * sub * sub)                       # s per (ordered pair, frame)
    um, uf = oracle.normalise_counts(cnt_m, tot_m), oracle.normalise_counts(cnt_f, tot_f)
    if real is not None:
        rum, ruf = real
        t = time.perf_counter()
        for a in range(2):
            for b in range(4):
                oracle.unary_distance_matrix(rum[a], ruf[b])
        t_chi = (time.perf_counter() - t) / (8.0 * rum.shape[1] * ruf.shape[1])    # s per (pair, matrix)
    else:
        rows = min(rows, sub)
        t = time.perf_counter()
        for a in range(2):
            for b in range(4):
                oracle.unary_distance_matrix(um[a][:rows], uf[b][:sub // 2])
        t_chi = (time.perf_counter() - t) / (8.0 * rows * (sub // 2))          # s per (pair, matrix)
    nn_rows = min(nn_rows, start.shape[1])
    t = time.perf_counter()
    oracle.nn_argmin(start[:, :nn_rows], fx)
    t_nn = (time.perf_counter() - t) / (nn_rows * m)                           # s per pair per ICP iteration
    oracle.set_threads(1)
    return t_md, t_sc, t_chi, t_nn


def cpu_baseline(mv, fx, start, icp_iters, real=None):
    """The oracle (C port of the reference's loops, oracle/pm_oracle.c) on the GPU box's host cores, on a bounded sample of
    the same workload (~10-30 s of CPU work), per-pair costs extrapolated to the full N x M problem.  `value` uses every
    core this process may run on (row loops under OpenMP: rows are independent, results do not depend on the thread
    count); `one_core` is the same port on a single core."""
    import oracle
    oracle.build()
    n, m = mv.shape[1], fx.shape[1]
    cores = oracle.host_threads()

    def extrapolate(t_md, t_sc, t_chi, t_nn):
        return (t_md * (n * (n - 1) / 2 + m * (m - 1) / 2) + t_sc * (2.0 * n * n + 4.0 * m * m)) / (n * m) + 8.0 * t_chi + icp_iters * t_nn

    sub1, rows1, nn1 = min(n, 3072), 128, 1024
    one = _cpu_per_pair(oracle, mv, fx, start, 1, sub1, rows1, nn1)
    scale = max(1, min(cores, 64))
    subT = min(n, 3072 * max(1, int(scale ** 0.5)))
    rowsT, nnT = 128 * scale, 1024 * scale
    allc = _cpu_per_pair(oracle, mv, fx, start, cores, subT, rowsT, nnT, real=real) if (cores > 1 or real is not None) else one
    chi_note = ("8 chi2 blocks of %d x %d = %.2f %% of the real N x M pairs per matrix, on the descriptors of the full clouds"
                % (real[0].shape[1], real[1].shape[1], 100.0 * real[0].shape[1] / n) if real is not None
                else "8 chi2 blocks of %d x %d" % (min(rowsT, subT), subT // 2))
    names = ("mean_distance", "shape_context_per_frame", "chi2_per_matrix", "icp_nn_per_iteration")
    return {"value": 1.0 / extrapolate(*allc), "unit": "point-pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle/pm_oracle.c, row loops on %d threads (OpenMP; os.cpu_count() = %s): mean distance + descriptors of a "
                      "%d-point subsample, %s, 1 NN pass of %d x %d; per-pair costs extrapolated to N=M=%d, "
                      "%d ICP iterations" % (cores, os.cpu_count(), subT, chi_note, min(nnT, n), m, n, icp_iters),
            "per_pair_ns": dict(zip(names, (x * 1e9 for x in allc))),
            "one_core": {"value": 1.0 / extrapolate(*one), "cores": 1, "per_pair_ns": dict(zip(names, (x * 1e9 for x in one))),
                         "sample": "%d-point subsample, chi2 blocks %d x %d, NN %d x %d" % (sub1, rows1, sub1 // 2, nn1, m)},
            # the literal reference (pure-Python loops), extrapolated from the per-pair costs the survey measured by running it
            # (3.6 us / 17 us / 100 us / 50 ns per pair for mean distance / descriptor frame / chi2 matrix / ICP)
            "reference_python_extrapolated_s": (3.6e-6 * (n * (n - 1) / 2 + m * (m - 1) / 2) + 17e-6 * (2.0 * n * n + 4.0 * m * m)
                                                + 100e-6 * 8.0 * n * m + 50e-9 * icp_iters * n * m)}


def pmc_traffic(kernel_substring, n):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_traffic.py: FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950) -> (GB, source) or
    (None, None) when no counters were collected for this configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, None
    if int(d.get("n", -1)) != int(n):
        return None, None
    # the counters describe the kernel sources they were collected on: a source edited since makes them stale
    import hashlib
    stale = []
    for fname, sha in (d.get("kernel_source_sha16") or {}).items():
        try:
            with open(os.path.join(ROOT, "platymatch_amd", "csrc", fname), "rb") as fh:
                if hashlib.sha256(fh.read()).hexdigest()[:16] != sha:
                    stale.append(fname)
        except OSError:
            stale.append(fname)
    for name, rec in d.get("kernels", {}).items():
        if kernel_substring in name:
            src = "profiles/pmc_traffic.json (round %s: %s)" % (d.get("round"), ", ".join(d.get("source", [])))
            if stale:
                src += "; STALE: %s changed since these passes were collected" % ", ".join(stale)
            return float(rec["traffic_gb"]), src
    return None, None


def shader_clock_during(launch, dev, expected_ms):
    """The shader clock WHILE `launch` runs (VERDICT r04 next #4b), measured on the product build: one extra wave on a side stream
    (pm_clock_probe, include/platymatch_hip.h) stores (s_memtime, s_memrealtime) once a millisecond for the length of the launch
    plus 0.4 s; clock between two samples = d s_memtime / d s_memrealtime x 100 MHz (MI355X_MICROARCH.md's in-kernel clock test).
    -> dict: median / min / max under load (the middle 80 % of the launch), the idle clock after it, the launch's own duration."""
    import torch
    from platymatch_amd import _native as nat
    n_s = int(min(4900, expected_ms + 400))
    side = torch.cuda.Stream(dev)
    samples = torch.zeros(2 * n_s, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    nat.check(nat.load().pm_clock_probe(samples.data_ptr(), n_s, 100_000, side.cuda_stream))
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = samples.cpu().numpy().reshape(-1, 2).astype(np.float64)
    d = np.diff(st, axis=0)
    ghz = d[:, 0] / np.maximum(d[:, 1], 1.0) * 0.1                    # ticks of the shader clock per 10 ns -> GHz
    load = ghz[int(0.1 * ms):max(int(0.9 * ms), int(0.1 * ms) + 1)]
    idle = ghz[int(ms) + 100:]
    return {"under_load_ghz": {"median": float(np.median(load)), "min": float(load.min()), "max": float(load.max()), "samples": int(load.size)},
            "after_the_launch_ghz": float(np.median(idle)) if idle.size else None, "launch_ms_with_probe": ms,
            "method": "pm_clock_probe: one extra wave samples (s_memtime, s_memrealtime) every millisecond beside the product kernel"}


def rank_devices(dist, group, dev, world, rank, ident=None):
    """Which physical device every rank drives (VERDICT r04 next #2a): name | uuid (or PCI ids), all-gathered as bytes."""
    import torch
    if ident is None:
        props = torch.cuda.get_device_properties(dev)
        # uuid AND PCI address: two ranks on one card agree in both, two cards differ in the address even where a
        # driver reports one uuid for all of them (a legitimate N-GPU run must never be refused as a rehearsal)
        ident = "%s | %s | pci %s:%s.%s" % (props.name, str(getattr(props, "uuid", "")), getattr(props, "pci_domain_id", "?"),
                                            getattr(props, "pci_bus_id", "?"), getattr(props, "pci_device_id", "?"))
    text = ident.encode()[:120]
    mine = torch.zeros(128, dtype=torch.uint8)
    mine[:len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8)
    on_host = dist.get_backend(group) == "gloo"
    mine = mine if on_host else mine.to(dev)
    everyone = torch.zeros(world * 128, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    names = [bytes(everyone[g * 128:(g + 1) * 128].cpu().tolist()).split(b"\0", 1)[0].decode() for g in range(world)]
    return {"backend": dist.get_backend(group), "world": world, "distinct_devices": len(set(names)), "devices": names}


def preflight(dist, group, dev, world, timeout_s=60.0):
    """Every collective the step and its extras use, once, on 1-KB tensors, under a watchdog (VERDICT r04 next #2c): a rendezvous
    or transport problem becomes a clear error within timeout_s instead of a hang inside the timed region."""
    import threading
    import torch
    on_host = dist.get_backend(group) == "gloo"
    d = torch.device("cpu") if on_host else dev
    done, err = threading.Event(), []

    def run():
        try:
            if not on_host:
                torch.cuda.set_device(dev)                # (a new thread starts on device 0: barrier and synchronize go by the current device)
            t = torch.ones(128, dtype=torch.float64, device=d)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)                      # mean-distance piece sums
            f = torch.zeros(1, dtype=torch.int32, device=d)
            dist.all_reduce(f, op=dist.ReduceOp.MAX, group=group)                      # symmetry flag, agree_max
            out = torch.empty(world * 128, dtype=torch.float64, device=d)
            dist.all_gather_into_tensor(out, t, group=group)                           # descriptor gather
            dist.broadcast(t, src=0, group=group)                                      # assignment queries
            whole = [torch.empty_like(t) for _ in range(world)] if dist.get_rank(group) == 0 else None
            dist.gather(t, whole, dst=0, group=group)                                  # assignment answers
            dist.barrier(group=group)
            if not on_host:
                torch.cuda.synchronize(dev)
            if float(out.sum()) != 128.0 * world * world:          # (ones, summed over the ranks, gathered from every rank)
                err.append("preflight all-gather returned %r, expected %r" % (float(out.sum()), 128.0 * world * world))
        except Exception as e:       # noqa: BLE001
            err.append("%s: %s" % (type(e).__name__, e))
        finally:
            done.set()

    th = threading.Thread(target=run, name="pm-bench-preflight", daemon=True)
    t0 = time.perf_counter()
    th.start()
    if not done.wait(timeout_s):
        sys.stderr.write("bench.py rank %d: collective preflight did not finish in %.0f s (backend %s, world %d): check MASTER_ADDR / "
                         "MASTER_PORT, HSA_ENABLE_IPC_MODE_LEGACY=0, one visible device per rank\n" % (dist.get_rank(group), timeout_s, dist.get_backend(group), world))
        sys.stderr.flush()
        os._exit(3)
    if err:
        raise SystemExit("bench.py rank %d: collective preflight failed: %s" % (dist.get_rank(group), err[0]))
    return time.perf_counter() - t0


def spawn_ranks(n_ranks, argv):
    """`python bench.py --gpus N` started as ONE process (no WORLD_SIZE in the environment): become the launcher.  N fresh
    children of this same script are started, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set — what `python -m torch.distributed.run --nproc-per-node N` would set.  The launcher itself never imports torch and
    never touches a GPU (a process that has initialised the GPU must not be replaced or forked on this pool); rank 0's JSON
    line reaches stdout because the children inherit it.  When a rank fails the others are given a grace period (they may be
    waiting in a collective for it) and are then ended by their own PIDs.  -> the worst child exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(base.get("MASTER_PORT") or port), WORLD_SIZE=str(n_ranks),
                LOCAL_WORLD_SIZE=str(n_ranks), PM_BENCH_SPAWNED="1")
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    procs = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    grace = float(os.environ.get("PM_BENCH_SPAWN_GRACE_S", "60"))
    worst, failed_at = 0, None
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is not None:
                live.remove(p)
                if rc != 0:
                    worst = worst or rc
                    failed_at = failed_at or time.monotonic()
        if failed_at is not None and live and time.monotonic() - failed_at > grace:
            for p in live:
                p.terminate()                    # exactly the children this launcher started
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            worst = worst or 1
            break
        time.sleep(0.05)
    return worst


def dry_run(args):
    """--dry-run: the launcher / rendezvous path without a GPU (CPU test of `bench.py --gpus N`): every rank joins a gloo
    group, the timing all-reduce (MAX) and the barrier of the real run are exercised, rank 0 prints a JSON line."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    ranks = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pre_s = preflight(dist, dist.group.WORLD, None, world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seen = float(t.item())
        ranks = rank_devices(dist, dist.group.WORLD, None, world, rank, ident="no device (dry run), rank %d" % rank)
        ranks["preflight_s"] = pre_s
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1.0
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "max_rank_plus_one": seen, "ranks": ranks,
                          "spawned": os.environ.get("PM_BENCH_SPAWNED") == "1", "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=50000, help="points per cloud (default: the BASELINE 50k configuration)")
    ap.add_argument("--icp-iters", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-config2", action="store_true", help="skip the fully timed host run of BASELINE config 2 (5 000 nuclei, ~1 min of CPU)")
    ap.add_argument("--no-assignment", action="store_true", help="skip the untimed extra leg (eight assignments of the same build)")
    ap.add_argument("--dry-run", action="store_true", help="launcher + rendezvous only (gloo, no GPU): what the CPU test-suite runs")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # One process asked for several GPUs (the driver's `python bench.py --gpus N`): start the N ranks as fresh children
    # BEFORE torch is imported or a GPU is touched, relay rank 0's line, return the worst child's exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start %d ranks (python bench.py --gpus %d does so by itself when WORLD_SIZE is unset)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist
    from platymatch_amd import _kernels as K
    from platymatch_amd import _native as nat
    from platymatch_amd import pipeline as P

    if os.environ.get("PM_BENCH_ONE_DEVICE") != "1" and local >= torch.cuda.device_count():
        raise SystemExit("rank %d: device %d not found (%d visible)" % (rank, local, torch.cuda.device_count()))
    nat.load()                                   # fails loudly if the HIP library is missing
    # rehearsal switches (not used by the driver): PM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # PM_BENCH_BACKEND=gloo replaces RCCL, so the sharded code path can be exercised on a one-GPU box
    if os.environ.get("PM_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("PM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
    ranks_record = None
    if world > 1:
        pre_s = preflight(dist, group, dev, world)
        ranks_record = rank_devices(dist, group, dev, world, rank)
        ranks_record["preflight_s"] = pre_s
        ranks_record["rehearsal_on_one_device"] = os.environ.get("PM_BENCH_ONE_DEVICE") == "1"
        if ranks_record["distinct_devices"] != world and not ranks_record["rehearsal_on_one_device"]:
            raise SystemExit("bench.py: %d ranks drive only %d distinct devices (%s): one GPU per rank is what --gpus N measures "
                             "(PM_BENCH_ONE_DEVICE=1 is the rehearsal switch)" % (world, ranks_record["distinct_devices"], ranks_record["devices"]))

    n = m = args.points
    mv_h, fx_h, start_h = synth(n)
    mov, fix, start = (nat.to_dev(x, dev=dev) for x in (mv_h, fx_h, start_h))
    be = P.GpuBackend(dev)
    from platymatch_amd.estimate_transform.shape_context import pca_view
    views = (pca_view(mv_h), pca_view(fx_h))
    bn, bm = P.shard_bounds(n, world), P.shard_bounds(m, world)
    r0, r1 = bn[rank], bn[rank + 1]
    U = torch.empty((8, r1 - r0, m), dtype=torch.float64, device=dev)     # this rank's cost rows, resident output
    icp_ws = nat.workspace(nat.load().pm_icp_workspace(n, m), dev)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]

    def step(marks=None):
        if marks: marks[0].record()
        # one GPU: the two clouds' statistics on two streams; sharded: the pieces of the pair sum interleaved over the ranks
        # (the PCA axis is sklearn's own NumPy calls on the host, run inside the step on the host arrays the clouds came from —
        # what the boundary hands over; the O(N^2) statistics read the resident tensors)
        (cm, mdm, x0m), (cf, mdf, x0f) = P.statistics_of_both(be, mov, fix, group, views)
        if marks: marks[1].record()
        sc_m = be.shape_context(mov, cm, mdm, x0m, 2, r0, r1 - r0)
        sc_f = be.shape_context(fix, cf, mdf, x0f, 4, bm[rank], bm[rank + 1] - bm[rank])
        # verify the frame-permutation relation on the local rows (a 4-byte read-back; sharded: flags max-reduced, then
        # only frame 1 is gathered) -- inside the timed step, because the choice of kernel depends on it
        if world == 1:
            symmetric[0] = K.chi2_symmetric(sc_m, sc_f)
        else:
            sc_f = P.gather_fixed_descriptors(be, sc_m, sc_f, bm, group)
            symmetric[0] = sc_f.shape[0] == 1
        if marks: marks[2].record()
        sc_m_last[0], sc_f_last[0] = sc_m, sc_f
        if symmetric[0]:
            K.chi2_cost8_frame1(sc_m[0], sc_f[0], out=U, info=table_info if marks is None else None)   # (report read back in warm-up only)
        else:
            K.chi2_cost8(sc_m, sc_f, out=U, path="general")
        if marks: marks[3].record()
        if world == 1 or n < P.ICP_SHARD_MIN_POINTS:
            # every rank refines on its own (replicas): one grid-search iteration over 50k points costs less than the
            # latency of the collective a sharded iteration would need (pipeline.ICP_SHARD_MIN_POINTS)
            work = start.clone()
            A, res, _ = K.icp(work, fix, args.icp_iters, ws=icp_ws)
        else:
            A, res = P.icp_sharded(be, start, fix, args.icp_iters, group)
        if marks: marks[4].record()
        return A, res

    symmetric = [False]                          # which chi-square kernel the last step's descriptors selected
    table_info = {}                              # which shells the half-cost kernel took from its term table
    sc_m_last, sc_f_last = [None], [None]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=group)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()                                      # barrier + synchronize on both sides of exactly `steps` steps
    t0 = time.perf_counter()
    for k in range(args.steps):
        A, res = step(ev[k])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    stage = np.array([[ev[k][i].elapsed_time(ev[k][i + 1]) for i in range(4)] for k in range(args.steps)]).mean(0)  # ms
    chi2_ms = float(stage[2])
    rows = r1 - r0
    algo_bytes = 8.0 * 8 * rows * m + 2880.0 * (2 * rows + 4 * m)          # 8 matrices written + descriptors read once
    achieved = algo_bytes / (chi2_ms * 1e-3) / 1e9
    # float64 VALU view of the same launch.  Half-cost kernel: per (pair, bin) 4 terms x (sub, 2 mul, add, rcp, 5 fma) + 8
    # running-sum adds = 48 instructions, 68 flop (fma = 2); general kernel: 8 terms x 11 instructions, 128 flop.
    # Issue model (measured, tools/microbench/fp64_issue.hip + SQ counters): 4 cycles per instruction, 16 for v_rcp_f64.
    sym = bool(symmetric[0])
    if sym and not table_info:                   # no warm-up step ran: one untimed launch for the report
        K.chi2_cost8_frame1(sc_m_last[0][0], sc_f_last[0][0], out=U, info=table_info)
    tabled = sum(table_info.get("tabled_shells", [])) if sym else 0
    ft = tabled / 30.0                           # fraction of the (pair, bin) terms served from the table
    kernel_name = ("pm::chi2_sym_kernel<4,2,-1,%d>" % table_info.get("table_size", 0)) if sym else "pm::chi2_kernel<2,4>"
    # a tabled (pair, bin): 4 address adds + 4 ds_read_b64 + 8 running-sum adds (8 flop, 12 VALU instructions)
    flops = ((68.0 * (1 - ft) + 8.0 * ft) if sym else 128.0) * 360 * rows * m
    instr = ((48.0 * (1 - ft) + 12.0 * ft) if sym else 88.0) * 360 * rows * m / 64.0          # wave64 VALU instructions
    issue_cycles = (((44 * 4 + 4 * 16) * (1 - ft) + 12 * 4 * ft) if sym else (80 * 4 + 8 * 16)) * 360.0 * rows * m / 64.0 / 1024.0   # per SIMD
    tflops = flops / (chi2_ms * 1e-3) / 1e12
    ns_per_instr = chi2_ms * 1e6 / (instr / 1024.0)                          # per SIMD (256 CUs x 4)
    issue_bound_ms = issue_cycles / 2.4e9 * 1e3                              # at the 2.4 GHz maximum clock

    add_floor_s = 8.0 * 360.0 * rows * m / 64.0 / 1024.0 * 4.0 / 2.4e9      # the running-sum adds alone (183 ms at 50k)
    # HBM traffic of that launch: from the committed rocprofv3 PMC passes of this configuration (1 GPU, this N), else null
    traffic, traffic_src = pmc_traffic("chi2_sym_kernel" if sym else "chi2_kernel<", n) if world == 1 else (None, None)
    traffic_why = None
    if traffic is None:
        traffic_why = ("no PMC pass of this configuration is committed: profiles/pmc_traffic.json holds the 1-GPU N = M = 50 000 launch"
                       + ("; a rank's row block reads the same descriptors and writes rows/%d of the matrices (the launch is linear in "
                          "its rows, profiles/r04_chi2_blocks.txt)" % world if world > 1 else ""))

    # per-shell maxima of the integer counts behind the descriptors (what decides which shells the 94 x 94 term table can serve)
    shell_max = None
    if sym:
        def shell_maxima(sc1):
            pos = sc1[sc1 > 0]
            tot = float(torch.round(1.0 / pos.min())) if pos.numel() else 1.0
            return torch.round(sc1 * tot).view(-1, 30, 12).amax(dim=(0, 2)).to(torch.int64).cpu().tolist()
        cm_, cf_ = shell_maxima(sc_m_last[0][0]), shell_maxima(sc_f_last[0][0])
        per_ring = [[int(min(min(cm_[6 * r:6 * r + 6]), min(cf_[6 * r:6 * r + 6]))), int(max(max(cm_[6 * r:6 * r + 6]), max(cf_[6 * r:6 * r + 6])))]
                    for r in range(5)]
        shell_max = {"moving": cm_, "fixed": cf_, "per_ring_min_max_of_the_shell_maxima": per_ring, "table_side": table_info.get("table_size"),
                     "note": "shell g = 6 x ring + theta sector; a shell is tabled when both clouds' largest count in it is below the table's side"}

    # the shader clock while the cost kernel runs (one GPU; an extra untimed launch with a one-wave probe beside it)
    clock = None
    if world == 1 and sym and not args.no_assignment:
        clock = shader_clock_during(lambda: K.chi2_cost8_frame1(sc_m_last[0][0], sc_f_last[0][0], out=U), dev, chi2_ms)

    # the other two stages: their compulsory HBM traffic is O(N) against O(N^2) work, so HBM is not what binds
    sc_bytes = (24.0 * n + 2880.0 * 2 * (r1 - r0)) + (24.0 * m + 2880.0 * 4 * (bm[rank + 1] - bm[rank]))
    sc_pairs = float(r1 - r0) * n + float(bm[rank + 1] - bm[rank]) * m
    sc_ms = float(stage[1])
    icp_bytes_iter = 24.0 * (n + m) + 4.0 * n
    icp_ms = float(stage[3])
    stage_roofline = {
        "shape_context": {
            "kernel": "pm::sc_tile_kernel<2>, <4> (+ sc_prepare / sc_finish)", "bound": "hbm", "achieved": sc_bytes / (sc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": sc_bytes / (sc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": sc_bytes,
            "binding_bound": "VALU issue: ~70 vector instructions per ordered pair (6 float64, the rest float32 / integer: the neighbour is "
                             "pre-classified in float32 and decided in float64 only when within float32's reach of a bin boundary, "
                             "DESIGN.md §4.5), 16 queries per workgroup share every neighbour load, one histogram per query (frames 2..4 are "
                             "its phi permutations); round 2's kernel: ~140 float64 instructions per pair, 17.4 ms",
            "ns_per_pair": sc_ms * 1e6 / sc_pairs,
            "valu_issue_estimate_ms": ((6 * 4.0 + 64 * 2.0) / 2.4e9) * sc_pairs / 64.0 / 1024.0 * 1e3},
        "icp": {
            "kernel": "pm::icp_iter_kernel (one launch per iteration: apply + residual + bounded search + moment tree + solve)", "bound": "hbm",
            "achieved": icp_bytes_iter * args.icp_iters / (icp_ms * 1e-3) / 1e9 if args.icp_iters else None, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": (icp_bytes_iter * args.icp_iters / (icp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if args.icp_iters else None,
            "algorithmic_bytes_per_iteration": icp_bytes_iter, "us_per_iteration": icp_ms * 1e3 / max(args.icp_iters, 1),
            "binding_bound": "latency: one dependent launch per iteration whose workgroups each walk ~7 dependent memory round "
                             "trips (previous match -> cell ranges -> candidates -> partial sums -> arrival counter); 2.6 MB of "
                             "compulsory traffic per iteration cannot load HBM (DESIGN.md §4)"},
    }

    # Extra leg, OUTSIDE the timed region and not part of `value`: the eight assignment problems of the same build, with the
    # 160 GB of cost matrices still resident in HBM (device-resident route only: bounded, a hypothesis that cannot be certified
    # is reported, never handed to the hours-long dense solver).  One GPU only (sharded runs hold row blocks, not matrices).
    assignment = None
    extra_hung = [False]
    try:
        if world == 1 and not args.no_assignment:
            from platymatch_amd import lsap as L
            torch.cuda.synchronize()
            t_as = time.perf_counter()
            a_info = {}
            lsa = L.solve_eight_on_device(U, info=a_info, allow_host=False)
            t_as = time.perf_counter() - t_as
            certified = [x is not None for x in lsa]
            assignment = {"seconds": t_as, "hypotheses_certified_unique": int(sum(certified)), "routes": a_info.get("routes"),
                          "pricing_rounds": [d.get("rounds") for d in a_info.get("details", [])[:4]],
                          "dijkstra_steps": [d.get("steps") for d in a_info.get("details", [])[:4]],
                          "note": "scipy.optimize.linear_sum_assignment's answer for the eight N x M matrices (_dock_widget.py:604-611) by a sparse "
                                  "core solved on the host and priced + certified against every entry on the device (DESIGN.md §4.3); not in `value`"}
    except Exception as e:      # noqa: BLE001 — an extra must not cost the headline line
        import traceback
        traceback.print_exc()
        assignment = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    if world > 1 and not args.no_assignment:
        # the eight assignments by the DEFAULT route of a sharded registration (cost_mode='auto': row blocks of the float32 filter on
        # their ranks, the root's queries answered by every rank, exact costs on the root; pipeline.assign_sharded_filtered) — or,
        # where the frames do not permute, by the exact sharded route on the row blocks of the timed build.  Untimed extra.
        # Run under a WATCHDOG: this protocol has never met more than one real device under RCCL before the driver's scaling run; a
        # collective that does not return must cost this extra, not the headline line — on a timeout every rank reports, rank 0 prints
        # its line, and the process leaves without the closing barrier (PM_BENCH_EXTRA_TIMEOUT_S, default 300).
        import threading
        box = {}

        def sharded_extra():
            try:
                torch.cuda.set_device(dev)
                fence()
                t0_ = time.perf_counter()
                info_ = {}
                if symmetric[0]:
                    lsa_ = P.assign_sharded_filtered(be, sc_m_last[0], sc_f_last[0], bn, group, info=info_)
                else:
                    lsa_ = P.assign(U, bn, group, info=info_)
                fence()
                t_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX, group=group)
                box.update(seconds=float(t_.item()), lsa=lsa_, info=info_, error=None)
            except Exception as e:      # noqa: BLE001 — an extra must not cost the headline line
                box.update(seconds=None, lsa=None, info={}, error="%s: %s" % (type(e).__name__, str(e)[:300]))

        th = threading.Thread(target=sharded_extra, name="pm-bench-sharded-extra", daemon=True)
        th.start()
        th.join(float(os.environ.get("PM_BENCH_EXTRA_TIMEOUT_S", "300")))
        if th.is_alive():
            extra_hung[0] = True
            box.update(seconds=None, lsa=None, info={}, error="timed out after %s s (watchdog): the line is printed without this extra and the "
                                                               "process exits without the closing barrier" % os.environ.get("PM_BENCH_EXTRA_TIMEOUT_S", "300"))
        lsa, a_info = box.get("lsa"), box.get("info", {})
        assignment = {"seconds": box.get("seconds"), "route": "sharded filter (cost_mode='auto')" if symmetric[0] else "sharded exact", "error": box.get("error"),
                      "routes": a_info.get("routes"), "mode": a_info.get("mode"),
                      "perfect_matchings": None if lsa is None else [bool(len(set(c.tolist())) == len(c)) for _, c in lsa],
                      "note": "the protocol of lsap_sharded.py timed at %d ranks: every query of the root's sparse-core solver is a broadcast + "
                              "gather round; max over ranks, four pairings one after the other; not in `value`" % world}

    # Second extra, OUTSIDE the timed region and never the headline: the relaxed-rounding cost build (what cost_mode='auto' starts
    # from between 1 024 and 8 192 nuclei; pm_chi2_cost8_relaxed: no bit identity, every entry within delta of the exact one, used
    # only behind a certificate against the exact matrix's listed entries).  Launch time by HIP events on the launching stream.
    relaxed_extra = filter_extra = None
    try:
        if world == 1 and symmetric[0] and not args.no_assignment:
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                K.chi2_cost8_relaxed(sc_m_last[0][0], sc_f_last[0][0], out=U, variant=P.RELAXED_VARIANT)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            # the eight assignments solved on the relaxed matrices and certified against the EXACT matrices on their matched and
            # near-tight entries (pm_chi2_entries_sym, lsap.certify_listed): bounded like the leg above, nothing is rebuilt here
            pairing_of = {p[0]: t for t, p in enumerate(K.PAIRINGS)}
            a1, b1 = sc_m_last[0][0], sc_f_last[0][0]

            def exact_entries(h):
                return lambda rows, cols: tuple(x.cpu().numpy() for x in K.chi2_entries(a1, b1, pairing_of[h], rows, cols))
            r_info = {}
            torch.cuda.synchronize()
            t_rs = time.perf_counter()
            lsa_r = L.solve_eight_on_device(U, info=r_info, allow_host=False, exact_entries=exact_entries, cost_delta=K.chi2_relaxed_delta(),
                                            exact_rebuild=lambda h: None)
            t_rs = time.perf_counter() - t_rs
            on_relaxed = [str(d.get("cost_mode", "")).startswith("relaxed") for d in r_info.get("details", [])]
            relaxed_extra = {"kernel": "pm::chi2_sym_kernel<4,3,-1,64,RELAX>", "launch_ms": min(ts), "exact_launch_ms": chi2_ms, "speedup": chi2_ms / min(ts),
                             "per_entry_error_bound": K.chi2_relaxed_delta(),
                             "assignment_seconds": t_rs, "hypotheses_certified_on_exact_entries": int(sum(on_relaxed)),
                             "equal_to_exact_matrices_assignments": [bool(ok and x is not None and y is not None and np.array_equal(x[1], y[1]))
                                                                     for ok, x, y in zip(on_relaxed, lsa_r, lsa)],
                             "note": "NOT in `value` (the headline stays the exact build).  What estimate_transform's default cost_mode='auto' starts from below 8 192 nuclei: U = 0.5 (sum a + sum b) - 2 sum ab/(a+b), "
                                     "v_rcp_f64 + one Newton step, four running sums per row (the twins coincide); an assignment solved on "
                                     "these matrices counts only once it is proven to be the exact matrix's unique optimum from the exact "
                                     "values of its matched and near-tight entries (a few N of them, evaluated by a small kernel), else that "
                                     "pairing is rebuilt exactly (profiles/r04_chi2_relaxed.txt)"}
            # third extra: the float32 FILTER build (cost_mode='filter') and the eight assignments solved through it — four approximate
            # matrices select entries, every cost is exact; nothing exact is built
            ts = []
            Uf = U.view(torch.float32).reshape(-1)[:4 * n * m].view(4, n, m)       # float32 storage (the product's), carved out of the resident buffer
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                K.chi2_filter4(a1, b1, out=Uf)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            f_info = {}
            t_fs = time.perf_counter()
            lsa_f = L.solve_four_filtered(Uf, lambda t: (lambda rows, cols: tuple(x.cpu().numpy() for x in K.chi2_entries(a1, b1, t, rows, cols))),
                                          lambda t: (lambda rows, cols: K.chi2_entries(a1, b1, t, rows, cols, trusted=True)),
                                          K.chi2_filter_delta() + 1e-13, lambda t: K.chi2_cost_pair(sc_m_last[0], sc_f_last[0], t, True),
                                          info=f_info, allow_host=False)
            t_fs = time.perf_counter() - t_fs
            through = [str(d.get("cost_mode", "")).startswith("filter") for d in f_info.get("details", [])]
            filter_extra = {"kernel": "pm::filter4_kernel<-1, float>", "launch_ms": min(ts), "exact_launch_ms": chi2_ms, "speedup": chi2_ms / min(ts),
                            "per_entry_error_bound": K.chi2_filter_delta(), "assignment_seconds": t_fs,
                            # its own roofline: packed-float32 VALU against the vector peak, counting the ALGORITHMIC
                            # 5 flop per (pair, bin, pairing) — add, multiply, reciprocal, fused multiply-add — whether a term was computed
                            # or read from the 94 x 94 float32 table; and the write rate of its 16 N M bytes
                            "roofline": {"bound": "fp32_packed_valu", "achieved": 5.0 * 4 * 360 * n * m / (min(ts) * 1e-3) / 1e12, "peak": FP32_VALU_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": 5.0 * 4 * 360 * n * m / (min(ts) * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                                         "hbm_achieved": 16.0 * n * m / (min(ts) * 1e-3) / 1e9, "hbm_unit": "GB/s",
                                         "hbm_frac": 16.0 * n * m / (min(ts) * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": 16.0 * n * m + 2880.0 * (n + m),
                                         # the launch against ITS OWN instruction stream (profiles/r05_filter_table.txt: per wave and stage of 192
                                         # terms 192 v_rcp_f32 at ~8 cycles + 373 other instructions at 4): what binds it is issue, not the peak above
                                         "issue_bound_ms": FILTER_CYCLES_PER_TERM * 4 * 360 * n * m / 64.0 / 1024.0 / 2.4e9 * 1e3,
                                         "frac_of_issue_bound": FILTER_CYCLES_PER_TERM * 4 * 360 * n * m / 64.0 / 1024.0 / 2.4e9 * 1e3 / min(ts)},
                            "hypotheses_settled_without_an_exact_matrix": int(sum(through)),
                            "equal_to_exact_matrices_assignments": [bool(x is not None and y is not None and np.array_equal(x[1], y[1]))
                                                                    for x, y in zip(lsa_f, lsa)],
                            "note": "NOT in `value` (the headline stays the exact build).  What estimate_transform's default cost_mode='auto' does at this size: four matrices in packed float32 arithmetic only select "
                                    "entries for the assignment solver, whose costs and certificate are evaluated exactly "
                                    "(lsap.FilteredMatrix; profiles/r04_e2e.txt)"}
            K.chi2_cost8_frame1(sc_m_last[0][0], sc_f_last[0][0], out=U)          # leave the exact matrices behind
    except Exception as e:      # noqa: BLE001 — an extra must not cost the headline line
        import traceback
        traceback.print_exc()
        filter_extra = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    if rank == 0:
        final = (A.reshape(4, 4).cpu().numpy())
        out = {
            "metric": "point-pairs/s (shape-context + chi2 cost + ICP)",
            "value": n * m / (ms_per_step * 1e-3),
            "unit": "point-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]+north_star 50k build: N=M=%d synthetic clouds, descriptors (2+4 frames), "
                                   "8 chi2 cost matrices to HBM, %d-iteration affine ICP" % (n, args.icp_iters),
                       "n": n, "m": m, "chi2_matrices": 8, "icp_iterations": args.icp_iters, "pairs_per_step": n * m,
                       "sharding": "rows/%d" % world},
            "stage_ms": {"statistics": float(stage[0]), "shape_context": float(stage[1]), "chi2_cost8": chi2_ms, "icp": float(stage[3])},
            # the dominant kernel is bound by float64 VALU issue, not by HBM: the headline `frac` is the fraction of
            # the vector-float64 peak; the HBM view of the same launch stands beside it (hbm_*), with the most any kernel that adds
            # the reference's 360 terms per (pair, matrix) one after the other in float64 could reach (attainable_hbm_frac_upper_bound)
            "roofline": {"kernel": kernel_name, "bound": "fp64_valu", "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                         "hbm_achieved": achieved, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": achieved / HBM_PEAK_GBS,
                         "attainable_hbm_frac_upper_bound": (algo_bytes / (HBM_PEAK_GBS * 1e9)) / add_floor_s,
                         "attainable_note": "the eight running sums of a pair take 8 x 360 dependent float64 adds: %.0f ms per launch at 4 "
                                            "cycles per wave-instruction on 1 024 SIMDs at 2.4 GHz, against %.1f ms for the bytes at 8 TB/s "
                                            "(every term free)" % (add_floor_s * 1e3, algo_bytes / (HBM_PEAK_GBS * 1e9) * 1e3),
                         "traffic": traffic, "traffic_unit": "GB per launch", "traffic_source": traffic_src, "traffic_why_null": traffic_why,
                         "algorithmic_bytes": algo_bytes,
                         "tabled_shells": tabled if sym else None, "shell_count_maxima": shell_max,
                         "rows_of_this_rank": rows,
                         "note": "hbm_achieved = compulsory bytes / measured launch time.  The >=70 %-of-HBM target of north_star is NOT reachable "
                                 "with bit-identical float64 costs: 360 correctly rounded divisions per pair and matrix need ~70x "
                                 "more float64-VALU time than the 20 ms the bytes need.  Round 2 takes the terms of the sparsely "
                                 "filled shells (tabled_shells of 30) from a count-indexed table in LDS instead of dividing; the "
                                 "remaining shells run at the instruction-issue bound (fp64_valu)"},
            "stage_roofline": stage_roofline,
            "fp64_valu": {"achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                          "ns_per_wave_instruction_per_simd": ns_per_instr,
                          "issue_bound_ms": issue_bound_ms, "frac_of_issue_bound": issue_bound_ms / chi2_ms,
                          "shader_clock": clock,
                          "issue_bound_ms_at_measured_clock": (issue_cycles / (clock["under_load_ghz"]["median"] * 1e9) * 1e3) if clock else None,
                          "frac_of_issue_bound_at_measured_clock": (issue_cycles / (clock["under_load_ghz"]["median"] * 1e9) * 1e3 / chi2_ms) if clock else None,
                          "algorithmic_flop_view": {"flop_per_pair_and_matrix": 1800, "tflops": 1800.0 * 8 * rows * m / (chi2_ms * 1e-3) / 1e12,
                                                    "frac": 1800.0 * 8 * rows * m / (chi2_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS},
                          "note": "`achieved` counts EXECUTED flops of the launch (68 per divided (pair, bin) for all eight matrices, the Newton "
                                  "fused multiply-adds at 2 flop each; 8 per tabled one), which is ~10 % more generous than SURVEY.md §8(d)'s "
                                  "ALGORITHMIC count of 1 800 flop per pair and matrix (sub, add, mul, div, add per bin): algorithmic_flop_view "
                                  "gives that figure.  Measured issue cost of one wave64 float64 instruction on this chip: ~2.0 ns "
                                  "(fma/mul/add), ~6.9 ns (rcp); tools/microbench/fp64_issue.hip"},
            "icp_residual_first_last": [float(res[0]), float(res[-1])] if args.icp_iters else None,
            "icp_affine_finite": bool(np.isfinite(final).all()),
            "ranks": ranks_record,
            "assignment_extra": assignment,
            "relaxed_cost_build_extra": relaxed_extra,
            "filter_cost_build_extra": filter_extra,
        }
        if extra_hung[0]:
            out["cpu_baseline"] = {"skipped": "the sharded extra's watchdog fired: nothing further touches the device in this process"}
        elif not args.no_cpu_baseline:         # (rank 0, whatever the world: a SCALE line carries its baseline too, VERDICT r04 next #2b)
            try:
                # the chi-square leg of the CPU baseline runs on the real clouds' descriptors (the GPU's, verified equal to the oracle's
                # by the parity tests): >= 1 % of the N x M pairs of every matrix
                r_rows = max(1, min(n, (n + 99) // 100 + 12))
                real = (sc_m_last[0][:, :r_rows].cpu().numpy(), (sc_f_last[0] if sc_f_last[0].shape[0] == 4 else P.expand_frames(sc_f_last[0][0], 4)).cpu().numpy())
                out["cpu_baseline"] = cpu_baseline(mv_h, fx_h, start_h, args.icp_iters, real=real)
                if not args.no_cpu_config2:
                    # BASELINE config 2 complete on the host, TIMED (tools/cpu_config2.py), beside the same pair through the product
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import cpu_config2
                    c2 = cpu_config2.run(5000)
                    from conftest import synth_pair
                    from platymatch_amd.estimate_transform import perform_icp as pi
                    pi.VERBOSE = False
                    mv2, fx2, _ = synth_pair(5000, 42)
                    gpu_s = []
                    for _ in range(4):
                        torch.cuda.synchronize()
                        t_g = time.perf_counter()
                        P.estimate_transform(mv2, fx2, ransac_trials=8000, ransac_error=16, icp_iterations=50, seed=0)
                        torch.cuda.synchronize()
                        gpu_s.append(time.perf_counter() - t_g)
                    out["cpu_baseline"]["config2_timed_s"] = c2["seconds"]
                    out["cpu_baseline"]["config2"] = dict(c2, gpu_same_pair_s={"first_call": gpu_s[0], "median_of_next_three": float(np.median(gpu_s[1:]))},
                                                          note="a complete 5 000-nucleus registration (seeded: the reference's RANSAC index sets), host oracle against "
                                                               "the product on this GPU; host arrays in, host arrays out")
            except Exception as e:      # noqa: BLE001 — the baseline leg must not cost the headline line (whatever it got so far stays)
                import traceback
                traceback.print_exc()
                out["cpu_baseline"] = dict(out.get("cpu_baseline") or {}, error="%s: %s" % (type(e).__name__, str(e)[:300]))
        print(json.dumps(out), flush=True)
    if extra_hung[0]:
        sys.stdout.flush()
        os._exit(0)                      # a collective of the extra never returned: no barrier can be trusted any more
    if world > 1:
        # The closing barrier, under a watchdog as well: the timed region is over and rank 0's line is out, so a peer that has left
        # already (its own watchdog fired a moment before a collective of this rank's extra failed with "connection closed": the
        # ranks then disagree about the extra, seen on the GPU box) must cost neither a hang nor the exit code.
        import threading
        closed = threading.Event()

        def close():
            try:
                dist.barrier(group=group)
                dist.destroy_process_group()
            except Exception as e:       # noqa: BLE001
                sys.stderr.write("bench.py rank %d: closing barrier: %s: %s (the line above stands)\n" % (rank, type(e).__name__, str(e)[:200]))
            finally:
                closed.set()

        th = threading.Thread(target=close, name="pm-bench-close", daemon=True)
        th.start()
        if not closed.wait(float(os.environ.get("PM_BENCH_CLOSE_TIMEOUT_S", "300"))):
            sys.stderr.write("bench.py rank %d: closing barrier did not return; leaving without it\n" % rank)
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
