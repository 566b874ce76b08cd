"""Several independent registrations (BASELINE config 5: "replicas only" — pairs never exchange data): worker threads with a HIP
stream each on one GPU, pairs dealt to ranks largest first on several."""
import contextlib
import dataclasses
import threading

import numpy as np

from . import _native as nat
from .device_memory import idle_bytes, through_torch


def batch_costs(sizes):
    """Relative cost of a registration of N x M nuclei, for sharing a batch out: the eight N x M cost matrices grow
    with N*M, the eight Hungarian solves (the dominant host step) roughly with N*M*sqrt(min(N, M)) (measured at 2k-20k,
    profiles/r02_batch64.json)."""
    return [float(n) * float(m) * float(min(n, m)) ** 0.5 for n, m in sizes]


def batch_assignment(sizes, world):
    """Pairs -> ranks, largest first onto the least loaded rank (LPT): a pure function of the sizes, so every rank
    computes the same table without talking.  -> list of rank indices, one per pair."""
    cost = batch_costs(sizes)
    load = [0.0] * world
    owner = [0] * len(cost)
    for k in sorted(range(len(cost)), key=lambda k: (-cost[k], k)):
        g = min(range(world), key=lambda g: (load[g], g))
        owner[k] = g
        load[g] += cost[k]
    return owner


def _pair_size(pair):
    return tuple(int(x.shape[1]) for x in pair[:2])


def _run_local(pairs, ks, workers, seeds, kwargs, timings=None, reports=None):
    """This process's share of a batch: pairs ks on `workers` host threads, one HIP stream each."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from .pipeline import Options, cost_bytes, estimate_transform
    kwargs = dict(kwargs)
    given = kwargs.pop("options", None)
    opts = dataclasses.replace(Options.of(given), private_rng=True)
    stated = set(given) if isinstance(given, dict) else {f.name for f in dataclasses.fields(Options) if getattr(opts, f.name) != f.default}
    be = opts.backend
    on_gpu = be is None or getattr(be, "device", None) is None or torch.device(be.device).type == "cuda"
    if on_gpu:
        dev = nat.device(None if be is None else be.device)
        nat.load()

    # HBM gate: the eight cost matrices of a pair (64 N M bytes, plus descriptors) live on the device while it is being
    # assigned; workers wait until the pairs in flight leave room for theirs (a pair larger than the whole budget runs alone)
    gate = threading.Condition()
    in_flight = [0.0]
    budget = 0.0
    if on_gpu:
        free_b = torch.cuda.mem_get_info(dev)[0] + max(int(torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)), 0) + idle_bytes(dev)
        budget = 0.8 * free_b                # (what torch's allocator holds unused is there to be drawn on, as in GpuBackend.free_bytes)

    def need(k):
        n, m = _pair_size(pairs[k])
        return cost_bytes(n, n, m) + 2880.0 * (2 * n + 4 * m) * 2

    slots = threading.local()
    next_slot = [0]

    def worker_slot():
        if not hasattr(slots, "id"):
            with gate:
                slots.id = next_slot[0]
                next_slot[0] += 1
        return slots.id

    def one(k):
        det = {"timing": True} if timings is not None else ({} if reports is not None else None)
        if not on_gpu:                       # a caller-supplied host backend (tests): no stream to set
            out = estimate_transform(pairs[k][0], pairs[k][1], seed=seeds[k], details=det, options=opts, **kwargs)
        else:
            want = min(need(k), budget)
            with gate:
                while in_flight[0] + want > budget and in_flight[0] > 0:
                    gate.wait()
                in_flight[0] += want
            try:
                from . import lsap
                lsap.set_pin_base(None if (workers > 1 and len(ks) > 1) else 0)     # several registrations side by side: placement left to the scheduler (measured)
                stream = nat.side_stream(dev, ("batch worker", worker_slot()))   # persistent per worker thread
                side_by_side = through_torch() if (workers > 1 and len(ks) > 1) else contextlib.nullcontext()
                with torch.cuda.device(dev), torch.cuda.stream(stream), side_by_side:
                    out = estimate_transform(pairs[k][0], pairs[k][1], seed=seeds[k], details=det, options=opts, **kwargs)
                    stream.synchronize()
            finally:
                with gate:
                    in_flight[0] -= want
                    gate.notify_all()
        if timings is not None:
            timings[k] = det["timing"]
        if reports is not None:
            reports[k] = {"routes": det.get("assignment", {}).get("routes"), "mode": det.get("assignment", {}).get("mode"),
                          "cost_modes": [d.get("cost_mode") for d in det.get("assignment", {}).get("details", [])]}
        return out

    if on_gpu and workers > 1 and len(ks) > 1 and "icp_one_launch" not in stated:
        opts = dataclasses.replace(opts, icp_one_launch=False)       # several streams in flight: no persistent grid (perform_icp.ONE_LAUNCH)
    if on_gpu and workers > 1 and len(ks) > 1 and "keep_cost_buffer" not in stated:
        opts = dataclasses.replace(opts, keep_cost_buffer=False)     # (a buffer kept per worker stream would pin memory the HBM gate counts as free)
    # largest first: the long Hungarian solves start early and the short pairs fill the gaps at the end
    cost = batch_costs([_pair_size(pairs[k]) for k in ks])
    order = [ks[i] for i in sorted(range(len(ks)), key=lambda i: (-cost[i], ks[i]))]
    if workers <= 1 or len(order) <= 1:
        return {k: one(k) for k in order}
    with ThreadPoolExecutor(max_workers=min(workers, len(order))) as ex:
        return dict(zip(order, ex.map(one, order)))


def estimate_transform_batch(pairs, workers=8, seeds=None, group=None, timings=None, reports=None, **kwargs):
    """Several independent registrations (BASELINE config 5: "replicas only" — pairs never exchange data).

    One GPU (group=None): each worker thread drives its pairs on its own HIP stream — a PERSISTENT one (nat.side_stream:
    torch's allocator caches per stream; with a fresh stream per pair every cost buffer was a new hipMalloc, ~13 s of a 15 s
    batch) —, so the GPU stages of different pairs overlap and the host stages (assignment cores, RANSAC draws: GIL-free)
    run concurrently.  workers: 8 measured best on the 16 cores a one-GPU box grants (64 pairs of 2k-20k nuclei: 5 workers
    10.0 s, 8 5.8-6.9 s, 10 5.7-5.9 s, 12 6.6 s); the HBM gate below bounds what is in flight.
    Several GPUs (group = a torch.distributed group, one process per GPU): every rank holds the whole list; pairs are
    dealt to ranks largest first (batch_assignment, no communication), each rank registers its share as above, and ONE
    all-reduce of 40 doubles per pair (A_sc, A_icp, inlier counts; every entry is non-zero on its owner only, so the sum is
    exact) hands every result to every rank.  A failure on any rank is raised on all of them.

    Seeded pairs draw their RANSAC index sets from a private RandomState(seed) (the sets np.random.seed(seed) would
    give); unseeded pairs draw from NumPy's global generator one after the other.
    pairs: iterable of (moving, fixed); seeds: optional per-pair RANSAC seeds; timings: optional dict, filled with
    {pair index: wall-clock split of its stages} for the pairs this process registered (adds stream synchronisations);
    reports: optional dict, filled with {pair index: {"routes": how each of its eight assignments was obtained, "cost_modes": with
    cost_mode='relaxed', whether each was certified on the relaxed build or after an exact rebuild}}.
    -> list of (A_sc, A_icp, inliers) in input order, each identical to a stand-alone estimate_transform call."""
    import torch
    from .pipeline import Options, _dist, _world
    pairs = list(pairs)
    seeds = list(seeds) if seeds is not None else [None] * len(pairs)
    if len(seeds) != len(pairs):
        raise ValueError("one seed per pair")
    if "details" in kwargs:
        raise ValueError("details is per registration: call estimate_transform for the pair of interest")
    rank, world = _world(group)
    if world == 1:
        res = _run_local(pairs, list(range(len(pairs))), workers, seeds, kwargs, timings, reports)
        return [res[k] for k in range(len(pairs))]
    dist = _dist()
    owner = batch_assignment([_pair_size(p) for p in pairs], world)
    mine = [k for k in range(len(pairs)) if owner[k] == rank]
    failure = None
    try:
        res = _run_local(pairs, mine, workers, seeds, kwargs, timings, reports)
    except Exception as e:                     # keep the collective below matched on every rank, then raise everywhere
        failure, res = e, {}
    be = Options.of(kwargs.get("options")).backend
    on_host = dist.get_backend(group) == "gloo"
    dev = torch.device("cpu") if on_host else (nat.device(None if be is None else be.device))
    table = torch.zeros((len(pairs) + 1, 40), dtype=torch.float64, device=dev)
    table[len(pairs), 0] = 0.0 if failure is None else 1.0
    for k, (A_sc, A_icp, inl) in res.items():
        table[k, :16] = torch.as_tensor(np.asarray(A_sc.cpu() if nat.is_torch(A_sc) else A_sc, dtype=np.float64).reshape(16))
        table[k, 16:32] = torch.as_tensor(np.asarray(A_icp.cpu() if nat.is_torch(A_icp) else A_icp, dtype=np.float64).reshape(16))
        table[k, 32:] = torch.as_tensor(np.asarray(inl, dtype=np.float64))
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    if failure is not None:
        raise failure
    if float(table[len(pairs), 0]) != 0.0:
        raise RuntimeError("estimate_transform_batch: a registration failed on another rank")
    out = []
    host = table.cpu().numpy()
    for k, p in enumerate(pairs):
        A_sc, A_icp, inl = host[k, :16].reshape(4, 4).copy(), host[k, 16:32].reshape(4, 4).copy(), host[k, 32:].astype(np.int64)
        if nat.is_torch(p[0]):
            A_sc, A_icp = torch.as_tensor(A_sc, device=p[0].device), torch.as_tensor(A_icp, device=p[0].device)
        out.append((A_sc, A_icp, inl))
    return out
