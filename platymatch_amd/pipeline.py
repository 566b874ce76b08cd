"""Headless estimate_transform: the stage order of the napari widget's worker
(EstimateTransform._click_run, platymatch/_dock_widget.py:526-718) as a function, on one GPU or
row-sharded over the GPUs of a node (one process per GPU, torch.distributed / RCCL).

Sharding: rank g owns a contiguous block of moving rows and of fixed rows.
  descriptors   each rank builds its rows; the fixed descriptor sets are all-gathered (the one
                exchange the cost build needs — every moving row is compared with every fixed row);
  cost matrices each rank fills its [8, rows_g, M] block — no communication;
  assembly      for the host Hungarian solves, hypothesis h's full matrix is collected on rank
                h mod G (one collective per hypothesis), the ranks solve their hypotheses in
                parallel and the index vectors are all-gathered;
  ICP           moving rows sharded, fixed replicated; per iteration ONE all-gather of 26 doubles (24
                moment sums + the previous iteration's residual parts), reduced identically on every
                rank, so every rank solves the identical 4x4.
Cloud statistics (centroid, mean distance, PCA axis) are O(N)…O(N^2) on 24·N bytes and are computed
redundantly by every rank: identical inputs and a fixed reduction order give identical values
without a collective.
"""
import dataclasses
import os
import threading

import numpy as np

from . import _native as nat
from . import lsap
from .lsap import linear_sum_assignment, solve_many

from ._pairings import HYPOTHESES, PAIRINGS  # noqa: E402
_RNG_LOCK = threading.RLock()

# (the kept cost buffers and their leases: cost_buffers.py)
COST_CACHE_MIN_BYTES = 8 << 30     # cost buffers of at least this size are kept per (device, stream) between registrations (cost_buffers.py)
from .device_memory import big_empty  # noqa: E402
from .cost_buffers import _CostLease, _EarlyLease, cost_buffer, kept_cost_bytes, release_cost_buffers  # noqa: E402,F401


def kept_bytes_wanted(mode, n, m):
    """Bytes a one-GPU registration of N x M nuclei writes into the kept cost buffer when everything is resident: the four float32
    filter matrices (16 N M), or eight float64 matrices (64 N M)."""
    return filter_bytes(n, m)[0] if mode == 'filter' else 64.0 * n * m


def reserve(n, m=None, cost_mode='auto', device=None):
    """Make the kept cost buffer of this (device, current stream) large enough for registrations of n x m nuclei AHEAD of the first
    call — for callers who know their sizes (a widget that has loaded its detections, a batch driver): the first
    estimate_transform then starts from a warm buffer (50 000 nuclei: ~0.9 s of allocation for the default mode, ~3.5 s for
    cost_mode='exact', paid here instead).  Nothing is allocated below COST_CACHE_MIN_BYTES (8 GiB; such buffers go through
    torch's allocator).  -> bytes now kept.  release_cost_buffers() gives the memory back."""
    m = n if m is None else m
    dev = nat.device(device)
    mode = resolve_cost_mode(cost_mode, n, m, 1, GpuBackend)
    want = kept_bytes_wanted(mode, n, m)
    if want >= COST_CACHE_MIN_BYTES:
        import torch
        with torch.cuda.device(dev):
            lease = cost_buffer(dev, ((int(want) + 7) // 8,))
            if lease is not None:
                lease.release()
    lsap.warm_up(dev)                             # (and the first-use costs of the assignment stage's selection code, ~0.3-0.7 s once)
    return kept_cost_bytes(dev)


class EdgeGuardWarning(UserWarning):
    """estimate_transform met neighbours that sit on a bin boundary of the shape context within the reference's own rounding
    noise: the integer histograms are then not defined by the reference's source alone."""


from .backend import GpuBackend, RELAXED_VARIANT  # noqa: E402,F401


def cost_bytes(rows, n, m, world=1):
    """Device memory the assignment stage of an N x M registration holds at its peak on one rank: the eight cost matrices of
    its `rows` moving rows (64 rows M bytes) and, when N > M, the transposed copies the solver works on — the short side
    must be the rows (lsap.solve_pair_on_device): one 8 N M-byte copy per pairing in flight, four pairings side by side on
    one GPU (96 N M in all); sharded, a hypothesis with N > M is assembled whole on its owner next to its transpose
    (pipeline.assign: at most ceil(8 / G) hypotheses per owner, one transposed at a time)."""
    need = 64.0 * rows * m
    if n > m:
        need += 32.0 * n * m if world == 1 else 8.0 * n * m * (-(-8 // world) + 1)
    return need


def agree_max(value, group, device=None):
    """max over the ranks of a small non-negative integer (a 4-byte all-reduce): how ranks settle a decision each of them
    could take differently (free memory, local row counts), so that all of them enter the same sequence of collectives."""
    _, world = _world(group)
    if world == 1:
        return int(value)
    import torch
    dist = _dist()
    on_host = dist.get_backend(group) == "gloo"
    t = torch.tensor([int(value)], dtype=torch.int32, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def shard_bounds(n, world):
    """Contiguous row blocks, sizes differing by at most one: block g = [b[g], b[g+1])."""
    base, extra = divmod(n, world)
    b = [0]
    for g in range(world):
        b.append(b[-1] + base + (1 if g < extra else 0))
    return b


def _dist():
    import torch.distributed as dist
    return dist


def _world(group):
    if group is None:
        return 0, 1
    dist = _dist()
    return dist.get_rank(group), dist.get_world_size(group)


def _global_rank(group, group_rank):
    dist = _dist()
    if group is None or group is dist.group.WORLD:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def all_gather_rows(local, bounds, dim, group):
    """All-gather blocks of unequal row counts along `dim` (0 or 1) -> the complete tensor on every rank.  One
    all_gather_into_tensor per leading slice, each rank's block landing where it belongs in ONE output allocation: with equal
    blocks (rows divisible by the ranks) that allocation IS the result — the 144 MB of frame-1 descriptors at 50k are written
    once, by RCCL —; unequal blocks are padded to the largest and squeezed by one copy afterwards."""
    import torch
    dist = _dist()
    world = len(bounds) - 1
    if world == 1:
        return local
    if dim not in (0, 1) or (dim == 1 and local.dim() < 2):
        raise ValueError("all_gather_rows gathers along dimension 0 or 1")
    sizes = [bounds[g + 1] - bounds[g] for g in range(world)]
    biggest = max(sizes)
    lead = local.shape[0] if dim == 1 else 1                  # leading slices gathered one by one (frames of a descriptor set)
    tail = tuple(local.shape[dim + 1:])
    src = local if dim == 1 else local.unsqueeze(0)           # [lead, rows_g, *tail]
    if sizes[dist.get_rank(group)] != biggest:
        padded = torch.zeros((lead, biggest) + tail, dtype=local.dtype, device=local.device)
        padded[:, :src.shape[1]].copy_(src)
        src = padded
    src = src.contiguous()
    out = torch.empty((lead, world * biggest) + tail, dtype=local.dtype, device=local.device)
    for f in range(lead):
        dist.all_gather_into_tensor(out[f], src[f], group=group)
    if any(sz != biggest for sz in sizes):
        out = torch.cat([out[:, g * biggest:g * biggest + sizes[g]] for g in range(world)], dim=1)
    return out if dim == 1 else out[0]


def cloud_statistics(be, xyz, group=None, view=None):
    """(centroid [3], mean pairwise distance [1], first PCA axis [3]) of one cloud, identical on every rank.
    The O(N) parts are recomputed by every rank (same input, same reduction order, same bits).  The O(N^2) mean distance
    is shared out: rank g adds up the 256 x 256 tiles of the tile rows g, g + G, ..., the tile sums (every one non-zero
    on one rank only, so the element-wise all-reduce is exact) are combined, and every rank adds them in the fixed
    order of the one-device kernel — the same bits as on one GPU, at 1/G of the pair work."""
    rank, world = _world(group)
    kw = {"view": view} if (view is not None and getattr(be, "device_sampler", False)) else {}      # (the GPU backend; test doubles take none)
    if world == 1:
        return be.stats(xyz, **kw)
    dist = _dist()
    c, x0 = be.centroid_and_axis(xyz, **kw)
    part = be.mean_distance_partials(xyz, rank, world)
    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return c, be.mean_distance_finish(part, xyz.shape[1]), x0


def gather_fixed_descriptors(be, sc_m_loc, sc_f_loc, bounds, group=None):
    """Complete fixed-cloud descriptors on every rank.  Frames 2..4 of get_unary are, for all but edge-case rows,
    phi-sector permutations of frame 1: every rank verifies that bit for bit on its own rows of both
    clouds, the flags are max-reduced (4 bytes), and if it holds everywhere only frame 1 travels — [1, M, 360], a
    quarter of the bytes over xGMI; chi2_cost8 derives the rest.  Otherwise all four frames are gathered [4, M, 360]."""
    _, world = _world(group)
    if world == 1:
        return sc_f_loc
    dist = _dist()
    flag = be.symmetry_flag(sc_m_loc, sc_f_loc)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()) == 0:
        return all_gather_rows(sc_f_loc[:1].contiguous(), bounds, 1, group)
    return all_gather_rows(sc_f_loc, bounds, 1, group)


STATS_ON_TWO_STREAMS = True
RELAXED_CERTIFY_ON_EXACT_ENTRIES = True     # cost_mode='relaxed': certify on the exact matrix's listed entries (lsap.certify_listed); False: a 2 N delta margin on the relaxed one
FILTER_STORAGE_F32 = os.environ.get("PM_FILTER_F64") != "1"      # the filter matrices as float32 (half the memory and dense-pass traffic; same bound); PM_FILTER_F64=1: float64
FILTER_MIN_POINTS = 8192       # cost_mode='filter' below this: the relaxed mode (the filter's extra round trips cost more than its build saves: measured 36 / 30 ms at 5k, 72 / 78 at 10k, 178 / 212 at 20k, 0.99 / 1.24 s at 50k)
RELAXED_MIN_POINTS = 1024      # cost_mode='relaxed' below this: exact (the dense host solver takes such matrices, no certificate to lean on)


def statistics_of_both(be, mov, fix, group=None, views=(None, None)):
    """cloud_statistics of the moving and of the fixed cloud -> ((centroid, mean distance, axis), (...)).  On one GPU the two
    clouds' statistics — independent of each other — run on two streams: each cloud's serial pieces (the chain over the mean
    distance's piece sums, ~0.8 ms at 50 000 points; the one-workgroup centroid and axis kernels) then run beside the other
    cloud's wide launch instead of after it.  Sharded runs keep one stream (their collectives are ordered on it)."""
    _, world = _world(group)
    if world == 1 and getattr(mov, "is_cuda", False) and getattr(be, "device_sampler", False) and STATS_ON_TWO_STREAMS:
        import torch
        main = torch.cuda.current_stream(mov.device)
        side = nat.side_stream(mov.device, ("statistics", main.cuda_stream))
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sf = be.stats(fix, views[1])
        sm = be.stats(mov, views[0])
        main.wait_stream(side)
        for t in sf:
            t.record_stream(main)
        return sm, sf
    return cloud_statistics(be, mov, group, views[0]), cloud_statistics(be, fix, group, views[1])


def build_descriptors(be, mov, fix, group=None, guards=None, views=(None, None)):
    """Stages 526-545 of the widget: statistics and get_unary for both clouds.
    -> (sc_m [2, rows_g, 360], sc_f [4, M, 360] complete (or [1, M, 360], see gather_fixed_descriptors), moving row bounds).
    guards (optional list): receives the edge-guard counters of the two launches (this rank's rows), moving first."""
    rank, world = _world(group)
    n, m = mov.shape[1], fix.shape[1]
    if world > min(n, m):        # every rank sees the same clouds: all raise, before the first collective
        raise ValueError("cannot shard %d x %d points over %d ranks: every rank needs at least one row of each cloud" % (n, m, world))
    (cm, mdm, x0m), (cf, mdf, x0f) = statistics_of_both(be, mov, fix, group, views)
    bn, bm = shard_bounds(n, world), shard_bounds(m, world)
    if guards is not None and getattr(be, "device_sampler", False):       # (the GPU backend; test doubles have no guard)
        sc_m = be.shape_context(mov, cm, mdm, x0m, 2, bn[rank], bn[rank + 1] - bn[rank], guards=guards)
        sc_f_loc = be.shape_context(fix, cf, mdf, x0f, 4, bm[rank], bm[rank + 1] - bm[rank], guards=guards)
    else:
        sc_m = be.shape_context(mov, cm, mdm, x0m, 2, bn[rank], bn[rank + 1] - bn[rank])
        sc_f_loc = be.shape_context(fix, cf, mdf, x0f, 4, bm[rank], bm[rank + 1] - bm[rank])
    sc_f = gather_fixed_descriptors(be, sc_m, sc_f_loc, bm, group)
    return sc_m, sc_f, bn


def build_costs(be, mov, fix, group=None):
    """Descriptors + the eight chi-square matrices for this rank's moving rows -> (U [8, rows_g, M], bounds)."""
    sc_m, sc_f, bn = build_descriptors(be, mov, fix, group)
    return be.chi2_cost8(sc_m, sc_f), bn


_PHI = None


def expand_frames(sc1, n_frames):
    """Frames 1..n_frames of get_unary from frame 1 alone [rows, 360] -> [n_frames, rows, 360]: frames 2..4 permute the twelve
    phi sectors of every (ring, theta) shell (shape_context.py:172-181; csrc/pm_binning.h: pm_phi_perm) — exact copies, made
    by an index gather.  Only valid where gather_fixed_descriptors' bit-for-bit check of that relation has passed."""
    global _PHI
    import torch
    if _PHI is None:
        q = np.arange(12)
        perms = [q, (q + 6) % 12, 11 - q, (5 - q) % 12]        # frame f's sector p reads frame 1's sector perms[f][p] (involutions / half turn)
        _PHI = [np.concatenate([s * 12 + p for s in range(30)]) for p in perms]
    return torch.stack([sc1[:, torch.as_tensor(_PHI[f], device=sc1.device)] for f in range(n_frames)])


def assign_streamed(be, sc_m, sc_f, bounds, group=None, info=None, local_matrix=None, accept_near_ties=False, pairings=(0, 1, 2, 3)):
    """Cost matrices and assignments two matrices at a time, for clouds whose eight matrices (64 N M bytes per rank-block) do not
    fit in HBM together: for each pairing t the hypothesis and its twin are built (be.chi2_cost_pair: a quarter of the
    eight-matrix launch, the same bits), assigned with the matrices resident (one GPU: lsap.solve_pair_on_device; sharded:
    lsap_sharded.solve_pair_sharded, nothing gathered) and released.  -> list of eight (row_ind, col_ind), widget order.
    A hypothesis that can be neither certified nor (for its size) handed to the dense solver raises.
    pairings: which of the four (hypothesis, twin) pairings to build and assign (the others' slots stay None) — the sharded filter
    route's exact fallback asks for one."""
    from . import lsap
    rank, world = _world(group)
    n, m = bounds[-1], sc_f.shape[1]
    out = [None] * 8
    routes = [None] * 8
    buf = None
    if world > 1 and n > m:
        # More moving than fixed nuclei: the solver wants the short side as its rows (SciPy transposes likewise), and the rows must
        # be what is sharded.  chi-square is symmetric in its two descriptors BIT FOR BIT ((a - b)^2 = (b - a)^2, a + b = b + a,
        # same bin order: test_cost_symmetry_property_large), so the transposed matrices are simply built with the roles
        # swapped: every rank gathers the moving descriptors (2 frames x N x 2 880 B), keeps its block of FIXED rows and fills
        # U_h^T [M_g, N] for the pairing's hypothesis and twin with the one-matrix kernel; the sharded solve then runs on fixed-row
        # blocks.  (Twice the arithmetic of the frame-symmetric kernel — only this orientation pays it.)
        from .lsap_sharded import solve_pair_sharded
        lm = local_matrix or lsap.DeviceMatrix
        bm = shard_bounds(m, world)
        sc_m_full = all_gather_rows(sc_m, bounds, 1, group)                           # [2, N, 360]
        f_all = sc_f if sc_f.shape[0] == 4 else expand_frames(sc_f[0], 4)             # (frame 1 only travelled: derive 2..4)
        f_loc = f_all[:, bm[rank]:bm[rank + 1]]
        for t in pairings:
            h, twin = PAIRINGS[t]
            (ah, bh), (at, bt) = ((int(c) - 1 for c in HYPOTHESES[h]), (int(c) - 1 for c in HYPOTHESES[twin]))
            UT_h = be.chi2_cost_single(f_loc[bh], sc_m_full[ah])                      # [M_g, N] = U_h[:, block]^T
            UT_t = be.chi2_cost_single(f_loc[bt], sc_m_full[at])
            ih = {}
            c_h, c_t = solve_pair_sharded(lm(UT_h), lm(UT_t), bm, n, group, h % world, ih, accept_near_ties=accept_near_ties,
                                          settle_near_ties=n * m > lsap.DENSE_FALLBACK_MAX_ENTRIES)
            for idx, c4r, UT in ((h, c_h, UT_h), (twin, c_t, UT_t)):
                if c4r is None and n * m <= lsap.DENSE_FALLBACK_MAX_ENTRIES and getattr(UT, "is_cuda", False):
                    # ties / near-ties the sharded scheme cannot settle, and a matrix the dense solver can take: the blocks of
                    # U^T go to the hypothesis's owner; SciPy solves an N > M problem through its transpose as well
                    _, c4r = _gather_and_solve(UT, bm, idx % world, group)
                    routes[idx] = "gathered (transposed)"
                elif c4r is None:
                    raise RuntimeError("hypothesis %s: the assignment could not be certified unique (an alternative within ~1e-11 of the "
                                       "optimum, exact ties or non-finite costs) and the matrix is too large for the dense solver; "
                                       "accept_near_ties=True takes the certified optimum as it is" % HYPOTHESES[idx])
                else:
                    routes[idx] = "sharded device (transposed: fixed rows sharded)"
                out[idx] = lsap._answer(np.asarray(c4r), n, m)                       # fixed j -> moving i, as SciPy reports an N > M problem
            del UT_h, UT_t
        if info is not None:
            info["routes"] = routes
            info["mode"] = "streamed: two matrices resident at a time"
        return out
    for t in pairings:
        h, twin = PAIRINGS[t]
        U2 = be.chi2_cost_pair(sc_m, sc_f, t, out=buf)
        buf = U2
        ih, it = {}, {}
        if world == 1:
            got = lsap.solve_pair_on_device(U2[0], U2[1], ih, it, accept_near_ties=accept_near_ties)
            routes[h], routes[twin] = ih.get("route"), it.get("route")
        else:
            from .lsap_sharded import solve_pair_sharded
            lm = local_matrix or lsap.DeviceMatrix
            c_h, c_t = solve_pair_sharded(lm(U2[0]), lm(U2[1]), bounds, m, group, h % world, ih, accept_near_ties=accept_near_ties,
                                          settle_near_ties=n * m > lsap.DENSE_FALLBACK_MAX_ENTRIES)
            rows = np.arange(n, dtype=np.int64)
            got = [None if c_h is None else (rows, np.asarray(c_h, dtype=np.int64)),
                   None if c_t is None else (rows, np.asarray(c_t, dtype=np.int64))]
            routes[h] = "sharded device" if c_h is not None else "uncertified"          # (details of the route: on the root only)
            routes[twin] = "sharded device" if c_t is not None else "uncertified"
            for k2, idx in enumerate((h, twin)):
                if got[k2] is None and n * m <= lsap.DENSE_FALLBACK_MAX_ENTRIES and getattr(U2, "is_cuda", False):
                    got[k2] = _gather_and_solve(U2[k2], bounds, idx % world, group)   # ties at a size the dense solver can take
                    routes[idx] = "gathered"
        for idx, g in zip((h, twin), got):
            if g is None:
                raise RuntimeError("hypothesis %s: the assignment could not be certified unique (an alternative within ~1e-11 of the "
                                   "optimum, exact ties or non-finite costs) and the matrix is too large for the dense solver; "
                                   "accept_near_ties=True takes the certified optimum as it is" % HYPOTHESES[idx])
            out[idx] = g
    if info is not None:
        info["routes"] = routes
        info["mode"] = "streamed: two matrices resident at a time"
    return out


def _gather_and_solve(block, bounds, owner, group):
    """One hypothesis whose sharded solve could not certify its answer (ties, non-finite costs): its row blocks [rows_g, C] are
    assembled on `owner` and solved there (lsap.solve_on_device: the device scheme once more on the whole matrix, else SciPy's
    algorithm itself), the answer goes to every rank.  Only for matrices the dense solver can take (the caller checks).
    -> (row_ind, col_ind) of the R x C problem the blocks belong to (R = bounds[-1]); a solver refusal is raised on every rank."""
    import torch
    from .lsap import linear_sum_assignment, solve_on_device
    dist = _dist()
    rank, world = _world(group)
    biggest = max(bounds[g + 1] - bounds[g] for g in range(world))
    cols = block.shape[1]
    padded = torch.zeros((biggest, cols), dtype=block.dtype, device=block.device)
    padded[:block.shape[0]].copy_(block)
    whole = torch.empty((world, biggest, cols), dtype=block.dtype, device=block.device) if owner == rank else None
    dist.gather(padded, [whole[g] for g in range(world)] if owner == rank else None, dst=_global_rank(group, owner), group=group)
    status = torch.zeros(1, dtype=torch.int32, device=block.device)
    k = min(bounds[-1], cols)
    buf = torch.zeros((2, k), dtype=torch.int64, device=block.device)
    if owner == rank:
        U = torch.cat([whole[g][:bounds[g + 1] - bounds[g]] for g in range(world)], dim=0)
        try:
            r, c = solve_on_device(U) if U.is_cuda else linear_sum_assignment(U.numpy())
            buf[0], buf[1] = torch.as_tensor(r, device=block.device), torch.as_tensor(c, device=block.device)
        except ValueError as e:
            status[0] = 2 if "infeasible" in str(e) else 1
    dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
    if int(status.item()):
        raise ValueError("cost matrix is infeasible" if int(status.item()) == 2 else "matrix contains invalid numeric entries")
    dist.broadcast(buf, src=_global_rank(group, owner), group=group)
    return buf[0].cpu().numpy(), buf[1].cpu().numpy()


def iter_cost_blocks(be, mov, fix, rows_per_block, group=None):
    """The same cost rows in slabs, for sizes at which a rank's eight row blocks do not fit in HBM at once
    (BASELINE config 4: 200k x 200k on 8 GPUs is 8 x 40 GB per rank).  Yields (first_row, U [8, r, M]) with r <=
    rows_per_block; first_row is a global moving-row index.  The slab buffer is reused: consume (reduce, copy out)
    each slab before advancing.  Every slab is bit-identical to the corresponding rows of build_costs."""
    import torch
    sc_m, sc_f, bn = build_descriptors(be, mov, fix, group)
    rank, _ = _world(group)
    rows = sc_m.shape[1]
    rows_per_block = max(1, min(int(rows_per_block), max(rows, 1)))
    buf = big_empty((8, rows_per_block, sc_f.shape[1]), torch.float64, sc_m.device)
    for r0 in range(0, rows, rows_per_block):
        r1 = min(rows, r0 + rows_per_block)
        out = buf[:, :r1 - r0]
        yield bn[rank] + r0, be.chi2_cost8(sc_m[:, r0:r1].contiguous(), sc_f, out=out)


def cost_row_argmins(be, mov, fix, rows_per_block=None, group=None):
    """np.argmin(U_h, axis=1) for the eight cost matrices without ever holding them: each rank streams its rows
    through a slab of `rows_per_block` rows (default: its whole block), reduces the slab on the device and keeps 4
    bytes per (hypothesis, row).  Returns int32 [8, N], identical on every rank (one all-gather of the indices).
    This is the greedy correspondence for clouds beyond the assignment solver's reach, and the quantity
    BASELINE.json's 200k-point configuration is checked on."""
    import torch
    rank, world = _world(group)
    n = mov.shape[1]
    bn = shard_bounds(n, world)
    rows = bn[rank + 1] - bn[rank]
    idx = torch.empty((8, rows), dtype=torch.int32, device=mov.device)
    for first, U in iter_cost_blocks(be, mov, fix, rows if rows_per_block is None else rows_per_block, group):
        r0 = first - bn[rank]
        idx[:, r0:r0 + U.shape[1]] = be.row_argmin(U)
    return all_gather_rows(idx, bn, 1, group)


# the four pairings of a sharded filter-route registration with their roots' host solvers side by side (round 5); PM_SHARDED_SEQUENTIAL=1:
# one pairing after the other (round 5's first form; the A/B switch)
SHARDED_PAIRINGS_CONCURRENT = os.environ.get("PM_SHARDED_SEQUENTIAL") != "1"
SHARDED_ASSIGN_MIN_ROWS = 1024     # below this the gather + dense host solve is quicker than the sharded solve's round trips


def assign(U_loc, bounds, group=None, info=None, local_matrix=None, accept_near_ties=False, ready=None):
    """linear_sum_assignment on each of the eight matrices (_dock_widget.py:604-611) -> list of
    (row_ind, col_ind) int64 arrays, identical on every rank.  On a GPU the matrices never leave HBM: a sparse core
    of each is solved on the host and certified against every entry on the device (lsap.solve_on_device; tied or small
    matrices fall back to SciPy's algorithm restated in C++, lsap.linear_sum_assignment: identical indices either way).
    Sharded: the matrices stay where the cost build left them — row blocks on their ranks — and only the sparse core's
    candidates, the column duals and the certificate's counters travel (lsap_sharded.py; N <= M); a hypothesis that cannot
    be certified that way (ties, non-finite costs, N > M, small clouds) is assembled on rank h mod G and solved there.
    info (dict): which route each hypothesis took.  local_matrix: how to query a rank's block (default: lsap.DeviceMatrix
    for GPU tensors; the CPU tests pass a NumPy double)."""
    import torch
    from .lsap import solve_eight_on_device, solve_on_device
    rank, world = _world(group)
    if world == 1:
        if U_loc.is_cuda:
            # matrices stay in HBM: sparse-core solves driven from four host threads, certified on the device against every
            # entry (lsap.solve_eight_on_device); small or tied matrices take the dense host solver, SciPy's algorithm itself
            out = solve_eight_on_device(U_loc, info=info, accept_near_ties=accept_near_ties, ready=ready)
            for h, ans in enumerate(out):
                if ans is None:          # only for matrices beyond the dense solver's reach (lsap.DENSE_FALLBACK_MAX_ENTRIES)
                    raise RuntimeError("hypothesis %s: the assignment could not be certified unique (an alternative within ~1e-11 of "
                                       "the optimum, exact ties or non-finite costs) and the matrix is too large for the dense "
                                       "solver; accept_near_ties=True takes the certified optimum as it is" % HYPOTHESES[h])
            return out
        return solve_many([U_loc[h].numpy() for h in range(8)])
    dist = _dist()
    n = bounds[-1]
    done = {}
    if local_matrix is None and U_loc.is_cuda:
        from .lsap import DeviceMatrix as local_matrix
    if local_matrix is not None and n <= U_loc.shape[2] and n >= SHARDED_ASSIGN_MIN_ROWS:
        from .lsap import TWINS, DENSE_FALLBACK_MAX_ENTRIES as DENSE_FALLBACK_MAX
        from .lsap_sharded import solve_pair_sharded
        routes = {}
        for twin, h in sorted(TWINS.items(), key=lambda kv: kv[1]):
            pinfo = {} if info is not None else None
            c_h, c_t = solve_pair_sharded(local_matrix(U_loc[h]), local_matrix(U_loc[twin]), bounds, U_loc.shape[2], group, h % world, pinfo,
                                          settle_near_ties=n * U_loc.shape[2] > DENSE_FALLBACK_MAX)     # (else: gathered below, SciPy's algorithm)
            rows = np.arange(n, dtype=np.int64)
            if c_h is not None:
                done[h] = (rows, np.asarray(c_h, dtype=np.int64))
                routes[h] = "sharded device"
            if c_t is not None:
                done[twin] = (rows, np.asarray(c_t, dtype=np.int64))
                routes[twin] = "sharded device"
        if info is not None:
            info["routes"] = [routes.get(h, "gathered") for h in range(8)]
    biggest = max(bounds[g + 1] - bounds[g] for g in range(world))
    mine = {}
    for h in range(8):
        if h in done:
            continue
        owner = h % world
        padded = torch.zeros((biggest, U_loc.shape[2]), dtype=U_loc.dtype, device=U_loc.device)
        padded[:U_loc.shape[1]].copy_(U_loc[h])
        # the owner receives the blocks into ONE allocation; with equal blocks (N divisible by the ranks) that allocation IS
        # the matrix, otherwise the padding rows are squeezed out by one copy
        whole = torch.empty((world, biggest, U_loc.shape[2]), dtype=U_loc.dtype, device=U_loc.device) if owner == rank else None
        blocks = [whole[g] for g in range(world)] if owner == rank else None
        dist.gather(padded, blocks, dst=_global_rank(group, owner), group=group)   # row blocks -> the owner only
        del padded
        if owner == rank:    # keep the assembled matrix (on the device if that is where it is); solve after all gathers so ranks solve concurrently
            if all(bounds[g + 1] - bounds[g] == biggest for g in range(world)):
                mine[h] = whole.view(world * biggest, U_loc.shape[2])
            else:
                mine[h] = torch.cat([blocks[g][:bounds[g + 1] - bounds[g]] for g in range(world)], dim=0)
            del blocks, whole
    # A solver refusal (NaN / -inf costs from degenerate descriptors, infeasible matrix) on the owner of one hypothesis
    # must not leave the other ranks waiting in the broadcasts below: collect a status word per hypothesis, agree on it,
    # and raise the same exception everywhere.
    status = torch.zeros(8, dtype=torch.int32, device=U_loc.device)
    for h in list(mine):
        try:
            mine[h] = solve_on_device(mine[h]) if mine[h].is_cuda else linear_sum_assignment(mine[h].numpy())
        except ValueError as e:
            status[h] = 2 if "infeasible" in str(e) else 1
            mine[h] = None
    dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
    bad = status.cpu().numpy()
    if bad.any():
        h = int(np.flatnonzero(bad)[0])
        raise ValueError("hypothesis %s: %s" % (HYPOTHESES[h], "cost matrix is infeasible" if bad[h] == 2
                                                 else "matrix contains invalid numeric entries"))
    k = min(n, U_loc.shape[2])
    out = []
    for h in range(8):
        if h in done:
            out.append(done[h])
            continue
        buf = torch.zeros((2, k), dtype=torch.int64, device=U_loc.device)
        if h in mine:
            buf[0] = torch.as_tensor(mine[h][0], device=U_loc.device)
            buf[1] = torch.as_tensor(mine[h][1], device=U_loc.device)
        dist.broadcast(buf, src=_global_rank(group, h % world), group=group)
        out.append((buf[0].cpu().numpy(), buf[1].cpu().numpy()))
    return out


# Below this many moving points every rank simply runs the whole ICP itself: with the grid search one iteration
# over 50 000 points costs ~80 us on one GPU, less than the latency of the collective a sharded iteration needs
# ("latency-bound, so for N <= 50k single-GPU ICP is preferable").  Replicas give identical results.
ICP_SHARD_MIN_POINTS = 400_000


class PlanarCloud(ValueError):
    """icp_sharded: the moving cloud is (nearly) planar — the fit needs the reference's pinv on the host, which the sharded loop
    does not have.  Raised on every rank together; estimate_transform answers it with the replicated loop."""


def icp_sharded(be, moved, fix, iters, group=None):
    """Affine ICP with the moving rows sharded (perform_icp.py:7-26).  -> (A_icp [4,4], residuals [iters])."""
    import torch
    rank, world = _world(group)
    dist = _dist() if world > 1 else None
    bn = shard_bounds(moved.shape[1], world)
    loc = moved[:, bn[rank]:bn[rank + 1]].contiguous().clone()
    grid = be.icp_grid(fix) if (iters and hasattr(be, "icp_grid")) else None     # owned by this run
    A_icp = torch.eye(4, dtype=torch.float64, device=moved.device).reshape(16).contiguous()
    origin = torch.cat([fix[:, 0], fix[:, 0]]).contiguous()
    # One collective per iteration: the 24 moment sums of this iteration travel together with the residual
    # parts of the previous one.  Every rank reduces the same gathered [G, 26] block with the same kernel, so all
    # ranks hold bit-identical sums and solve the identical 4x4 — they stay in lockstep without a broadcast.
    mine = torch.zeros(26, dtype=torch.float64, device=moved.device)       # [0:24] moment sums, [24:26] residual parts
    sums_view, rp_view = mine[:24], mine[24:]
    gathered = torch.empty(world * 26, dtype=torch.float64, device=moved.device) if world > 1 else None     # (rank-major concatenation)
    res_buf = torch.zeros((max(iters, 1), 2), dtype=torch.float64, device=moved.device)
    status = torch.zeros(1, dtype=torch.int32, device=moved.device) if moved.is_cuda else None
    for it in range(iters + (1 if world > 1 and iters else 0)):
        last = it == iters
        if not last:
            nn = be.icp_nn(loc, fix, grid) if grid is not None else be.icp_nn(loc, fix)
            be.icp_accumulate(loc, fix, nn, origin, out=sums_view)
        if world > 1:
            dist.all_gather_into_tensor(gathered, mine, group=group)
            total = gathered.view(world, 26).sum(0)
            if it > 0:
                res_buf[it - 1].copy_(total[24:])
            sums = total[:24]
        else:
            sums = sums_view
        if not last:
            be.icp_update(sums, origin, loc, fix, nn, A_icp, parts_out=rp_view, status=status)
            if world == 1:
                res_buf[it].copy_(rp_view)
    if status is not None and int(status.item()) != 0:       # identical sums on every rank: all ranks leave together
        raise PlanarCloud("sharded ICP met a (nearly) planar moving cloud: the reference's pinv fit is needed there; run the "
                          "refinement unsharded (icp_shard_min_points above the cloud size)")
    residuals = list((res_buf[:iters, 0] / res_buf[:iters, 1]).unbind(0)) if iters else []
    res = torch.stack(residuals) if residuals else torch.empty(0, dtype=torch.float64, device=moved.device)
    return A_icp.reshape(4, 4), res


def pca_alignment(moving, fixed):
    """The widget's PCA-only branch (_dock_widget.py:722-731): sklearn PCA(3).components_ of each cloud (the widget centres them
    first; PCA centres again) -> (moving_transform 3 x 3, fixed_transform 3 x 3), exported by io.save_pca_transforms.  sklearn's
    own NumPy calls on the caller's arrays (shape_context.pca_components_host: the reference's bits, O(N) host work); the device
    kernel pm_pca_components (1e-11) remains a C-ABI entry."""
    from .estimate_transform.shape_context import pca_components_host, pca_view
    mt, ft = pca_components_host(pca_view(moving)), pca_components_host(pca_view(fixed))
    if nat.is_torch(moving):
        return nat.to_dev(mt, dev=moving.device if moving.is_cuda else None), nat.to_dev(ft, dev=moving.device if moving.is_cuda else None)
    return mt, ft


class _SampleDraws:
    """The index sets of the eight RANSAC runs, drawn on a helper thread while the Hungarian solves run.

    What do_ransac draws depends only on the number of matched pairs (min(N, M)), min_samples and trials — not on the
    matching — so the 8 x trials np.random.choice calls (each a full permutation of the pair list: the dominant host cost
    after the solver) need not wait for it.  The draws come from NumPy's global generator in exactly the order the
    reference consumes it (optional seed, then run 11, 12, ... 24), under _RNG_LOCK.  `private` (seeded runs of a
    batch): draw from a RandomState(seed) of the run's own instead — the same sets, but runs no longer queue for the
    global generator (whose state after a batch of concurrent runs would be order-dependent anyway)."""

    def __init__(self, be, n_pairs, min_samples, trials, seed, private=False):
        import threading
        self.sets, self.error, self.seconds = None, None, 0.0

        def work():
            import time
            t0 = time.perf_counter()
            try:
                _work()
            finally:
                self.seconds = time.perf_counter() - t0

        def _work():
            try:
                if trials <= 0:
                    self.sets = [None] * 8
                elif private and seed is not None:
                    rng = np.random.RandomState(seed)
                    self.sets = [be.draw_samples(n_pairs, min_samples, trials, rng=rng) for _ in range(8)]
                else:
                    with _RNG_LOCK:
                        if seed is not None:
                            np.random.seed(seed)
                        self.sets = [be.draw_samples(n_pairs, min_samples, trials) for _ in range(8)]
            except BaseException as e:                # e.g. min_samples > n: raised where the reference would raise it
                self.error = e

        self.thread = threading.Thread(target=work, name="pm-ransac-draws")
        self.thread.start()

    def result(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.sets


def _shared_device_seed(group, device):
    """64 bits for the device sampler, the same on every rank: rank 0 takes them from NumPy's global generator
    (shape_context.fresh_device_seed) and broadcasts them (8 bytes) — unseeded ranks would otherwise draw different index
    sets, fit different A_sc and refine different clouds while exchanging moment sums as if they were one."""
    from .estimate_transform.shape_context import fresh_device_seed
    rank, world = _world(group)
    with _RNG_LOCK:
        seed = fresh_device_seed()
    if world == 1:
        return seed
    import torch
    dist = _dist()
    on_host = dist.get_backend(group) == "gloo"
    t = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=torch.int64, device="cpu" if on_host else device)
    dist.broadcast(t, src=_global_rank(group, 0), group=group)
    return int(t.item()) & ((1 << 64) - 1)


COST_MODES = ('auto', 'exact', 'relaxed', 'filter')


@dataclasses.dataclass
class Options:
    """What estimate_transform takes BESIDE the reference's arguments (and seed / details / cost_mode / group): pass
    options=Options(...) or a dict of these fields.

    backend         the compute backend (default: GpuBackend() on the current device; the CPU tests pass a double)
    sampler         where the RANSAC index sets come from.  'numpy': NumPy's global generator, call for call as the reference
                    consumes it (8 x trials np.random.choice calls = full shuffles, drawn on a helper thread) — with a seed, the
                    reference's own sets; 'device': drawn on the GPU in front of each trial's fit (Philox + Floyd's subset
                    algorithm, keyed by 64 bits from NumPy's global generator, or by `seed` if one is given); 'auto' (default):
                    'numpy' when a seed is given — the reference's seeded result is reproduced bit for bit —, 'device' when
                    not: the reference never seeds, so an unseeded run is random there too
    private_rng     with a seed: draw from a private RandomState(seed) and leave NumPy's global generator untouched
                    (same index sets; what estimate_transform_batch uses so that concurrent runs do not queue)
    accept_near_ties  for matrices beyond the reach of SciPy's dense algorithm (> 2^30 entries): if a hypothesis has a second
                    assignment within ~1e-11 of the optimal cost, which of the two SciPy's rounding would return cannot be
                    told; False raises, True takes the certified optimum (details['assignment']['routes'] says so)
    stream_hypotheses  None: keep all matrices of the chosen build resident if they fit in HBM, else build them pairing by
                    pairing; True / False force one or the other
    keep_cost_buffer  large registrations (>= 8 GiB of cost matrices) write them into a buffer this module keeps per (device,
                    stream) between calls (COST_CACHE_MIN_BYTES; release_cost_buffers() frees it, reserve() makes it ahead of
                    the first call); False: a fresh allocation per call, returned to torch's allocator afterwards
    icp_shard_min_points  moving-cloud size from which a sharded run shards ICP too (below it every rank runs it whole)
    icp_one_launch  None: perform_icp.ONE_LAUNCH decides (default False: one launch per iteration); True: iterations 1 .. n-1 of
                    the Affine ICP loop in one launch of persistent workgroups (only for a device that does nothing else meanwhile;
                    estimate_transform_batch always passes False) — identical results"""
    backend: object = None
    sampler: str = 'auto'
    private_rng: bool = False
    accept_near_ties: bool = False
    stream_hypotheses: object = None
    keep_cost_buffer: bool = True
    icp_shard_min_points: int = ICP_SHARD_MIN_POINTS
    icp_one_launch: object = None

    @classmethod
    def of(cls, options):
        if options is None:
            return cls()
        if isinstance(options, cls):
            return options
        if isinstance(options, dict):
            return cls(**options)                      # (an unknown field is a TypeError naming it)
        raise TypeError("options must be a pipeline.Options, a dict of its fields, or None")


def resolve_cost_mode(cost_mode, n, m, world, be):
    """Which build the assignment stage STARTS from -> 'filter' | 'relaxed' | 'exact'.  'auto' (the default) and 'filter': the
    float32 filter matrices from FILTER_MIN_POINTS points on (one GPU or sharded), the relaxed float64 build from
    RELAXED_MIN_POINTS on (one GPU), the exact build below; 'relaxed': that build or the exact one; 'exact': always the exact
    one.  Every route returns the exact build's assignment vectors — a pairing the cheaper build cannot PROVE (ties, near-ties
    inside the exact mode's own margin, non-finite costs, descriptors whose frames do not permute) has its exact matrices built."""
    if cost_mode not in COST_MODES:
        raise ValueError("cost_mode must be one of %s" % (COST_MODES,))
    small = min(int(n), int(m))
    want = 'filter' if cost_mode == 'auto' else cost_mode
    if want == 'filter' and small >= FILTER_MIN_POINTS and hasattr(be, "chi2_filter4"):
        return 'filter'
    if want in ('filter', 'relaxed') and world == 1 and small >= RELAXED_MIN_POINTS and hasattr(be, "chi2_cost8_relaxed"):
        return 'relaxed'
    return 'exact'


def filter_bytes(n, m, rows_short=None):
    """Device memory of the filter route at its peak on one rank: the four filter matrices of its block of the SHORT side's rows
    (float32: 16 bytes per row and column; PM_FILTER_F64=1: 32) and, at the worst, one pairing's two exact matrices beside them."""
    short, long_ = min(n, m), max(n, m)
    rows = short if rows_short is None else rows_short
    return (16.0 if FILTER_STORAGE_F32 else 32.0) * rows * long_, 16.0 * rows * long_


def _filter_dtype():
    import torch
    return torch.float32 if FILTER_STORAGE_F32 else torch.float64


def _decide_streamed(be, need, opts, group, device):
    """All matrices of the chosen build at once, or pairing by pairing?  One rank that must stream makes all of them stream (free
    memory and row counts differ from rank to rank: near the threshold they would enter different sequences of collectives)."""
    if opts.stream_hypotheses is not None:
        return bool(opts.stream_hypotheses)
    streamed = hasattr(be, "free_bytes") and hasattr(be, "chi2_cost_pair") and need > 0.85 * be.free_bytes()
    return bool(agree_max(1 if streamed else 0, group, device))


def correspondences_one_gpu(be, mov, fix, sc_m, sc_f, bn, mode, opts, a_info, mark=None, early=None):
    """The eight assignments (_dock_widget.py:547-611) on one GPU from the descriptors -> list of (row_ind, col_ind).
    mode: resolve_cost_mode's answer (already 'exact' where the frames do not permute)."""
    import torch
    from . import lsap
    n, m = mov.shape[1], fix.shape[1]
    on_gpu = bool(getattr(mov, "is_cuda", False))
    full = cost_bytes(sc_m.shape[1], n, m, 1)
    if mode == 'filter':
        f_bytes, x_bytes = filter_bytes(n, m)
        need = f_bytes + x_bytes
    else:
        f_bytes, need = 0.0, full
    want = kept_bytes_wanted(mode, n, m)                                   # (bytes actually written into the kept buffer)
    lease = early.result() if early is not None else None
    if lease is not None and (lease.view.numel() * 8 < want or opts.stream_hypotheses):
        lease.release()                                                      # (asked for before the descriptors said which build it would be)
        lease = None
    # with the buffer already in hand the matrices are resident by construction; otherwise: do they fit?
    streamed = False if lease is not None else _decide_streamed(be, need, opts, None, mov.device)
    U = relaxed_delta = None
    try:
        if not streamed:
            if lease is None and opts.keep_cost_buffer and on_gpu and want >= COST_CACHE_MIN_BYTES:
                lease = cost_buffer(mov.device, ((int(want) + 7) // 8,))     # None: another registration holds it
            if mode == 'filter':
                # the filter matrices with the SHORT side as rows (N > M: the descriptors' roles swapped — the terms are symmetric
                # and every pairing's bin map is an involution, so that is the transposed filter to within its bound), carved
                # out of the kept buffer when there is one
                fshape = (4, min(n, m), max(n, m))
                fout = None if lease is None else lease.view.view(_filter_dtype())[:4 * n * m].view(fshape)
                a_, b_ = (sc_m[0], sc_f[0]) if n <= m else (sc_f[0], sc_m[0])
                U = be.chi2_filter4(a_, b_, out=fout, dtype=_filter_dtype())
            else:
                out8 = None if lease is None else lease.view[:8 * n * m].view(8, n, m)
                if mode == 'relaxed':
                    U, relaxed_delta = be.chi2_cost8_relaxed(sc_m, sc_f, out=out8)
                else:
                    U = be.chi2_cost8(sc_m, sc_f, out=out8)
        if mark is not None:
            mark("gpu_descriptors_costs")
        if mode == 'filter':
            sc_m1, sc_f1 = sc_m[0], sc_f[0]

            def entries_np(t):
                return lambda rows, cols: tuple(x.cpu().numpy() for x in be.chi2_entries(sc_m1, sc_f1, t, rows, cols))

            def entries_t(t):              # (index lists from the library's own kernels: no range check, no read-back)
                return lambda rows, cols: be.chi2_entries(sc_m1, sc_f1, t, rows, cols, trusted=True)

            def build_pairing(t, out):     # streamed: one pairing's filter matrix, the short side as its rows
                return be.chi2_filter_pair(sc_m1, sc_f1, t, out=out) if n <= m else be.chi2_filter_pair(sc_f1, sc_m1, t, out=out)
            in_flight = 4
            if streamed:
                in_flight = max(1, min(4, int(0.85 * be.free_bytes() // (f_bytes / 4.0))))
            lsa = lsap.solve_four_filtered(U, entries_np, entries_t, be.chi2_filter_delta() + 1e-13,
                                           lambda t: be.chi2_cost_pair(sc_m, sc_f, t), info=a_info, accept_near_ties=opts.accept_near_ties,
                                           build=build_pairing, shape=(n, m), device=mov.device, in_flight=in_flight, storage=_filter_dtype())
            if a_info is not None and streamed:
                a_info["mode"] = "streamed: %d filter matri%s resident at a time" % (in_flight, "x" if in_flight == 1 else "ces")
        elif streamed:
            lsa = assign_streamed(be, sc_m, sc_f, bn, None, info=a_info, local_matrix=getattr(be, "local_matrix", None),
                                  accept_near_ties=opts.accept_near_ties)
        elif mode == 'relaxed':
            sc_m1, sc_f1 = sc_m[0], sc_f[0]
            pairing_of = {p[0]: t for t, p in enumerate(PAIRINGS)}

            def exact_entries(h):
                # (rows, cols) -> the listed entries of hypothesis h's exact matrix and of its twin's (pm_chi2_entries_sym)
                def fetch(rows, cols):
                    return tuple(x.cpu().numpy() for x in be.chi2_entries(sc_m1, sc_f1, pairing_of[h], rows, cols))
                return fetch
            lsa = lsap.solve_eight_on_device(U, info=a_info, accept_near_ties=opts.accept_near_ties,
                                             exact_entries=exact_entries if RELAXED_CERTIFY_ON_EXACT_ENTRIES else None, cost_delta=relaxed_delta,
                                             min_eps=2.0 * min(n, m) * relaxed_delta,
                                             exact_rebuild=lambda h: be.chi2_cost_pair_into(sc_m1, sc_f1, pairing_of[h], U))
        else:
            lsa = assign(U, bn, None, info=a_info, local_matrix=getattr(be, "local_matrix", None), accept_near_ties=opts.accept_near_ties)
        if any(a is None for a in lsa):
            raise RuntimeError("a hypothesis could not be assigned (see Options.accept_near_ties)")
        return lsa
    finally:
        del U
        if lease is not None:
            if on_gpu:
                torch.cuda.current_stream(mov.device).synchronize()     # the assignment's last passes have read the buffer
            lease.release()


def correspondences_sharded(be, mov, fix, sc_m, sc_f, bn, mode, opts, group, a_info, mark=None):
    """The eight assignments with the moving rows (and the cost rows) sharded over the group's ranks -> the same list on every
    rank.  Nothing of a cost matrix travels unless a hypothesis must go to the dense solver (assign)."""
    rank, world = _world(group)
    n, m = mov.shape[1], fix.shape[1]
    lm = getattr(be, "local_matrix", None)
    if mode == 'filter':
        rows_short = (bn[rank + 1] - bn[rank]) if n <= m else (shard_bounds(m, world)[rank + 1] - shard_bounds(m, world)[rank])
        f_bytes, x_bytes = filter_bytes(n, m, rows_short)
        streamed = _decide_streamed(be, f_bytes + x_bytes, opts, group, mov.device)
        if mark is not None:
            mark("gpu_descriptors_costs")
        return assign_sharded_filtered(be, sc_m, sc_f, bn, group, streamed=streamed, info=a_info, local_matrix=lm,
                                       accept_near_ties=opts.accept_near_ties)
    need = cost_bytes(sc_m.shape[1], n, m, world)
    streamed = _decide_streamed(be, need, opts, group, mov.device)
    U = None if streamed else be.chi2_cost8(sc_m, sc_f)
    if mark is not None:
        mark("gpu_descriptors_costs")
    if streamed:
        return assign_streamed(be, sc_m, sc_f, bn, group, info=a_info, local_matrix=lm, accept_near_ties=opts.accept_near_ties)
    return assign(U, bn, group, info=a_info, local_matrix=lm, accept_near_ties=opts.accept_near_ties)


def assign_sharded_filtered(be, sc_m, sc_f, bounds, group, streamed=False, info=None, local_matrix=None, accept_near_ties=False):
    """cost_mode 'auto' / 'filter' on several ranks.  Every rank builds ITS ROW BLOCK of the four float32 filter
    matrices (pm_chi2_filter4_f32 on its rows of the short side's frame-1 descriptors against all of the long side's: 16 bytes per
    row and column, a quarter of the exact block) and answers the root solver's selection queries from it
    (lsap_sharded.solve_pair_sharded_filtered); every COST the solver or the certificate uses is evaluated exactly on the root,
    which holds both clouds' frame-1 descriptors (the moving ones are all-gathered: 2 880 N bytes, 144 MB at 50 000 — the only
    exchange beside the queries' answers).  A pairing that cannot be proven there (ties, near-ties) has its two exact matrices
    built in row blocks and goes the exact mode's sharded way (assign_streamed on that pairing alone).
    N > M: the short side is the fixed cloud — its rows are sharded instead, the filter is built with the roles swapped.
    streamed: one pairing's filter block at a time (pm_chi2_filter_pair) in a reused buffer.
    -> list of eight (row_ind, col_ind), identical on every rank."""
    from . import lsap
    from .lsap_sharded import solve_pair_sharded_filtered
    rank, world = _world(group)
    n, m = bounds[-1], sc_f.shape[1]
    lm = local_matrix or lsap.DeviceMatrix
    sc_f1 = sc_f[0]
    sc_m1_full = all_gather_rows(sc_m[:1].contiguous(), bounds, 1, group)[0]           # [N, 360] on every rank
    if n <= m:
        rb, a_loc, b_all = bounds, sc_m[0], sc_f1
    else:
        rb = shard_bounds(m, world)
        a_loc, b_all = sc_f1[rb[rank]:rb[rank + 1]].contiguous(), sc_m1_full
    delta = be.chi2_filter_delta() + 1e-13
    F4 = None if streamed else be.chi2_filter4(a_loc, b_all, dtype=_filter_dtype())
    buf = None
    out, routes, details = [None] * 8, [None] * 8, [dict() for _ in range(8)]
    fallback = []

    def fetcher(t):
        def fetch(rows, cols):              # (root only) exact entries of the SHORT-side-by-long-side problem the solver sees
            r, c = (rows, cols) if n <= m else (cols, rows)
            return tuple(np.asarray(x.cpu().numpy() if nat.is_torch(x) else x, dtype=np.float64) for x in be.chi2_entries(sc_m1_full, sc_f1, t, r, c))
        return fetch

    on_gpu = bool(getattr(sc_f1, "is_cuda", False))

    def fetcher_t(t):                       # the same with GPU tensors (index lists from the library's own kernels / the ranks' answers)
        if not on_gpu:
            return None

        def fetch_t(rows, cols):
            r, c = (rows, cols) if n <= m else (cols, rows)
            return be.chi2_entries(sc_m1_full, sc_f1, t, r, c, trusted=True)
        return fetch_t

    together = None
    pinfos = [dict() for _ in range(4)]
    if F4 is not None and SHARDED_PAIRINGS_CONCURRENT:
        # all four pairings at once: the roots' host solvers run concurrently (pairing t on rank t mod G, a thread each), one serving
        # loop per rank answers their queries in turn (lsap_sharded.solve_pairs_sharded_filtered)
        from .lsap_sharded import solve_pairs_sharded_filtered
        import torch
        jobs = [dict(local=lm(F4[t]), exact_entries=fetcher(t), exact_entries_t=fetcher_t(t), bounds=rb, n_cols=max(n, m), root=t % world, info=pinfos[t],
                     device=sc_f1.device if on_gpu else None, stream=torch.cuda.current_stream(sc_f1.device) if on_gpu else None)
                for t in range(4)]
        together = solve_pairs_sharded_filtered(jobs, group, delta)
    for t, (h, twin) in enumerate(PAIRINGS):
        pinfo = pinfos[t]
        if together is not None:
            c_h, c_t = together[t]
        else:
            if F4 is not None:
                Ft = F4[t]
            else:
                Ft = buf = be.chi2_filter_pair(a_loc, b_all, t, out=buf, dtype=_filter_dtype())
            c_h, c_t = solve_pair_sharded_filtered(lm(Ft), fetcher(t), delta, rb, max(n, m), group, t % world, pinfo,
                                                   exact_entries_t=fetcher_t(t), entries_device=sc_f1.device if on_gpu else None)
        if c_h is None or c_t is None:
            fallback.append(t)
            continue
        out[h], out[twin] = lsap._answer(np.asarray(c_h), n, m), lsap._answer(np.asarray(c_t), n, m)
        routes[h], routes[twin] = "sharded device (filter)", "sharded device (filter; sibling's duals certified)"
        for k in (h, twin):
            details[k].update(pinfo if k == h else pinfo.get("twin", {}))
            details[k]["cost_mode"] = "filter (row blocks of the approximate matrix as selector, exact costs on the listed entries)"
    del F4, buf
    if fallback:
        finfo = {}
        got = assign_streamed(be, sc_m, sc_f, bounds, group, info=finfo, local_matrix=local_matrix, accept_near_ties=accept_near_ties,
                              pairings=tuple(fallback))
        for t in fallback:
            for k in PAIRINGS[t]:
                out[k], routes[k] = got[k], finfo["routes"][k]
                details[k]["cost_mode"] = "exact (built: the filtered solve did not certify)"
    if info is not None:
        info["routes"], info["details"] = routes, details
        info["mode"] = "sharded filter: %s" % ("one pairing's row block resident at a time" if streamed else
                                               "four row blocks resident" + (", roots' solvers side by side" if together is not None else ""))
    return out


def assignments(moving, fixed, *, cost_mode='auto', group=None, options=None, details=None):
    """Stages 526-611 of the widget alone: statistics, descriptors, the cost build `cost_mode` starts from and the eight
    linear_sum_assignment results -> list of eight (row_ind, col_ind) in widget order (11 ... 24), on every rank."""
    opts = Options.of(options)
    be = opts.backend or GpuBackend()
    mov, fix = be.cloud(moving), be.cloud(fixed)
    info = None if details is None else details.setdefault("assignment", {})
    return _correspondences(be, mov, fix, moving, fixed, cost_mode, opts, group, info, details)


def _start_early_lease(be, mov, fix, cost_mode, opts, group):
    """Ask for the kept cost buffer NOW, on a helper thread, while the first-call costs, statistics and descriptors run (one GPU,
    buffers of at least COST_CACHE_MIN_BYTES that fit) -> _EarlyLease or None."""
    _, world = _world(group)
    if not (getattr(be, "device_sampler", False) and world == 1 and opts.keep_cost_buffer and opts.stream_hypotheses is None
            and getattr(mov, "is_cuda", False)):
        return None
    n_, m_ = mov.shape[1], fix.shape[1]
    mode0 = resolve_cost_mode(cost_mode, n_, m_, 1, be)
    want0 = kept_bytes_wanted(mode0, n_, m_)
    if want0 < COST_CACHE_MIN_BYTES:
        return None
    extra = filter_bytes(n_, m_)[1] if mode0 == 'filter' else (cost_bytes(n_, n_, m_, 1) - 64.0 * n_ * m_)
    if want0 + extra > 0.85 * be.free_bytes():
        return None
    return _EarlyLease(mov.device, want0)


def _correspondences(be, mov, fix, moving, fixed, cost_mode, opts, group, a_info, details, mark=None, early=None):
    if early is None:
        early = _start_early_lease(be, mov, fix, cost_mode, opts, group)
    try:
        return _correspondences_with(be, mov, fix, moving, fixed, cost_mode, opts, group, a_info, details, mark, early)
    except BaseException:
        if early is not None:
            early.cancel()
        raise


def _correspondences_with(be, mov, fix, moving, fixed, cost_mode, opts, group, a_info, details, mark, early):
    rank, world = _world(group)
    gpu = bool(getattr(be, "device_sampler", False))
    guards = [] if gpu else None          # (the GPU backend: its descriptor launches count the neighbours on bin boundaries)
    views = (None, None)
    if gpu:
        from .estimate_transform.shape_context import pca_view
        views = (pca_view(moving), pca_view(fixed))       # what the reference would hand to sklearn: the caller's own arrays
    mode = resolve_cost_mode(cost_mode, mov.shape[1], fix.shape[1], world, be)
    sc_m, sc_f, bn = build_descriptors(be, mov, fix, group, guards=guards, views=views)
    if mode != 'exact':
        # the cheaper builds derive frames 2..4 from frame 1: only where that relation holds bit for bit (sharded: every rank has
        # verified its rows and the verdicts are max-reduced — frame 1 alone travelled)
        ok = (sc_f.shape[0] == 1) if world > 1 else bool(be.chi2_symmetric(sc_m, sc_f))
        if not ok:
            mode = 'exact'
    if a_info is not None:
        a_info["cost_mode"] = mode
    try:
        if world == 1:
            lsa = correspondences_one_gpu(be, mov, fix, sc_m, sc_f, bn, mode, opts, a_info, mark, early)
        else:
            lsa = correspondences_sharded(be, mov, fix, sc_m, sc_f, bn, mode, opts, group, a_info, mark)
    finally:
        del sc_m, sc_f
    if guards:
        _report_edge_guard(guards, details)
    return lsa


def _report_edge_guard(guards, details):
    """Neighbours (of this rank's rows) whose bin the reference itself decides by the rounding noise of its linear algebra: 0
    everywhere = the integer histograms are the reference's by construction for this call."""
    gm, gf = (g.cpu().numpy() for g in guards[:2])
    guard = {"moving": {"ring": int(gm[0]), "sector": int(gm[1])}, "fixed": {"ring": int(gf[0]), "sector": int(gf[1])}}
    if details is not None:
        details["edge_guard"] = guard
    if int(gm.sum()) + int(gf.sum()) > 0:
        import warnings
        warnings.warn("estimate_transform: %d neighbour relations of the moving cloud and %d of the fixed cloud lie on a bin "
                      "boundary of the shape context (ring radius, sector plane, polar cone, or a duplicate of the queried "
                      "nucleus) to within the rounding noise of the reference's own np.linalg.inv (shape_context.py:61-84): "
                      "the reference bins them as its LAPACK build happens to round, so its histograms, the eight assignment "
                      "vectors and the inlier counts are not reproducible for this input (lattice / voxel coordinates, planar "
                      "clouds, duplicates).  What this call returns is self-consistent (the direct projection's histograms, "
                      "their exact costs and optimal assignments) and the final 4 x 4 normally agrees with the reference's to "
                      "ICP's tolerance — it is refitted on nearest neighbours, not on the descriptors; details['edge_guard'] "
                      "has the counts" % (int(gm.sum()), int(gf.sum())), EdgeGuardWarning, stacklevel=5)


def _ransac_stage(be, mov, fix, lsa, sets, on_device, transform, ransac_samples, ransac_trials, ransac_error, seed, group, details):
    """The eight do_ransac runs and the arg-max of their inlier counts (_dock_widget.py:622-703) -> (A_sc, inliers [8])."""
    import torch
    inliers = np.zeros(8, dtype=np.int64)
    A_h = []
    if on_device:
        dseed = (int(seed) & ((1 << 64) - 1)) if seed is not None else _shared_device_seed(group, mov.device)
    # 'Affine': a run's winner is the device's fit of its sample; the hypothesis the registration goes on with gets the
    # reference's own expression on the host afterwards (one read-back instead of eight; the other seven only appear in
    # details["ransac_A"], equal to the reference's to ~1e-12)
    can_defer = transform == 'Affine' and hasattr(be, "refit_winner")
    deferred = [dict() for _ in range(8)]
    pre = [None] * 8
    if (on_device and transform == 'Affine' and int(ransac_samples) >= 4 and int(ransac_trials) > 0 and hasattr(be, "ransac_prelaunch")
            and all(len(r) >= int(ransac_samples) for r, _ in lsa)):
        # all eight hypotheses' fused draw + fit + score launches go out back to back; results are read afterwards
        pre = [be.ransac_prelaunch(mov, fix, r.astype(np.int32), c.astype(np.int32), ransac_trials, ransac_error, ransac_samples, dseed, h)
               for h, (r, c) in enumerate(lsa)]
    for h, (r, c) in enumerate(lsa):
        extra = {"defer": deferred[h]} if can_defer else {}
        if pre[h] is not None:
            extra["prelaunched"] = pre[h]
        if on_device:                            # stream h of the registration's seed: the eight runs draw independent sets
            A, k = be.do_ransac(mov, fix, r.astype(np.int32), c.astype(np.int32), ransac_trials, ransac_error, transform,
                                ransac_samples, device_seed=dseed, run=h, **extra)
        else:
            A, k = be.do_ransac(mov, fix, r.astype(np.int32), c.astype(np.int32), ransac_trials, ransac_error, transform,
                                ransac_samples, samples=sets[h], **extra)
        A_h.append(nat.to_dev(A, dev=mov.device))
        inliers[h] = k
    h_best = int(np.argmax(inliers))             # first maximum (_dock_widget.py:683-703)
    if can_defer and deferred[h_best]:
        A_h[h_best] = be.refit_winner(deferred[h_best])
    if details is not None:
        details.update(lsa=lsa, ransac_A=torch.stack(A_h).cpu().numpy())
    return A_h[h_best], inliers


def _icp_stage(be, mov, fix, moving, A_sc, transform, icp_iterations, opts, group, details):
    """apply_affine_transform + perform_icp (_dock_widget.py:714-717) -> A_icp."""
    _, world = _world(group)
    if transform == 'Similar':
        # this mode's chain is reproduced only by the reference's own NumPy calls (find_transform.similar_transform_host)
        from .estimate_transform.find_transform import apply_affine_host
        mov_h = moving.detach().cpu().numpy() if nat.is_torch(moving) else np.asarray(moving, dtype=np.float64)
        moved = apply_affine_host(np.ascontiguousarray(mov_h[:3]), A_sc.cpu().numpy() if nat.is_torch(A_sc) else np.asarray(A_sc))
    else:
        moved = be.apply_affine(A_sc, mov)                                          # :714
    if world > 1 and transform == 'Affine' and mov.shape[1] >= opts.icp_shard_min_points:
        try:
            A_icp, res = icp_sharded(be, moved, fix, int(icp_iterations), group)
            if details is not None:
                details['residuals'] = res.cpu().numpy()
            return A_icp
        except PlanarCloud:
            pass                     # (every rank arrives here together) -> the replicated loop below, which has the host's pinv fits
    log = {} if details is not None else None
    if opts.icp_one_launch is None:
        A_icp = be.icp(moved, fix, int(icp_iterations), transform, log)             # :715-717
    else:
        A_icp = be.icp(moved, fix, int(icp_iterations), transform, log, one_launch=opts.icp_one_launch)
    if details is not None:
        details.update(residuals=log['residuals'], nn=log['nn'])
    return A_icp


def estimate_transform(moving, fixed, *, transform='Affine', mode='unsupervised', ransac_samples=4, ransac_trials=8000,
                       ransac_error=16, icp_iterations=50, keypoints=None, seed=None, details=None, cost_mode='auto', group=None,
                       options=None):
    """Reproduces _dock_widget.py:526-718 -> (A_sc, A_icp, inliers[8]); final transform = A_icp @ A_sc (:428).

    moving, fixed   3 x N / 3 x M (rows z, y, x; a 4th row is dropped), NumPy or torch; float64 is the reference's arithmetic —
                    other dtypes are widened on the way in
    transform       'Affine' | 'Similar'
    mode            'unsupervised' (shape context + Hungarian + RANSAC) or 'supervised'
                    (keypoints=(kp_moving, kp_fixed), 3 x k each; _dock_widget.py:707-711)
    ransac_samples, ransac_trials, ransac_error, icp_iterations   the widget's spin boxes.  ransac_error: 16 for CSV detections
                    (_dock_widget.py:613-614); with nucleus sizes the widget uses 0.5 * (mean(size_m)**(1/3) + mean(size_f)**(1/3))
    seed            if not None, np.random.seed(seed) right before the eight RANSAC runs (the reference's seeded result, bit for bit)
    details         optional dict filled with intermediate results: lsa, ransac_A, residuals, nn, assignment (cost_mode started
                    from, route and counters per hypothesis), edge_guard (how many neighbours lie so close to a ring radius
                    ("ring") or to a sector plane / polar cone ("sector") — or coincide with the queried point — that the
                    reference itself would bin them by the rounding noise of its linear algebra; all zero = the integer
                    histograms are the reference's by construction for this call); details={"timing": True} on
                    entry adds a wall-clock split (and stream synchronisations)
    cost_mode       how the eight assignments are obtained — ALWAYS the vectors scipy.optimize.linear_sum_assignment returns on
                    the exact cost matrices.  'auto' (default): without building those matrices where that can be proven
                    (resolve_cost_mode: float32 filter matrices from 8 192 points, relaxed float64 from 1 024, exact below; a
                    pairing that cannot be proven — ties, near-ties — has its exact matrices built); 'exact': the eight exact
                    matrices always; 'relaxed' / 'filter': start from that build
    group           torch.distributed process group to shard over (None = this GPU only); every rank passes the same clouds
                    and gets the same results
    options         pipeline.Options (or a dict of its fields): backend, sampler, private_rng, accept_near_ties,
                    stream_hypotheses, keep_cost_buffer, icp_shard_min_points, icp_one_launch
    """
    import time
    import torch
    opts = Options.of(options)
    be = opts.backend or GpuBackend()
    if opts.sampler not in ('auto', 'numpy', 'device'):
        raise ValueError("sampler must be 'auto', 'numpy' or 'device'")
    if cost_mode not in COST_MODES:
        raise ValueError("cost_mode must be one of %s" % (COST_MODES,))
    if mode not in ('unsupervised', 'supervised'):
        raise ValueError("mode must be 'unsupervised' or 'supervised'")
    mov, fix = be.cloud(moving), be.cloud(fixed)
    # a fresh process allocates its cost buffer while everything else of a first call happens (_EarlyLease)
    early = _start_early_lease(be, mov, fix, cost_mode, opts, group) if mode == 'unsupervised' else None
    if opts.backend is None:
        from . import self_check
        try:
            self_check()                      # once per process: does this host's NumPy / BLAS round as the kernels restate it?
        except BaseException:
            if early is not None:
                early.cancel()
            raise
    inliers = np.zeros(8, dtype=np.int64)
    timing = {} if (details is not None and details.get("timing")) else None     # details={"timing": True}: wall-clock split
    clock = [time.perf_counter()]

    def mark(name):
        if timing is not None:
            if mov.is_cuda:
                torch.cuda.current_stream(mov.device).synchronize()
            timing[name] = timing.get(name, 0.0) + time.perf_counter() - clock[0]
        clock[0] = time.perf_counter()

    if mode == 'unsupervised':
        on_device = (opts.sampler == 'device' or (opts.sampler == 'auto' and seed is None)) and getattr(be, "device_sampler", False) \
            and int(ransac_samples) <= min(mov.shape[1], fix.shape[1])
        # what do_ransac draws depends only on the number of matched pairs: start drawing before the GPU has built anything
        # (device sampler: nothing to draw ahead — each trial's set is drawn in front of its fit)
        draws = _SampleDraws(be, min(mov.shape[1], fix.shape[1]), int(ransac_samples), 0 if on_device else int(ransac_trials),
                             seed, opts.private_rng)
        a_info = None if details is None else details.setdefault("assignment", {})
        try:
            lsa = _correspondences(be, mov, fix, moving, fixed, cost_mode, opts, group, a_info, details, mark, early)
        except BaseException:
            draws.thread.join()
            raise
        mark("host_assignment")
        sets = draws.result()                    # every rank draws the same 8 x trials: same RNG stream everywhere
        mark("host_draws_exposed")
        if timing is not None:
            timing["host_draws_thread"] = draws.seconds
        A_sc, inliers = _ransac_stage(be, mov, fix, lsa, sets, on_device, transform, ransac_samples, ransac_trials, ransac_error, seed,
                                      group, details)
        mark("gpu_ransac")
    elif mode == 'supervised':
        if keypoints is None:
            raise ValueError("supervised mode needs keypoints=(moving_keypoints, fixed_keypoints)")
        A_sc = be.fit(keypoints[0], keypoints[1], transform)
    else:
        raise ValueError("mode must be 'unsupervised' or 'supervised'")
    A_icp = _icp_stage(be, mov, fix, moving, A_sc, transform, icp_iterations, opts, group, details)
    mark("gpu_icp")
    if timing is not None:
        details["timing"] = timing
    if nat.is_torch(moving):
        return A_sc, nat.to_dev(A_icp, dev=mov.device), inliers
    return A_sc.cpu().numpy(), (A_icp.cpu().numpy() if nat.is_torch(A_icp) else np.asarray(A_icp)), inliers


# (several independent registrations: batch.py — imported at the end of this module, it needs estimate_transform)
from .batch import batch_assignment, batch_costs, estimate_transform_batch, _run_local  # noqa: E402,F401
