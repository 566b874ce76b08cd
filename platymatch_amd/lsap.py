"""Linear sum assignment on the host without the GIL (pm_lsap_solve, platymatch_amd/csrc/pm_lsap.cpp).

The widget calls scipy.optimize.linear_sum_assignment eight times in a row (platymatch/_dock_widget.py:604-611); SciPy
keeps the GIL, so those solves cannot overlap.  pm_lsap_solve restates SciPy 1.15.3's solver operation for operation
(identical indices, ties included — tests/test_lsap.py checks it against SciPy itself) and is called through ctypes,
which releases the GIL: `solve_many` runs the eight solves on eight host threads.  Host code only; nothing here runs on
the GPU."""
import ctypes
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as nat
from .device_memory import big_empty


def linear_sum_assignment(cost_matrix):
    """Same contract as scipy.optimize.linear_sum_assignment(cost) (minimisation): -> (row_ind, col_ind) int64,
    rows ascending; ValueError for NaN / -inf entries or an infeasible matrix."""
    cost = np.asarray(cost_matrix)
    if cost.ndim != 2:
        raise ValueError("expected a matrix (2-D array), got a %r array" % (cost.shape,))
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    k = min(cost.shape)
    rows = np.empty(k, dtype=np.int64)
    cols = np.empty(k, dtype=np.int64)
    rc = nat.load().pm_lsap_solve(cost.ctypes.data, cost.shape[0], cost.shape[1], rows.ctypes.data, cols.ctypes.data)
    if rc == -1:
        raise ValueError("matrix contains invalid numeric entries")
    if rc == -4:
        raise ValueError("cost matrix is infeasible")
    nat.check(rc)
    return rows, cols


def solve_many(cost_matrices, threads=None):
    """linear_sum_assignment for each matrix, concurrently (the foreign call releases the GIL) -> list of (rows, cols).
    An entry may be a zero-argument callable producing the matrix (e.g. a device-to-host copy): it is called on the
    worker thread, so fetching one matrix overlaps with solving another."""
    mats = list(cost_matrices)
    if threads is None:
        threads = min(len(mats), os.cpu_count() or 1, 8)

    def one(m):
        return linear_sum_assignment(m() if callable(m) else m)

    if threads <= 1 or len(mats) <= 1:
        return [one(m) for m in mats]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(one, mats))


# ---- the same solve with the matrix resident on the GPU ---------------------------------------------------------------
# include/platymatch_hip.h ("assignment with the matrix resident on the device") explains the scheme; csrc/pm_lsap_core.cpp
# is the sparse host solver, csrc/pm_lsap_dev.hip the two kernels that read the dense matrix.
CORE_EDGES_PER_ROW = 16          # initial core: up to this many cheap entries per row (measured at 19.5k: 16 -> 0.23/0.53 s per right/wrong hypothesis, 48 -> 0.36/0.79, 96 -> 0.67/1.42; pricing rounds unchanged)
PRICE_EDGES_PER_ROW = 8          # offenders a row may hand back per pricing round
MAX_PRICING_ROUNDS = 200
LISTING_FROM_FRACTION = 64       # a filtered solve leaves the selection rounds for the listing rounds once fewer than 1 row in this many is violated
REL_DELTA = 1e-13                # dual feasibility / tightness tolerance, relative to the largest dual or core cost
REL_EPS_COLLECT = 1e-7           # entries with reduced cost below this (relative) are collected by the certificate kernel
REL_EPS_FLOOR = 1e-11            # smallest uniqueness margin accepted (relative): ~1e5 x the rounding of one float64 operation
EPS_SAFETY = 16.0                # margin >= EPS_SAFETY * (sum of the rows' matched slack + sum of the rows' worst negative reduced cost)
COLUMN_REDUCTION = True          # start square solves from the column reduction
DEVICE_MIN_ROWS = 1024           # below this the dense host solver is quicker than the round trips of the device scheme
# The eight matrices hold four distinct sets of terms: U11/U22, U12/U21, U13/U24, U14/U23 differ only in
# summation order (<= 6e-16 per entry), so one solve serves both — the twin is CERTIFIED on its own entries, not assumed.
from ._pairings import PAIRINGS, TWINS  # noqa: E402


class _Core:
    def __init__(self, nr, nc):
        self.lib = nat.load()
        self.h = self.lib.pm_lsap_core_create(nr, nc)
        if not self.h:
            raise MemoryError("pm_lsap_core_create failed")
        self.nr, self.nc = nr, nc

    def close(self):
        if self.h:
            self.lib.pm_lsap_core_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add(self, cols, costs):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        costs = np.ascontiguousarray(costs, dtype=np.float64)
        nat.check(self.lib.pm_lsap_core_add(self.h, cols.shape[1], cols.ctypes.data, costs.ctypes.data))

    def init_state(self, u, v, col4row):
        u, v = np.ascontiguousarray(u, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
        c = np.ascontiguousarray(col4row, dtype=np.int32)
        nat.check(self.lib.pm_lsap_core_init_state(self.h, u.ctypes.data, v.ctypes.data, c.ctypes.data))

    def init_duals(self, u, v, argmin_col):
        u, v = np.ascontiguousarray(u, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
        a = np.ascontiguousarray(argmin_col, dtype=np.int32)
        nat.check(self.lib.pm_lsap_core_init_duals(self.h, u.ctypes.data, v.ctypes.data, a.ctypes.data))

    def auction(self, eps0, eps_min, factor, max_bids=0):
        import ctypes
        bids = ctypes.c_long(0)
        nat.check(self.lib.pm_lsap_core_auction(self.h, float(eps0), float(eps_min), float(factor), int(max_bids), ctypes.byref(bids)))
        return bids.value

    def auction_resume(self, price, assigned, eps0, eps_min, factor, max_bids=0):
        import ctypes
        price = np.ascontiguousarray(price, dtype=np.float64)
        assigned = np.ascontiguousarray(assigned, dtype=np.int32)
        if price.size != self.nc or assigned.size != self.nr:
            raise ValueError("price [nc] and assigned [nr] expected")
        bids = ctypes.c_long(0)
        nat.check(self.lib.pm_lsap_core_auction_resume(self.h, price.ctypes.data, assigned.ctypes.data, float(eps0), float(eps_min),
                                                       float(factor), int(max_bids), ctypes.byref(bids)))
        return bids.value

    def solve(self):
        rc = self.lib.pm_lsap_core_solve(self.h)
        if rc == -4:
            raise ValueError("cost matrix is infeasible")
        nat.check(rc)

    def reprice(self, cols, costs, delta):
        import ctypes
        n = ctypes.c_int(0)
        nat.check(self.lib.pm_lsap_core_reprice(self.h, cols.shape[1], cols.ctypes.data, costs.ctypes.data, float(delta), ctypes.byref(n)))
        return n.value

    def column_repairs(self):
        """Pricing rounds this core settled from the column side (pm_lsap_core_column_repairs)."""
        return int(self.lib.pm_lsap_core_column_repairs(self.h))

    def get(self):
        import ctypes
        u, v = np.empty(self.nr), np.empty(self.nc)
        c4r = np.empty(self.nr, dtype=np.int32)
        stats = (ctypes.c_long * 4)()
        nat.check(self.lib.pm_lsap_core_get(self.h, u.ctypes.data, v.ctypes.data, c4r.ctypes.data, stats))
        return u, v, c4r, list(stats)


def transposed(U):
    """A contiguous transpose of a float64 GPU matrix by the tiled kernel (pm_transpose_f64; torch's strided copy takes
    ~0.4 s for the 19 GB of a 50 000 x 47 000 matrix, the kernel ~15 ms)."""
    torch = nat.torch_mod()
    with torch.cuda.device(U.device):        # the launch goes to a stream of the device that owns U, whatever the current device is
        out = big_empty((U.shape[1], U.shape[0]), torch.float64, U.device)
        nat.check(nat.load().pm_transpose_f64(nat.ptr(U), U.shape[0], U.shape[1], U.stride(0), nat.ptr(out), out.stride(0), nat.stream_ptr(U)))
    return out


class DeviceMatrix:
    """The dense matrix as the solver sees it: three queries, each one streaming pass of a HIP kernel over HBM.
    (tests/test_lsap_core.py substitutes a NumPy double to exercise the host solver without a GPU.)"""

    def __init__(self, U):
        self.U = U
        self.shape = tuple(U.shape)
        self.ld = U.stride(0) if U.shape[0] > 1 else U.shape[1]      # a one-row tensor's row stride is arbitrary
        # a float32 matrix (a filter matrix, FilteredMatrix below) goes through the float32 forms of the three streaming passes
        self.f32 = str(U.dtype) == "torch.float32"

    def row_select_t(self, v, k):
        """pm_lsap_row_select with everything left on the device -> (cols [nr, k] int32, costs [nr, k] float64, nonfinite flag [1]
        int32) GPU tensors; v: None, a host array or a GPU tensor [nc]."""
        torch = nat.torch_mod()
        U = self.U
        nr, nc = U.shape
        cols = torch.empty((nr, k), dtype=torch.int32, device=U.device)
        costs = torch.empty((nr, k), dtype=torch.float64, device=U.device)
        flag = torch.empty(1, dtype=torch.int32, device=U.device)
        v_dev = None if v is None else nat.to_dev(v, dev=U.device)
        if v_dev is not None and v_dev.numel() != nc:
            raise ValueError("v must hold one dual per column")
        lib = nat.load()
        nat.check((lib.pm_lsap_row_select_f32 if self.f32 else lib.pm_lsap_row_select)(nat.ptr(U), nr, nc, self.ld, nat.ptr(v_dev), k, nat.ptr(cols),
                                                                                       nat.ptr(costs), nat.ptr(flag), nat.stream_ptr(U)))
        return cols, costs, flag

    def row_select(self, v, k):
        """-> (cols [nr, k] int32, costs [nr, k] float64, nonfinite flag): pm_lsap_row_select."""
        cols, costs, flag = self.row_select_t(v, k)
        return cols.cpu().numpy(), costs.cpu().numpy(), int(flag.item())

    def diagonal(self, n):
        return self.block_diagonal(0)[:n]

    def block_diagonal(self, row0):
        """U[r, row0 + r] for the rows of this block (the global matrix's diagonal when the block starts at row row0)."""
        torch = nat.torch_mod()
        nr, nc = self.U.shape
        r = torch.arange(max(0, min(nr, nc - row0)), device=self.U.device)
        return self.U[r, r + row0].cpu().numpy()

    def bid(self, v, rows):
        """Bids of the listed rows against prices v -> (j1 int32, u1, u2) per row: pm_lsap_bid."""
        torch = nat.torch_mod()
        U = self.U
        nr, nc = U.shape
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        if rows.size == 0:
            return np.zeros(0, np.int32), np.zeros(0), np.zeros(0)
        if rows.min() < 0 or rows.max() >= nr:
            raise IndexError("row outside the matrix")
        r_d = nat.to_dev(rows, dtype=torch.int32, dev=U.device)
        v_d = nat.to_dev(v, dev=U.device)
        j1 = torch.empty(rows.size, dtype=torch.int32, device=U.device)
        u12 = torch.empty((2, rows.size), dtype=torch.float64, device=U.device)
        nat.check(nat.load().pm_lsap_bid(nat.ptr(U), nr, nc, self.ld, nat.ptr(v_d), nat.ptr(r_d), int(rows.size), nat.ptr(j1),
                                         nat.ptr(u12[0]), nat.ptr(u12[1]), nat.stream_ptr(U)))
        u12_h = u12.cpu().numpy()
        return j1.cpu().numpy(), u12_h[0], u12_h[1]

    def entries(self, rows, cols):
        """U[rows[k], cols[k]] on the host (exact costs of given pairs)."""
        torch = nat.torch_mod()
        r = torch.as_tensor(np.asarray(rows, dtype=np.int64), device=self.U.device)
        c = torch.as_tensor(np.asarray(cols, dtype=np.int64), device=self.U.device)
        return self.U[r, c].cpu().numpy()

    def col_min(self):
        """-> v [nc] on the host: pm_lsap_col_min."""
        torch = nat.torch_mod()
        U = self.U
        nr, nc = U.shape
        lib = nat.load()
        ws = nat.workspace(lib.pm_lsap_col_min_workspace(nr, nc), U.device)
        v = torch.empty(nc, dtype=torch.float64, device=U.device)
        nat.check((lib.pm_lsap_col_min_f32 if self.f32 else lib.pm_lsap_col_min)(nat.ptr(U), nr, nc, self.ld, nat.ptr(v), nat.ptr(ws), ws.numel(),
                                                                                 nat.stream_ptr(U)))
        return v.cpu().numpy()

    def certificate(self, u, v, col4row, delta, eps, cap):
        """-> (violations, loose matched entries, tight edges [t, 2] int32 and their reduced costs [t] — None if more than
        cap —, sum over rows of (|reduced cost| of the matched entry + the row's worst negative reduced cost))."""
        viol, loose, tight, red, bound = self.certificate_t(u, v, col4row, delta, eps, cap)
        if tight is None:
            return viol, loose, None, None, bound
        return viol, loose, tight.cpu().numpy(), red.cpu().numpy(), bound

    def certificate_t(self, u, v, col4row, delta, eps, cap):
        """certificate with the list left on the device: tight [t, 2] int32 and red [t] float64 GPU tensors (or None, None)."""
        torch = nat.torch_mod()
        U = self.U
        nr, nc = U.shape
        u_d, v_d = nat.to_dev(u, dev=U.device), nat.to_dev(v, dev=U.device)
        c_d = nat.to_dev(col4row, dtype=torch.int32, dev=U.device)
        summary = torch.empty(4, dtype=torch.int32, device=U.device)
        stats = torch.empty(2, dtype=torch.float64, device=U.device)
        tight = torch.empty((cap, 2), dtype=torch.int32, device=U.device)
        red = torch.empty(cap, dtype=torch.float64, device=U.device)
        rows = torch.empty((2, nr), dtype=torch.float64, device=U.device)
        lib = nat.load()
        nat.check((lib.pm_lsap_certificate_f32 if self.f32 else lib.pm_lsap_certificate)(
            nat.ptr(U), nr, nc, self.ld, nat.ptr(u_d), nat.ptr(v_d), nat.ptr(c_d), float(delta), float(eps), nat.ptr(summary), nat.ptr(stats),
            nat.ptr(tight), nat.ptr(red), cap, nat.ptr(rows[0]), nat.ptr(rows[1]), nat.stream_ptr(U)))
        viol, n_tight, loose, _ = summary.cpu().tolist()
        rows_h = rows.cpu().numpy()
        bound = float(rows_h[0].sum() + rows_h[1].sum())         # what the observed imperfections can cost any alternative, in total
        if n_tight > cap:
            return viol, loose, None, None, bound
        return viol, loose, tight[:n_tight], red[:n_tight], bound


class FilteredMatrix:
    """An APPROXIMATE dense matrix as a filter in front of exact evaluation: `approx` (a DeviceMatrix, or the
    tests' NumPy double) holds a matrix whose every entry is within cost_delta of the exact one, which is never built;
    exact_entries(rows, cols) -> tuple of float64 arrays answers listed entries of the exact matrix (first array: the hypothesis
    this matrix is solved for; further arrays: its twin(s), used by certify_listed).  The solver's queries all ask WHICH entries
    have a small reduced cost: the approximate matrix selects them, every cost that reaches the sparse core or a verdict is exact.
      row_select        the k entries the approximate matrix ranks first, with their exact costs, re-sorted by exact reduced cost
      diagonal/entries  exact
      col_min           approximate (a starting price, nothing rests on it)
      threshold_select  ALL non-matched entries whose approximate reduced cost is below eps_collect + 2 cost_delta (so: every entry
                        whose exact reduced cost is below eps_collect), per row, with exact costs — what solve_core's last pricing
                        rounds and the certificate work on
      certificate       the listing pass of certify_listed (negative reduced costs are listed, not counted as violations)."""

    LIST_CAPACITY_PER_COLUMN = 64

    def __init__(self, approx, exact_entries, cost_delta, exact_entries_t=None, entries_device=None):
        """exact_entries_t (optional): the same question asked and answered with GPU tensors (rows, cols int32 [t] -> tuple of
        float64 [t]); selections and the per-row lists of threshold_select are then finished on the device — the selector's own
        (a DeviceMatrix as `approx`) or, for a selector that answers on the host (ShardedMatrix), entries_device."""
        self.A = approx
        self.entries_device = entries_device
        self.shape = tuple(approx.shape)
        self.exact_entries = exact_entries
        self.exact_entries_t = exact_entries_t
        self.cost_delta = float(cost_delta)
        self.exact_evaluated = 0
        self._listed = None             # the last complete listing: (u, v, col4row, eps, tight, red)

    def _exact(self, rows, cols):
        self.exact_evaluated += int(len(rows))
        return np.asarray(self.exact_entries(np.ascontiguousarray(rows, dtype=np.int32), np.ascontiguousarray(cols, dtype=np.int32))[0],
                          dtype=np.float64)

    def col_min(self):
        return self.A.col_min()

    def diagonal(self, n):
        i = np.arange(n, dtype=np.int32)
        return self._exact(i, i)

    def entries(self, rows, cols):
        return self._exact(rows, cols)

    def _row_select_t(self, v, k, with_diagonal=False):
        """row_select with selection, exact evaluation and re-sorting on the device: one read-back.  with_diagonal: the exact
        entries (i, i) ride along in the same evaluation launch and come back as a fourth value (solve_core's safety edges: one
        round trip less per hypothesis — with four pairings sharing the GPU a tiny launch of its own waited 1-34 ms in the queue)."""
        torch = nat.torch_mod()
        if hasattr(self.A, "row_select_t"):
            cols, _, flag = self.A.row_select_t(v, k)
        else:
            # a selector that answers on the host (lsap_sharded.ShardedMatrix: the ranks' candidate lists, gathered): the lists go up
            # to the device the exact entries are evaluated on (entries_device) and are finished there like a local selection's
            cols_h0, _, bad0 = self.A.row_select(v, k)
            cols = torch.as_tensor(np.ascontiguousarray(cols_h0, dtype=np.int32), device=self.entries_device)
            flag = torch.tensor([int(bad0)], dtype=torch.int32, device=self.entries_device)
        valid = cols >= 0
        rows = torch.arange(cols.shape[0], dtype=torch.int32, device=cols.device)[:, None].expand_as(cols)
        r_list, c_list = rows[valid].contiguous(), cols[valid].contiguous()
        n_diag = 0
        if with_diagonal:
            n_diag = min(self.shape)
            d = torch.arange(n_diag, dtype=torch.int32, device=cols.device)
            r_list, c_list = torch.cat([r_list, d]), torch.cat([c_list, d])
        exact = self.exact_entries_t(r_list, c_list)[0]
        self.exact_evaluated += int(exact.numel())
        diag = None
        if with_diagonal:
            diag, exact = exact[exact.numel() - n_diag:], exact[:exact.numel() - n_diag]
        costs = torch.full(cols.shape, float("inf"), dtype=torch.float64, device=cols.device)
        costs[valid] = exact
        red = costs
        if v is not None:
            v_d = nat.to_dev(v, dev=cols.device)
            red = costs - v_d[cols.clamp(min=0).long()]
        order = torch.argsort(torch.where(valid, red, torch.full_like(red, float("inf"))), dim=1, stable=True)
        cols_s, costs_s = torch.gather(cols, 1, order), torch.gather(costs, 1, order)
        bad = torch.stack([flag.reshape(-1)[0].to(torch.int32), (~torch.isfinite(exact)).any().to(torch.int32)])
        cols_h, costs_h, bad_h = cols_s.cpu().numpy(), costs_s.cpu().numpy(), bad.cpu().numpy()
        if with_diagonal:
            return cols_h, costs_h, int(bad_h.max()), diag.cpu().numpy()
        return cols_h, costs_h, int(bad_h.max())

    def _device_finish(self):
        """Can selections be finished on a device?  Yes with a tensor-valued entry function and either a selector that leaves its
        lists on that device (DeviceMatrix) or a stated entries_device for a host-side selector (the sharded route's root)."""
        return self.exact_entries_t is not None and (hasattr(self.A, "row_select_t") or self.entries_device is not None)

    def row_select_with_diagonal(self, v, k):
        """row_select(v, k) + diagonal(min(shape)) in one evaluation launch where the device path exists, else None."""
        if self._device_finish():
            return self._row_select_t(v, k, with_diagonal=True)
        return None

    def row_select(self, v, k):
        if self._device_finish():
            return self._row_select_t(v, k)
        cols, costs, bad = self.A.row_select(v, k)
        if bad:
            return cols, costs, bad
        valid = cols >= 0
        rows = np.broadcast_to(np.arange(cols.shape[0], dtype=np.int32)[:, None], cols.shape)
        exact = self._exact(rows[valid], cols[valid])
        if not np.isfinite(exact).all():
            return cols, costs, 1
        costs = np.where(valid, 0.0, np.inf)
        costs[valid] = exact
        red = costs - (0.0 if v is None else np.asarray(v)[np.maximum(cols, 0)])
        order = np.argsort(np.where(valid, red, np.inf), axis=1, kind="stable")
        return np.take_along_axis(cols, order, axis=1), np.take_along_axis(costs, order, axis=1), 0

    def certificate(self, u, v, col4row, delta, eps, cap):
        L = self._listed
        if (L is not None and eps <= L[3] and np.array_equal(L[0], u) and np.array_equal(L[1], v) and np.array_equal(L[2], col4row)):
            keep = L[5] <= eps                               # the same duals were listed (with a margin at least as wide): reuse
            return 0, 0, L[4][keep], L[5][keep], 0.0
        cap = max(int(cap), self.LIST_CAPACITY_PER_COLUMN * self.shape[1])
        viol, _, tight, red, _ = self.A.certificate(u, v, col4row, float("inf"), eps, cap)      # (delta = inf: nothing is set aside as a violation)
        if viol:
            tight = None                                     # (cannot happen with delta = inf; an unlisted entry must never pass for feasible)
        if tight is not None:
            self._listed = (np.array(u, copy=True), np.array(v, copy=True), np.array(col4row, copy=True), float(eps), tight, red)
        return 0, 0, tight, red, 0.0

    def _tau(self, u, v):
        scale = max(float(np.abs(u).max()), float(np.abs(v).max()), 1e-300)
        return REL_EPS_COLLECT * scale + 2.0 * self.cost_delta + 2.0 * REL_DELTA * scale

    def _threshold_select_t(self, u, v, col4row):
        """threshold_select with the list sorted, evaluated and laid out per row on the device."""
        torch = nat.torch_mod()
        nr, nc = self.shape
        tau = self._tau(u, v)
        if hasattr(self.A, "certificate_t"):
            viol, _, tight, red, _ = self.A.certificate_t(u, v, col4row, float("inf"), tau, self.LIST_CAPACITY_PER_COLUMN * nc)
            if tight is None or viol:
                return None
            tight_h, red_h = tight.cpu().numpy(), red.cpu().numpy()
        else:                                    # a host-side selector: its list goes up to the entries' device
            viol, _, tight_h, red_h, _ = self.A.certificate(u, v, col4row, float("inf"), tau, self.LIST_CAPACITY_PER_COLUMN * nc)
            if tight_h is None or viol:
                return None
            tight = torch.as_tensor(np.ascontiguousarray(tight_h, dtype=np.int32), device=self.entries_device)
        self._listed = (np.array(u, copy=True), np.array(v, copy=True), np.array(col4row, copy=True), tau, tight_h, red_h)
        if tight.shape[0] == 0:
            return np.full((nr, 1), -1, dtype=np.int32), np.full((nr, 1), np.inf)
        r = tight[:, 0].long()
        order = torch.argsort(r, stable=True)
        r, c = r[order], tight[order, 1].contiguous()
        exact = self.exact_entries_t(r.to(torch.int32), c)[0]
        self.exact_evaluated += int(r.numel())
        counts = torch.bincount(r, minlength=nr)
        start = torch.cumsum(counts, 0) - counts
        slot = torch.arange(r.numel(), device=r.device) - start[r]
        kmax = int(counts.max())
        cols = torch.full((nr, kmax), -1, dtype=torch.int32, device=r.device)
        costs = torch.full((nr, kmax), float("inf"), dtype=torch.float64, device=r.device)
        cols[r, slot], costs[r, slot] = c, exact
        return cols.cpu().numpy(), costs.cpu().numpy()

    def threshold_select(self, u, v, col4row):
        """-> (cols [nr, kmax] int32, -1 padded; exact costs [nr, kmax], inf padded) of every non-matched entry whose EXACT reduced
        cost can be below REL_EPS_COLLECT x scale, or None if the list overflowed."""
        if self.exact_entries_t is not None and (hasattr(self.A, "certificate_t") or self.entries_device is not None):
            return self._threshold_select_t(u, v, col4row)
        nr = self.shape[0]
        tau = self._tau(u, v)
        _, _, tight, _, _ = self.certificate(u, v, col4row, float("inf"), tau, 0)
        if tight is None:
            return None
        if len(tight) == 0:
            return np.full((nr, 1), -1, dtype=np.int32), np.full((nr, 1), np.inf)
        order = np.argsort(tight[:, 0], kind="stable")
        r, c = tight[order, 0].astype(np.int64), tight[order, 1].astype(np.int32)
        exact = self._exact(r, c)
        counts = np.bincount(r, minlength=nr)
        start = np.concatenate([[0], np.cumsum(counts)[:-1]])
        slot = np.arange(len(r)) - start[r]
        kmax = int(counts.max())
        cols = np.full((nr, kmax), -1, dtype=np.int32)
        costs = np.full((nr, kmax), np.inf)
        cols[r, slot], costs[r, slot] = c, exact
        return cols, costs


_WARM = {"lock": threading.Lock(), "done": set(), "threads": {}}


def warm_up(device, wait=True):
    """First use of the device-finished selection and listing code (FilteredMatrix._row_select_t / _threshold_select_t) on `device`,
    on a 64 x 64 toy matrix and a stream of its own — once per process and device.  Why: torch loads the code objects of its own
    kernels (stable sorting, gathering, masked indexing, bincount, cumsum ...) at their first launch — 0.3-0.7 s in a fresh process
    (tools/cold_start.py --torch-warm), which the first registration of a process used to pay inside its assignment stage, with the
    GPU idle.  It cannot be hidden INSIDE a registration — measured both ways (profiles/r05_cold_start.txt): started on a thread at
    the entry of the first call the load only gets in the way of the first launches (the runtime loads one code object at a time:
    2.5 s instead of ~1.9 on the same box), run while the filter build is in flight it waits for the build (a code object's device
    memory is allocated with the device idle) and then costs the same.  So it is paid where the caller says: platymatch_amd.reserve()
    and platymatch_amd.warm_up() run it AHEAD of the first registration (first 50 000-nucleus registration 1.40 -> 1.08-1.12 s).
    wait=False: start the thread and return at once.  Nothing of a registration depends on it; a failure in here is dropped."""
    torch = nat.torch_mod()
    dev = nat.device(device)
    if dev.type != "cuda":
        return
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    dev = torch.device("cuda", key)

    def work():
        try:
            with torch.cuda.device(dev), torch.cuda.stream(nat.side_stream(dev, ("warm-up",))):
                n = 64
                g = torch.arange(n * n, dtype=torch.float64, device=dev)
                U = torch.remainder(g * 0.6180339887498949, 1.0).view(n, n).contiguous()      # (no generator is touched)
                M = FilteredMatrix(DeviceMatrix(U.to(torch.float32)), lambda r, c: (U[torch.as_tensor(r, device=dev).long(), torch.as_tensor(c, device=dev).long()].cpu().numpy(),),
                                   1e-6, exact_entries_t=lambda r, c: (U[r.long(), c.long()],))
                v = np.zeros(n)
                _, costs, _, _ = M._row_select_t(v, 8, with_diagonal=True)
                M._row_select_t(None, 8)
                M._threshold_select_t(costs[:, 0].copy(), v, np.arange(n, dtype=np.int32))
                M.col_min()
                torch.cuda.current_stream(dev).synchronize()
        except Exception:        # noqa: BLE001 — a warm-up proves nothing and must cost nothing
            pass

    with _WARM["lock"]:
        th = _WARM["threads"].get(key)
        if key not in _WARM["done"]:
            _WARM["done"].add(key)
            th = _WARM["threads"][key] = threading.Thread(target=work, name="pm-warm-up")
            th.start()
    if wait and th is not None:
        th.join()
        with _WARM["lock"]:
            _WARM["threads"].pop(key, None)


def certify(M, u, v, col4row, info=None, min_eps=0.0):
    """Is (u, v, col4row) a certified UNIQUE optimum of the matrix M (nr <= nc)?  Dual feasibility and complementary
    slackness on every entry (pm_lsap_certificate), the free columns carrying the largest column dual (nr < nc), and no
    alternating cycle among the entries within eps of tight (pm_lsap_unique).  info["optimal"] tells the two failures apart:
    True = the assignment IS optimal (to the rounding bound) but another one lies within the margin, so which of them
    SciPy's rounding would return cannot be told without running SciPy's algorithm.
    min_eps (absolute): a wider uniqueness margin demanded by the caller — the relaxed cost build (pm_chi2_cost8_relaxed) asks for
    2 min(N, M) delta, delta its per-entry error bound: an optimum of the relaxed matrix that beats every alternative by more
    than that is the exact matrix's unique optimum too."""
    if _native_ok(M):
        return certify_native(M, u, v, col4row, info, min_eps)
    lib = nat.load()
    nr, nc = M.shape
    scale = max(float(np.abs(u).max()), float(np.abs(v).max()), 1e-300)
    delta, eps_collect = REL_DELTA * scale, max(REL_EPS_COLLECT * scale, 4.0 * float(min_eps))
    cap = 8 * nc + 1024
    viol, loose, tight, red, bound = M.certificate(u, v, col4row, delta, eps_collect, cap)
    # How wide must the margin be?  Any other assignment costs at least (the reduced costs of its new entries) - bound more than
    # this one, bound = sum over rows of the matched entry's |reduced cost| and of the row's most negative reduced cost (both
    # ~1e-17 per row: rounding).  If every alternative needs an entry with reduced cost > eps, eps > bound separates them —
    # taken EPS_SAFETY times wider, and never below REL_EPS_FLOOR (far above the rounding any exact solver, SciPy's included,
    # accumulates along a cycle).
    eps = max(REL_EPS_FLOOR * scale, EPS_SAFETY * bound, float(min_eps) + bound)
    if info is not None:
        info.update(violations=viol, loose=loose, tight=None if tight is None else len(tight), slack_bound=bound, delta=delta, eps=eps)
    if info is not None:
        info["optimal"] = False            # set below once feasibility, tightness and (nr < nc) the free columns' prices check out
    if viol or loose:
        return False
    v_free = 0.0
    if nc > nr:
        free = np.ones(nc, dtype=bool)
        free[col4row] = False
        v_free = float(v[free].min())
        if float(v.max()) - v_free > delta:           # a matched column priced above a free one: not optimal for nr < nc
            return False
    if info is not None:
        info["optimal"] = True             # LP duality: no assignment is cheaper by more than `bound`; uniqueness is the open question
    if tight is None or eps > eps_collect:
        return False
    keep = red <= eps
    tight = tight[keep]
    if info is not None:
        info["tight_within_eps"] = int(keep.sum())
        info["_tight_edges"] = tight                   # (row, col) of the entries an alternative optimum would have to use: resolve_near_ties
    t = np.ascontiguousarray(tight, dtype=np.int32)
    c4r = np.ascontiguousarray(col4row, dtype=np.int32)
    vv = np.ascontiguousarray(v, dtype=np.float64)
    rc = lib.pm_lsap_unique(nr, nc, c4r.ctypes.data, vv.ctypes.data, v_free, eps, t.ctypes.data, int(len(t)))
    if rc < 0:
        nat.check(rc)
    if info is not None:
        info["unique"] = rc == 1
    return rc == 1


def certify_listed(M, u, v, col4row, exact_entries, cost_delta, infos=None):
    """(u, v, col4row) solved on a RELAXED matrix M (nr <= nc; every entry within cost_delta of the exact matrix C's, e.g.
    pm_chi2_cost8_relaxed's): is col4row the certified UNIQUE optimum of C — of each of the exact matrices exact_entries answers
    for — without C ever being built?
    exact_entries(rows, cols) -> tuple of float64 arrays: the listed entries of each exact matrix (hypothesis, twin).
    With the row duals retuned to the exact matched entries, u'_i = C[i][s(i)] - v[s(i)] (tight to the bit), an unlisted entry's
    exact reduced cost is at least its relaxed one minus 2 cost_delta.  So one pass over M (pm_lsap_certificate) that lists the
    entries with relaxed reduced cost <= eps_collect + 2 cost_delta leaves every OTHER entry feasible for C and further than
    eps_collect from tight; the matched and the listed entries (a few N) get their exact values, and feasibility, the slack
    bound, the free columns' prices and the uniqueness check (pm_lsap_unique) are decided on those exactly as lsap.certify
    decides them on an exact matrix — same margins, no 2 N delta anywhere.
    -> list of verdicts, one per exact matrix; infos (optional list of dicts) receive certify's keys."""
    lib = nat.load()
    nr, nc = M.shape
    u, v = np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64)
    c4r = np.ascontiguousarray(col4row, dtype=np.int32)
    scale = max(float(np.abs(u).max()), float(np.abs(v).max()), 1e-300)
    delta, eps_collect = REL_DELTA * scale, REL_EPS_COLLECT * scale
    widen = 2.0 * float(cost_delta)
    cap = 8 * nc + 1024
    viol, loose, tight, red, _ = M.certificate(u, v, c4r, delta, eps_collect + widen + delta, cap)
    n_out = None

    def verdicts(flag, **keys):
        out = [flag] * (n_out or (len(infos) if infos else 1))
        for i in (infos or []):
            if i is not None:
                i.update(optimal=False, **keys)
        return out

    if viol or loose or tight is None:
        return verdicts(False, violations=viol, loose=loose, tight=None if tight is None else len(tight))
    rows_all = np.concatenate([np.arange(nr, dtype=np.int32), tight[:, 0].astype(np.int32)])
    cols_all = np.concatenate([c4r, tight[:, 1].astype(np.int32)])
    exact = exact_entries(rows_all, cols_all)
    n_out = len(exact)
    v_free = 0.0
    free_ok = True
    if nc > nr:
        free = np.ones(nc, dtype=bool)
        free[c4r] = False
        v_free = float(v[free].min())
        free_ok = not (float(v.max()) - v_free > delta)     # a matched column priced above a free one: not optimal for nr < nc
    out = []
    for k, C in enumerate(exact):
        info = infos[k] if infos and k < len(infos) and infos[k] is not None else {}
        C = np.asarray(C, dtype=np.float64)
        info.update(listed=int(len(tight)), delta=delta, optimal=False)
        if C.shape != rows_all.shape or not np.isfinite(C).all():
            out.append(False)
            continue
        u2 = C[:nr] - v[c4r]                                  # the exact matched entries are tight to the bit
        # the premise |M - C| <= cost_delta, checked where C is known: u is tight on M's matched entries to within delta (`loose`
        # above; a FilteredMatrix's core holds exact costs, u2 = u to rounding), so |u2 - u| <= cost_delta + delta — and no more than
        # that may be granted: an unlisted entry's exact reduced cost is its approximate one (> eps_collect + 2 cost_delta + delta)
        # + (C - M) - (u2 - u), which stays above eps_collect only under this bound
        if float(np.abs(u2 - u).max()) > float(cost_delta) + delta:
            info["cost_delta_exceeded"] = float(np.abs(u2 - u).max())
            out.append(False)
            continue
        redC = (C[nr:] - v[tight[:, 1]]) - u2[tight[:, 0]]
        if len(redC) and float(np.abs(redC - red).max()) > widen + 2.0 * delta:      # the same premise on the listed entries: (C - M) - (u2 - u)
            info["cost_delta_exceeded"] = float(np.abs(redC - red).max())
            out.append(False)
            continue
        # Listed entries may come out (slightly) negative under the retuned duals — by at most 2 cost_delta.  As in certify, that
        # goes into the bound: another assignment costs (the reduced costs of its new entries) more than this one, its negative
        # entries can take back at most sum over rows of the row's worst one = bound, so an alternative that uses ANY entry
        # above eps > bound is dearer, and one that stays within eps of tight is an alternating cycle pm_lsap_unique would find.
        worst = np.zeros(nr)
        neg = redC < 0.0
        np.minimum.at(worst, tight[neg, 0], redC[neg])        # (only negative entries can lower a row's worst below 0)
        bound = float(-worst.sum())
        eps = max(REL_EPS_FLOOR * scale, EPS_SAFETY * bound)
        info.update(violations=int((redC < -delta).sum()), loose=0, slack_bound=bound, eps=eps)
        if not free_ok or eps > eps_collect:
            out.append(False)
            continue
        info["optimal"] = bound <= delta * nr                 # (to the rounding bound, as certify reports it; the verdict below does not need it)
        keep = redC <= eps
        t = np.ascontiguousarray(tight[keep], dtype=np.int32)
        info["tight_within_eps"] = int(keep.sum())
        info["_tight_edges"] = t
        rc = lib.pm_lsap_unique(nr, nc, c4r.ctypes.data, np.ascontiguousarray(v).ctypes.data, v_free, eps, t.ctypes.data, int(len(t)))
        if rc < 0:
            nat.check(rc)
        info["unique"] = rc == 1
        out.append(rc == 1)
    return out


RESOLVE_MAX_BLOCK_ROWS = 4096      # largest set of rows (all near-tied groups together) settled by the dense algorithm on their block


_F = -1        # the node "free columns / dummy rows" of the near-tie digraph (pm_lsap_unique's F)


def _cyclic_groups(edges, owner, releasable=()):
    """Rows that lie on an alternating cycle of the digraph i -> owner(col) over the near-tight entries (row, col): the strongly
    connected components with more than one node (iterative Tarjan over the nodes that appear at all).  With spare columns the
    digraph has one more node F (-1): i -> F for a near-tight entry into a column nobody holds, F -> i for every row in
    `releasable` (rows whose held column is priced like a free one: a dummy row could take it at no cost) — a cycle through F is
    an alternating PATH from a column that gets released to a column that gets taken.
    -> list of (sorted row array, passes through F)."""
    succ = {}
    for i, j in edges:
        o = int(owner[int(j)])
        if o >= 0 and o != int(i):
            succ.setdefault(int(i), []).append(o)
        elif o < 0 and len(releasable):
            succ.setdefault(int(i), []).append(_F)
    if len(releasable) and any(_F in nxt for nxt in succ.values()):
        succ[_F] = [int(i) for i in releasable]
    index, low, on, order, groups = {}, {}, set(), [], []
    for root in list(succ):
        if root in index:
            continue
        work = [(root, 0)]
        while work:
            node, k = work.pop()
            if k == 0:
                index[node] = low[node] = len(index)
                order.append(node)
                on.add(node)
            nxt = succ.get(node, [])
            if k < len(nxt):
                work.append((node, k + 1))
                w = nxt[k]
                if w not in index:
                    work.append((w, 0))
                elif w in on:
                    low[node] = min(low[node], index[w])
                continue
            for w in nxt:                                  # children are finished: fold their low links
                if w in on:
                    low[node] = min(low[node], low[w])
            if low[node] == index[node]:
                comp = []
                while True:
                    w = order.pop()
                    on.discard(w)
                    comp.append(w)
                    if w == node:
                        break
                if len(comp) > 1:
                    groups.append((np.array(sorted(x for x in comp if x != _F), dtype=np.int64), _F in comp))
    return groups


def resolve_near_ties(M, sol, info):
    """A certified OPTIMAL assignment whose uniqueness could not be proven (certify: info["optimal"], an alternative within eps):
    every alternative optimum differs from it only by alternating cycles over the near-tight entries, i.e. by a permutation of
    the columns WITHIN each group of rows such a cycle connects.  Each group's block (its rows x the columns they hold) is
    fetched — a few rows, whatever the size of the matrix — and assigned by SciPy's own algorithm (pm_lsap_solve), the result
    spliced in: the answer where the reference always answers (_dock_widget.py:604-611), optimal to the certificate's bound, with
    the near-tie settled by the reference's solver on the entries that decide it.  (For an EXACT tie inside a block SciPy's pick on
    the block need not be its pick on the whole matrix: its tie-breaking follows the order of its augmentations.)
    Spare columns (nr < nc; round 5): an alternative may also RELEASE a held column that is priced like a free one and TAKE a
    column nobody holds — an alternating path, a cycle through the node F of the digraph.  The rows on such cycles get a rectangular
    block: their rows x (the columns they hold + the free columns they are near-tight on).
    -> col4row, or None if it does not apply (no edges kept, groups beyond RESOLVE_MAX_BLOCK_ROWS rows in all, a block the dense
    solver refuses)."""
    import math
    nr, nc = M.shape
    edges = info.get("_tight_edges")
    if edges is None or len(edges) == 0 or not hasattr(M, "entries"):
        return None
    edges = np.asarray(edges)
    c4r = np.array(sol[2], dtype=np.int64, copy=True)
    owner = np.full(nc, -1, dtype=np.int64)
    owner[c4r] = np.arange(nr)
    releasable = ()
    if nr < nc:
        v = np.asarray(sol[1])
        v_free = float(v[owner < 0].min())
        releasable = np.flatnonzero(v_free - v[c4r] <= float(info.get("eps", 0.0)))      # rows whose column a dummy row could take
    groups = _cyclic_groups(edges, owner, releasable)
    total = int(sum(len(g) for g, _ in groups))
    if not groups or total > RESOLVE_MAX_BLOCK_ROWS:
        return None
    changed = 0
    for rows, through_free in groups:
        held = c4r[rows]
        cols = held
        if through_free:
            mine = np.isin(edges[:, 0], rows) & (owner[edges[:, 1]] < 0)
            cols = np.concatenate([held, np.unique(edges[mine, 1]).astype(np.int64)])
        k, kc = len(rows), len(cols)
        S = np.asarray(M.entries(np.repeat(rows, kc), np.tile(cols, k)), dtype=np.float64).reshape(k, kc)
        try:
            r, c = linear_sum_assignment(S)
        except ValueError:
            return None
        if math.fsum(S[r, c]) > math.fsum(S[np.arange(k), np.arange(k)]) + abs(info.get("eps", 0.0)) * k:      # cannot happen for a certified optimum
            return None
        changed += int((c != np.arange(k)).sum())
        c4r[rows[r]] = cols[c]
    if len(np.unique(c4r)) != nr:                          # (two groups sharing a free column: not settled block by block)
        return None
    info["resolved_groups"] = [int(len(g)) for g, _ in groups]
    info["resolved_through_spare_columns"] = int(sum(1 for _, f in groups if f))
    info["resolved_rows_moved"] = changed
    return c4r.astype(np.int32)


# Jacobi rounds of augmenting row reduction on the dense rows before the core is chosen (0 = off, the default).  Measured at
# 19 536 nuclei with 12 rounds: two thirds fewer augmentations, core solve -29 % (right hypothesis) / -8 % (wrong ones: the
# time is in the last few hundred rows' long searches, which a better start does not shorten) — and no wall-clock gain in the
# driver, where the extra passes and the host's conflict resolution cost what they save (batch of 64: 22.1 s vs 19.5 s).
ROW_REDUCTION_ROUNDS = 0

# eps-scaling auction over the sparse core before the first shortest-path solve: (eps0, eps_min) in units of the core's width,
# the scaling factor, how many auction + pricing rounds, the bid budget.  None = off.  Measured on chi-square matrices
# (round 2 probe, git history): eight assignments at 20 000 x 20 000 nuclei 0.56 s -> 0.13 s, at 50 000 x 50 000 2.65 s -> 0.65 s.
# max_free_columns: the largest share of spare columns (nc - nr) / nc for which the auction runs (1.0: always).  With spare
# columns, rows freed after the auction strand their columns below the dual a free column must carry; each of those is put
# right by a search from the column side (pm_lsap_core.cpp: reverse_augment), a few steps each.
AUCTION = None if os.environ.get("PM_LSAP_AUCTION") == "0" else dict(eps0=0.25, eps_min=1e-6, factor=5.0, rounds=3, later_eps0=0.01, bids_per_row=200, later_bids_per_row=60, stop_below=0.02, max_free_columns=1.0)


def _row_reduction(M, v, rounds):
    """Warm start on the DENSE matrix (Jonker & Volgenant's augmenting row reduction in its parallel, auction-like form): every
    free row bids for its cheapest column j1 at current prices; each column goes to the bid that lowers its price most —
    by u2 - u1, the margin over the bidder's second choice — and whoever held it is free again.  Prices only fall, so duals
    stay feasible; the winner's edge is tight (u = u2 = cost - new price).  A round is one pass of the bid kernel over the
    free rows (their dense rows: no core yet, nothing is missed) and a vectorised conflict resolution on the host.
    -> (u [nr] with NaN for free rows, v, col4row [nr] with -1 for free rows)."""
    nr, nc = M.shape
    v = np.array(v, dtype=np.float64, copy=True)
    u = np.full(nr, np.nan)
    col4row = np.full(nr, -1, dtype=np.int32)
    row4col = np.full(nc, -1, dtype=np.int32)
    free = np.arange(nr, dtype=np.int32)
    for _ in range(rounds):
        if free.size == 0:
            break
        j1, u1, u2 = M.bid(v, free)
        ok = (j1 >= 0) & np.isfinite(u1)
        if not ok.any():
            break
        rows_b, j1, u1, u2 = free[ok], j1[ok], u1[ok], u2[ok]
        inc = np.where(np.isfinite(u2), u2 - u1, 0.0)            # a one-column row bids without a margin
        order = np.lexsort((rows_b, -inc, j1))                    # per column: largest margin first, lowest row on ties
        first = np.ones(order.size, dtype=bool)
        first[1:] = j1[order][1:] != j1[order][:-1]
        win = order[first]
        wr, wc, winc, wu = rows_b[win], j1[win], inc[win], np.where(np.isfinite(u2[win]), u2[win], u1[win])
        if wr.size == 0:
            break
        old = row4col[wc]
        col4row[old[old >= 0]] = -1                               # displaced holders are free again
        row4col[wc] = wr
        col4row[wr] = wc
        u[wr] = wu
        v[wc] -= winc
        free = np.flatnonzero(col4row < 0).astype(np.int32)
        if not (winc > 0).any():                                  # only zero-margin moves left: rounds would ping-pong
            break
    u[col4row < 0] = np.nan
    return u, v, col4row


# ---- the native driver (csrc/pm_lsap_resident.hip): the same sequence as solve_core / certify below in one foreign call each --------
# Measured, round 4 (profiles/r04_lsap_phases.txt, r04_batch64_native*.json): the native driver is NOT faster — 11.8 / 14.0 ms against
# 9.1 / 15.5 ms per hypothesis at 5 000 nuclei, 79 / 87 against 73 / 84 at 20 000, the same for all eight on four threads and for
# the 64-pair batch: the glue it removes was ~1 ms per hypothesis, the time is in the host core's auction and shortest paths either
# way.  Kept as an option (PM_LSAP_NATIVE=1, lsap.NATIVE_DRIVER = True; identical answers: tests/test_gpu_lsap.py), not the default:
# the Python driver is the one every kind of matrix (resident, sharded, the CPU tests' double) goes through.
NATIVE_DRIVER = os.environ.get("PM_LSAP_NATIVE", "0") == "1"


class _Options(ctypes.Structure):
    _fields_ = [("core_edges", ctypes.c_int), ("price_edges", ctypes.c_int), ("max_pricing_rounds", ctypes.c_int), ("column_reduction", ctypes.c_int),
                ("rel_delta", ctypes.c_double), ("rel_eps_collect", ctypes.c_double), ("rel_eps_floor", ctypes.c_double), ("eps_safety", ctypes.c_double),
                ("auction", ctypes.c_int), ("a_rounds", ctypes.c_int), ("a_bids_per_row", ctypes.c_int), ("a_later_bids_per_row", ctypes.c_int),
                ("a_eps0", ctypes.c_double), ("a_eps_min", ctypes.c_double), ("a_factor", ctypes.c_double), ("a_later_eps0", ctypes.c_double),
                ("a_stop_below", ctypes.c_double), ("a_max_free_columns", ctypes.c_double), ("min_eps", ctypes.c_double)]


class _Report(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("rounds", ctypes.c_int), ("violations", ctypes.c_int), ("loose", ctypes.c_int),
                ("tight_within_eps", ctypes.c_int), ("n_tight", ctypes.c_int), ("optimal", ctypes.c_int), ("unique", ctypes.c_int),
                ("n_auction_violated", ctypes.c_int), ("pad_", ctypes.c_int),
                ("bids", ctypes.c_long), ("steps", ctypes.c_long), ("augmentations", ctypes.c_long), ("edges", ctypes.c_long), ("dummy_scans", ctypes.c_long),
                ("slack_bound", ctypes.c_double), ("delta", ctypes.c_double), ("eps", ctypes.c_double), ("seconds_total", ctypes.c_double),
                ("seconds_auction", ctypes.c_double), ("seconds_core", ctypes.c_double), ("seconds_device", ctypes.c_double),
                ("seconds_certify", ctypes.c_double), ("auction_violated", ctypes.c_int * 8), ("violated_per_round", ctypes.c_int * 32)]


def _native_options(min_eps=0.0):
    """This module's settings (CORE_EDGES_PER_ROW ... AUCTION) as the native driver's option record."""
    o = _Options()
    o.core_edges, o.price_edges, o.max_pricing_rounds = min(CORE_EDGES_PER_ROW, 256), min(PRICE_EDGES_PER_ROW, 256), MAX_PRICING_ROUNDS
    o.column_reduction = int(bool(COLUMN_REDUCTION))
    o.rel_delta, o.rel_eps_collect, o.rel_eps_floor, o.eps_safety = REL_DELTA, REL_EPS_COLLECT, REL_EPS_FLOOR, EPS_SAFETY
    a = AUCTION
    o.auction = int(a is not None)
    if a is not None:
        o.a_rounds, o.a_bids_per_row, o.a_later_bids_per_row = int(a["rounds"]), int(a["bids_per_row"]), int(a["later_bids_per_row"])
        o.a_eps0, o.a_eps_min, o.a_factor, o.a_later_eps0 = a["eps0"], a["eps_min"], a["factor"], a["later_eps0"]
        o.a_stop_below, o.a_max_free_columns = a["stop_below"], a["max_free_columns"]
    o.min_eps = float(min_eps)
    return o


def _native_ok(M):
    """The native driver takes a DeviceMatrix with this module's stock flow (no row-reduction warm start, <= 64 candidates per row)."""
    return (NATIVE_DRIVER and isinstance(M, DeviceMatrix) and not M.f32 and ROW_REDUCTION_ROUNDS == 0 and CORE_EDGES_PER_ROW <= 64
            and PRICE_EDGES_PER_ROW <= 64)


def _native_ws(M):
    torch = nat.torch_mod()
    nr, nc = M.shape
    ws = getattr(M, "_resident_ws", None)
    need = int(nat.load().pm_lsap_resident_workspace(nr, nc))
    if ws is None or ws.numel() < need:
        ws = M._resident_ws = torch.empty(need, dtype=torch.uint8, device=M.U.device)
    return ws


def solve_core_native(M, info=None):
    """solve_core for a DeviceMatrix by ONE foreign call (pm_lsap_solve_resident: the interpreter lock is released for the whole
    solve) -> (u, v, col4row) or None, as solve_core."""
    nr, nc = M.shape
    lib = nat.load()
    ws = _native_ws(M)
    u, v, c4r = np.empty(nr), np.empty(nc), np.empty(nr, dtype=np.int32)
    opt, rep = _native_options(), _Report()
    rc = lib.pm_lsap_solve_resident(nat.ptr(M.U), nr, nc, M.ld, ctypes.addressof(opt), u.ctypes.data, v.ctypes.data, c4r.ctypes.data,
                                    ctypes.addressof(rep), nat.ptr(ws), ws.numel(), nat.stream_ptr(M.U))
    nat.check(rc)
    if info is not None:
        info.update(driver="native", rounds=rep.rounds, edges=rep.edges, steps=rep.steps, augmentations=rep.augmentations, dummy_scans=rep.dummy_scans,
                    auction_bids=rep.bids, auction_seconds=rep.seconds_auction, core_seconds=rep.seconds_core,
                    device_seconds=rep.seconds_device, solve_seconds=rep.seconds_total,
                    violated_per_round=list(rep.violated_per_round[:min(rep.rounds, 32)]),
                    auction_violated=list(rep.auction_violated[:rep.n_auction_violated]))
    if rep.status == 4:
        raise ValueError("cost matrix is infeasible")
    if rep.status != 0:
        return None
    return u, v, c4r


def certify_native(M, u, v, col4row, info=None, min_eps=0.0):
    """certify for a DeviceMatrix by one foreign call (pm_lsap_certify_resident); same verdicts, same info keys."""
    nr, nc = M.shape
    lib = nat.load()
    ws = _native_ws(M)
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    c4r = np.ascontiguousarray(col4row, dtype=np.int32)
    cap = 8 * nc + 1024
    tight = np.empty((cap, 2), dtype=np.int32)
    opt, rep = _native_options(min_eps), _Report()
    nat.check(lib.pm_lsap_certify_resident(nat.ptr(M.U), nr, nc, M.ld, ctypes.addressof(opt), u.ctypes.data, v.ctypes.data, c4r.ctypes.data,
                                           tight.ctypes.data, cap, ctypes.addressof(rep), nat.ptr(ws), ws.numel(), nat.stream_ptr(M.U)))
    if info is not None:
        info.update(violations=rep.violations, loose=rep.loose, tight=None if rep.tight_within_eps < 0 else rep.tight_within_eps,
                    slack_bound=rep.slack_bound, delta=rep.delta, eps=rep.eps, optimal=bool(rep.optimal), certify_seconds=rep.seconds_certify)
        if rep.tight_within_eps >= 0:
            info["tight_within_eps"] = rep.tight_within_eps
            info["unique"] = bool(rep.unique)
            if rep.n_tight >= 0:
                info["_tight_edges"] = tight[:rep.n_tight].copy()
    return bool(rep.optimal and rep.unique)


def solve_core(M, info=None):
    """The sparse-core solve of one matrix M [nr, nc], nr <= nc, finite entries -> (u, v, col4row) with no entry of M
    violating dual feasibility beyond delta, or None if M holds non-finite entries / pricing did not converge."""
    if _native_ok(M):
        return solve_core_native(M, info)
    nr, nc = M.shape
    k = min(CORE_EDGES_PER_ROW, 256)
    # square problems start from the column reduction (v = column minima, u = row minima of cost - v, rows matched to their
    # minimising column where it is free): the core is then chosen by REDUCED cost and most rows never need a search
    v0 = M.col_min() if (nr == nc and COLUMN_REDUCTION) else np.zeros(nc)
    if not np.isfinite(v0).all():
        return None
    u_rr = None
    if ROW_REDUCTION_ROUNDS > 0 and hasattr(M, "bid"):
        # warm start: a dozen bidding rounds on the dense rows match most rows and leave prices close to the optimum's
        u_rr, v0, c4r_rr = _row_reduction(M, v0, ROW_REDUCTION_ROUNDS)
    both = M.row_select_with_diagonal(v0, k) if hasattr(M, "row_select_with_diagonal") else None
    if both is not None:
        cols, costs, bad, safety = both
        safety = safety[:nr]
    else:
        cols, costs, bad = M.row_select(v0, k)
        safety = None
    if bad:
        return None
    if safety is None:
        safety = M.diagonal(nr)                          # row i -> column i: the core always holds a perfect matching
    scale = max(float(np.abs(costs[cols >= 0]).max()), float(np.abs(safety).max()), float(np.abs(v0).max()), 1e-300)
    delta = REL_DELTA * scale
    kp = min(PRICE_EDGES_PER_ROW, 256)
    dense_min = costs[:, 0] - v0[cols[:, 0]]              # each row's minimum of cost - v over the DENSE row (rank 0 of the selection)
    with _Core(nr, nc) as core:
        core.add(cols, costs)
        core.add(np.arange(nr, dtype=np.int32)[:, None], safety[:, None])
        if u_rr is not None:
            held = np.flatnonzero(c4r_rr >= 0)
            pair_cost = M.entries(held, c4r_rr[held])
            extra_c = np.full((nr, 1), -1, dtype=np.int32)
            extra_v = np.zeros((nr, 1))
            extra_c[held, 0], extra_v[held, 0] = c4r_rr[held], pair_cost
            core.add(extra_c, extra_v)                    # the matched pairs are core edges, with their exact costs
            u0 = np.where(c4r_rr >= 0, u_rr, dense_min)
            # a held row's dual is its second choice's reduced cost at the time of its bid: never above the dense minimum now
            # (prices only fell since), except by the rounding of (cost - price) itself: keep it feasible to the bit
            u0 = np.minimum(u0, dense_min)
            tight = np.zeros(nr, dtype=bool)
            tight[held] = (pair_cost - v0[c4r_rr[held]]) == u0[held]
            core.init_state(u0, v0, np.where(tight, c4r_rr, -1).astype(np.int32))
        elif nr == nc and COLUMN_REDUCTION:
            core.init_duals(dense_min, v0, cols[:, 0])
        if AUCTION is not None and (nc - nr) <= AUCTION["max_free_columns"] * nc:
            # eps-scaling auctions over the core as a warm start (pm_lsap_core.cpp: auction), each followed by a pricing pass
            # that brings in the dense entries the new duals make attractive — before any shortest-path search is paid for.
            # Unit of eps: the core's width, the mean spread in reduced cost between a row's first and last core entry.
            last = cols[:, k - 1] if cols.shape[1] >= k else np.full(nr, -1)
            have = last >= 0
            spread = (costs[:, k - 1] - v0[np.maximum(last, 0)]) - (costs[:, 0] - v0[np.maximum(cols[:, 0], 0)])
            width = float(np.mean(spread[have])) if have.any() else 0.0
            if width > 0.0 and np.isfinite(width):
                t_a = time.perf_counter()
                eps0 = AUCTION["eps0"] * width
                for a_round in range(AUCTION["rounds"]):
                    budget = AUCTION["bids_per_row"] if a_round == 0 else AUCTION["later_bids_per_row"]
                    bids = core.auction(eps0, AUCTION["eps_min"] * width, AUCTION["factor"], budget * nr)
                    if a_round + 1 == AUCTION["rounds"]:
                        break
                    _, v_a, _, _ = core.get()
                    pc, pcost, _ = M.row_select(v_a, kp)
                    violated = core.reprice(pc, pcost, delta)
                    if info is not None:
                        info.setdefault("auction_violated", []).append(violated)
                    if violated <= AUCTION["stop_below"] * nr:
                        break
                    eps0 = AUCTION["later_eps0"] * width
                if info is not None:
                    info["auction_bids"] = bids
                    info["auction_seconds"] = time.perf_counter() - t_a
        rounds = 0
        t_core = 0.0
        while True:
            t_s = time.perf_counter()
            core.solve()
            t_core += time.perf_counter() - t_s
            u, v, c4r, stats = core.get()
            pc, pcost, _ = M.row_select(v, kp)
            violated = core.reprice(pc, pcost, delta)
            rounds += 1
            if info is not None:
                info.setdefault("violated_per_round", []).append(violated)
                info.setdefault("steps_after_solve", []).append(stats[1])
                info.setdefault("augmentations_after_solve", []).append(stats[2])
            if violated == 0:
                break
            if rounds >= MAX_PRICING_ROUNDS:
                return None
            if hasattr(M, "threshold_select") and violated * LISTING_FROM_FRACTION <= nr:
                # A FilteredMatrix with few offenders left: the listing rounds below see every entry a further selection round could
                # return (and are needed at the end anyway) — solve, then go straight to them: one to two dense passes less per
                # hypothesis (a selection round that only confirms "no violation" used to precede the listing that proves it)
                t_s = time.perf_counter()
                core.solve()
                t_core += time.perf_counter() - t_s
                u, v, c4r, stats = core.get()
                break
        if hasattr(M, "threshold_select"):
            # A FilteredMatrix: the rounds above priced the entries its approximate matrix ranks first.  Now EVERY entry whose exact
            # reduced cost can lie below the collection margin is listed and priced with its exact cost, until none violates: the duals
            # are then feasible on all of them to the bit and every unlisted entry is further than the margin from tight.
            polish = 0
            t_p = time.perf_counter()
            while True:
                lst = M.threshold_select(u, v, c4r)
                if lst is None or not np.isfinite(lst[1][lst[0] >= 0]).all():
                    return None
                violated = core.reprice(lst[0], lst[1], delta)
                polish += 1
                if info is not None:
                    info.setdefault("polish_violated", []).append(violated)
                if violated == 0:
                    if info is not None:
                        info.update(polish_seconds=time.perf_counter() - t_p, polish_list_shape=tuple(lst[0].shape),
                                    polish_listed=int((lst[0] >= 0).sum()))
                    break
                if polish >= MAX_PRICING_ROUNDS:
                    return None
                t_s = time.perf_counter()
                core.solve()
                t_core += time.perf_counter() - t_s
                u, v, c4r, stats = core.get()
        if info is not None:
            info.update(rounds=rounds, edges=stats[0], steps=stats[1], augmentations=stats[2], dummy_scans=stats[3], core_seconds=t_core,
                        column_repairs=core.column_repairs())
    return u, v, c4r


def solve_on_device(U, info=None, force=False):
    """scipy.optimize.linear_sum_assignment(U) for a float64 GPU matrix, without moving it to the host when the sparse-core
    scheme can certify its answer; otherwise (small matrix, non-finite entries, ties) the dense host solver, SciPy's
    algorithm itself.  -> (row_ind, col_ind) int64, rows ascending.  info (dict): which route was taken and its counters."""
    torch = nat.torch_mod()
    if not (nat.is_torch(U) and U.is_cuda and U.dtype == torch.float64 and U.dim() == 2 and U.stride(1) == 1):
        raise ValueError("U must be a float64 GPU matrix with unit column stride")
    info = {} if info is None else info
    n0, m0 = U.shape
    if min(n0, m0) >= (1 if force else DEVICE_MIN_ROWS):
        with torch.cuda.device(U.device):
            W = DeviceMatrix(U if n0 <= m0 else transposed(U))       # rows are the short side (SciPy transposes likewise)
            sol = solve_core(W, info)
            if sol is not None and certify(W, *sol, info=info):
                info["route"] = "device"
                return _answer(sol[2], n0, m0)
    info["route"] = "host"
    return linear_sum_assignment(U.cpu().numpy())


DENSE_FALLBACK_MAX_ENTRIES = 1 << 30      # matrices above this many entries are never handed to the dense host solver (hours)


def solve_pair_on_device(U_h, U_twin, info_h=None, info_twin=None, allow_host=True, accept_near_ties=False, min_eps=0.0,
                         exact_entries=None, cost_delta=0.0):
    """One hypothesis and its twin (the same terms summed in another order: U11/U22, U12/U21, U13/U24, U14/U23), both float64
    GPU matrices [N, M]: the hypothesis is solved on its sparse core and certified; the twin first tries its sibling's duals —
    accepted only if they are a certified unique optimum of the twin's OWN entries — and is solved on its own otherwise.
    -> [(row_ind, col_ind) or None, (row_ind, col_ind) or None]; None only with allow_host=False (or a matrix too large for
    the dense solver): the hypothesis could not be certified and the dense host solver was not allowed.
    accept_near_ties: where the dense solver is not an option, take an assignment that is certified OPTIMAL but not proven
    unique (an alternative within ~1e-11 of the total cost exists; SciPy might return either) instead of None.
    exact_entries / cost_delta: U_h is a relaxed matrix within cost_delta of the exact one; exact_entries(rows, cols) -> (exact
    entries of the hypothesis's matrix, of the twin's) for index arrays into the n x m matrices.  The assignment solved on U_h
    is then certified against the EXACT matrices on their listed entries (certify_listed); an answer that does not certify
    comes back as None (the caller rebuilds that pairing exactly); U_twin is not read."""
    n, m = U_h.shape
    info_h = {} if info_h is None else info_h
    info_twin = {} if info_twin is None else info_twin
    host_ok = allow_host and n * m <= DENSE_FALLBACK_MAX_ENTRIES
    refused = "uncertified (dense solver not allowed)" if not allow_host else "uncertified (too large for the dense solver)"
    if min(n, m) < DEVICE_MIN_ROWS:
        if exact_entries is not None:
            return [None, None]                              # (a relaxed matrix is never handed to the dense solver)
        info_h["route"] = info_twin["route"] = "host"
        return [linear_sum_assignment(U_h.cpu().numpy()), linear_sum_assignment(U_twin.cpu().numpy())]
    out = [None, None]
    near_tie = "device (optimal, a near-tie within the margin: not proven to be SciPy's pick)"
    W = DeviceMatrix(U_h if n <= m else transposed(U_h))
    Wt = None
    sol = solve_core(W, info_h)
    if exact_entries is not None:
        # U_h is a RELAXED matrix (its twin's relaxed matrix is the same numbers): one solve, then the exact matrices' certificate
        # on the listed entries (certify_listed) for the hypothesis and for the twin; whatever does not certify stays None
        if sol is None:
            return out
        fetch = exact_entries if n <= m else (lambda rows, cols: exact_entries(cols, rows))      # (the solve ran on the transpose)
        ok = certify_listed(W, *sol, exact_entries=fetch, cost_delta=cost_delta, infos=[info_h, info_twin])
        if ok[0]:
            info_h["route"] = "device"
            out[0] = _answer(sol[2], n, m)
        if len(ok) > 1 and ok[1]:
            info_twin["route"] = "device (sibling's duals certified)"
            out[1] = _answer(sol[2], n, m)
        return out
    if sol is not None and certify(W, *sol, info=info_h, min_eps=min_eps):
        info_h["route"] = "device"
        out[0] = _answer(sol[2], n, m)
        W = None                                        # (possibly a transposed copy: release it before the twin's is made)
        Wt = DeviceMatrix(U_twin if n <= m else transposed(U_twin))
        if certify(Wt, *sol, info=info_twin, min_eps=min_eps):
            info_twin["route"] = "device (sibling's duals certified)"
            out[1] = out[0]
            return out
    elif host_ok:
        info_h["route"] = "host"
        out[0] = linear_sum_assignment(U_h.cpu().numpy())
    elif sol is not None and info_h.get("optimal") and _resolved(W, sol, info_h, out, 0, n, m):
        pass
    elif accept_near_ties and sol is not None and info_h.get("optimal"):
        info_h["route"] = near_tie
        out[0] = _answer(sol[2], n, m)
    else:
        info_h["route"] = refused
    # the twin on its own
    W = None
    if Wt is None:
        Wt = DeviceMatrix(U_twin if n <= m else transposed(U_twin))
    sol_t = solve_core(Wt, info_twin)
    if sol_t is not None and certify(Wt, *sol_t, info=info_twin, min_eps=min_eps):
        info_twin["route"] = "device"
        out[1] = _answer(sol_t[2], n, m)
    elif host_ok:
        info_twin["route"] = "host"
        out[1] = linear_sum_assignment(U_twin.cpu().numpy())
    elif sol_t is not None and info_twin.get("optimal") and _resolved(Wt, sol_t, info_twin, out, 1, n, m):
        pass
    elif accept_near_ties and sol_t is not None and info_twin.get("optimal"):
        info_twin["route"] = near_tie
        out[1] = _answer(sol_t[2], n, m)
    else:
        info_twin["route"] = refused
    return out


def _resolved(W, sol, info, out, slot, n, m):
    """solve_pair_on_device: settle a certified optimum's near-ties on their blocks (resolve_near_ties) -> True if out[slot] was set."""
    c4r = resolve_near_ties(W, sol, info)
    if c4r is None:
        return False
    info["route"] = "device (optimal; near-tie settled by the dense algorithm on %d row(s) in %d block(s))" % (sum(info["resolved_groups"]), len(info["resolved_groups"]))
    out[slot] = _answer(c4r, n, m)
    return True


PIPELINED_PRIORITY = -1


# ---- where the four pairings' solver threads run ------------------------------------------------------------------------------------
# The host core of a 50 000-nucleus hypothesis walks ~20 MB (core edges, column records) in cache-hostile order.  Measured on the GPU
# box (EPYC 9575F: 16 CCDs of 8 cores with 32 MB of L3 each; profiles/r05_filter_phases.txt): one pairing alone 0.27-0.34 s of auction +
# shortest paths, the same four side by side 0.39-0.60 s each — threads started from one parent land on one CCD and share its L3.
# pin_solver_thread(slot) gives each of the four its own L3 domain (same NUMA node as the caller where possible).
PIN_SOLVER_THREADS = os.environ.get("PM_LSAP_PIN", "1") != "0"
# (A softer form — every fourth domain of the caller's NUMA node per thread, so that the scheduler could still dodge a busy core — was
# measured equal: nine warm 50k registrations each, median 865 ms pinned to one domain, 861 ms to a class of two, 897 ms unpinned
# with a tail up to 1 291 ms; profiles/r05_pin_ab.txt.  The simpler form stays.)
_L3_DOMAINS = None


def _l3_domains():
    """The L3 domains (lists of logical CPUs) this process may run on, the caller's own first, then its NUMA node's, then the rest."""
    global _L3_DOMAINS
    if _L3_DOMAINS is not None:
        return _L3_DOMAINS
    doms = []
    try:
        allowed = os.sched_getaffinity(0)
        seen = {}
        for c in sorted(allowed):
            try:
                with open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % c) as f:
                    key = f.read().strip()
            except OSError:
                key = "all"
            seen.setdefault(key, []).append(c)
        here = None
        try:
            here = os.sched_getcpu() if hasattr(os, "sched_getcpu") else int(ctypes.CDLL(None).sched_getcpu())
            if here < 0:
                here = None
        except (OSError, AttributeError):
            here = None

        def node_of(c):
            try:
                for name in os.listdir("/sys/devices/system/cpu/cpu%d" % c):
                    if name.startswith("node"):
                        return int(name[4:])
            except (OSError, ValueError):
                pass
            return 0
        home = node_of(here) if here is not None else 0
        mine = next((k for k, cpus in enumerate(sorted(seen.values(), key=lambda c: c[0])) if here in cpus), 0)
        order = sorted(seen.values(), key=lambda c: c[0])
        # the caller's own domain first (the scheduler put it somewhere sensible), then its neighbours on the same NUMA node, then the rest
        doms = sorted(order, key=lambda cpus: (0 if (here in cpus) else 1 if node_of(cpus[0]) == home else 2,
                                               (order.index(cpus) - mine) % max(len(order), 1)))
    except (AttributeError, OSError):
        doms = []
    _L3_DOMAINS = doms
    return doms


_PIN_LOCAL = threading.local()


def set_pin_base(base):
    """For callers that run several registrations side by side (pipeline.estimate_transform_batch's workers): the solver threads
    started from THIS thread take the L3 domains base, base + 1, ... instead of 0, 1, ...; None = do not pin them at all.
    (Measured on BASELINE config 5, eight workers x four solver threads on a 16-CPU quota: pinned with spread bases 15.4
    registrations/s unseeded, unpinned 17.9 — the pairs are small enough for any cache and the scheduler balances 32 threads better
    than a fixed map; the batch driver passes None.)"""
    _PIN_LOCAL.base = None if base is None else int(base)


def _pin_base():
    return getattr(_PIN_LOCAL, "base", 0)


def pin_solver_thread(slot):
    """Restrict the CALLING thread to the slot-th L3 domain (no-op for slot None, with fewer than two domains, off Linux, or
    PM_LSAP_PIN=0)."""
    if not PIN_SOLVER_THREADS or slot is None:
        return None
    doms = _l3_domains()
    if len(doms) < 2:
        return None
    cpus = doms[slot % len(doms)]
    try:
        os.sched_setaffinity(0, cpus)            # (Linux: pid 0 = the calling THREAD)
    except OSError:
        return None
    return cpus


def solve_eight_on_device(U8, info=None, allow_host=True, accept_near_ties=False, ready=None, min_eps=0.0, exact_rebuild=None,
                          exact_entries=None, cost_delta=0.0):
    """The widget's eight assignments (_dock_widget.py:604-611) for U8 [8, N, M] on the GPU: hypotheses 11, 12, 13, 14 are
    solved (four host threads drive their core solves and kernels concurrently), each together with its twin (22, 21, 24,
    23: solve_pair_on_device).  -> list of eight (row_ind, col_ind).
    allow_host=False: never take the dense host solver (hours at 50 000 nuclei); a hypothesis that cannot be certified comes
    back as None instead (bench.py's bounded extra leg).
    ready: four events, one per pairing (U11/U22, U12/U21, U13/U24, U14/U23), recorded when that pairing's two matrices have
    been written (_kernels.chi2_cost8_frame1_by_pairings): its solve starts then, while later pairings are still being built.
    (Measured at 50 000 nuclei, twice: no gain.  Before the auction warm start the three wrong-frame hypotheses needed the same
    ~2.8 s each, so the last one built decided (4.5 s against 4.0 s); with it the solves are short but made of dense passes
    that queue behind the cost kernel (2.2 s against 1.8 s).  Round 3 gave the solver's streams priority over the cost
    kernel's (PIPELINED_PRIORITY; round-3 probe, git history): no change, 2.14-2.32 s against 1.61-1.98 s back to back
    (profiles/r03_pipelined_probe.txt) — the cost kernel's resident workgroups hold every CU for the length of a pairing.
    The driver does not use it.)"""
    torch = nat.torch_mod()
    n, m = U8.shape[1], U8.shape[2]
    out = [None] * 8
    infos = [dict() for _ in range(8)]
    if min(n, m) < DEVICE_MIN_ROWS:       # eight dense host solves on eight threads, each fetching its own matrix (the round-1 path)
        stream = torch.cuda.current_stream(U8.device)

        def fetch(h):
            with torch.cuda.stream(stream):
                return U8[h].cpu().numpy()
        res = solve_many([lambda h=h: fetch(h) for h in range(8)])
        if info is not None:
            info["routes"] = ["host"] * 8
        return res

    caller = torch.cuda.current_stream(U8.device).cuda_stream          # (concurrent callers, e.g. batch workers, keep apart)

    pin_base = _pin_base()

    def pair(h):
        pin_solver_thread(None if pin_base is None else pin_base + h)
        twin = [t for t, s in TWINS.items() if s == h][0]
        # persistent: the allocator's cache is per stream; pipelined behind the cost build the dense passes must get in front of it
        stream = nat.side_stream(U8.device, ("pair", caller, h), priority=PIPELINED_PRIORITY if ready is not None else 0)
        with torch.cuda.device(U8.device), torch.cuda.stream(stream):
            if ready is not None:
                stream.wait_event(ready[h])
            if exact_rebuild is None:
                out[h], out[twin] = solve_pair_on_device(U8[h], U8[twin], infos[h], infos[twin], allow_host, accept_near_ties, min_eps)
            else:
                # RELAXED matrices (pm_chi2_cost8_relaxed: every entry within cost_delta of the exact cost).  An answer counts only
                # if it is proven to be the EXACT matrix's unique optimum: with exact_entries(h) (-> the function that answers the
                # listed entries of hypothesis h's and its twin's exact matrices) by the certificate on the exact matrix's listed
                # entries (certify_listed: the exact mode's own margins); without it by a uniqueness margin of min_eps = 2 min(N,
                # M) delta on the relaxed matrix (the round-4 first form: at 50 000 nuclei that margin is rarely there).
                # Anything else (a near-tie inside the margin, ties, non-finite costs) has this pairing's two matrices rebuilt by
                # the exact kernel, in place, and solved as usual.
                if exact_entries is not None:
                    got = solve_pair_on_device(U8[h], U8[twin], infos[h], infos[twin], False, False, exact_entries=exact_entries(h),
                                               cost_delta=cost_delta)
                    how = "relaxed (solved on the relaxed matrix, certified on the exact matrix's listed entries)"
                else:
                    got = solve_pair_on_device(U8[h], U8[twin], infos[h], infos[twin], False, False, min_eps)
                    how = "relaxed (certified with margin %.1e)" % min_eps
                ok = all(g is not None and i.get("route") in ("device", "device (sibling's duals certified)") for g, i in zip(got, (infos[h], infos[twin])))
                if ok:
                    for i in (infos[h], infos[twin]):
                        i["cost_mode"] = how
                else:
                    infos[h].clear()
                    infos[twin].clear()
                    exact_rebuild(h)
                    got = solve_pair_on_device(U8[h], U8[twin], infos[h], infos[twin], allow_host, accept_near_ties)
                    for i in (infos[h], infos[twin]):
                        i["cost_mode"] = "exact (rebuilt: the relaxed matrices did not certify)"
                out[h], out[twin] = got
            stream.synchronize()

    if ready is None:
        torch.cuda.current_stream(U8.device).synchronize()  # U8 was produced on the caller's stream
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(pair, range(4)))
    if info is not None:
        info["routes"] = [i.get("route") for i in infos]
        info["details"] = infos
    return out


def solve_four_filtered(F4, exact_entries, exact_entries_t, cost_delta, exact_pair, info=None, allow_host=True, accept_near_ties=False,
                        build=None, shape=None, device=None, in_flight=4, storage=None):
    """The widget's eight assignments from FOUR approximate matrices F4 (pm_chi2_filter4, float64 or float32 storage: matrix t within
    cost_delta of hypothesis PAIRINGS[t][0]'s exact matrix and of its twin's; for shape = (N, M) with N > M the matrices hold the
    TRANSPOSED problem, [4, M, N] — built with the descriptors' roles swapped, see below), none of the exact matrices built: per pairing (four host threads)
    the sparse-core solve runs on a FilteredMatrix — the approximate matrix selects entries, every cost comes from
    exact_entries(t)(rows, cols) -> (hypothesis's exact values, twin's) [exact_entries_t: the same with GPU tensors] — and the
    result is certified against both exact matrices on their listed entries (certify_listed).  A pairing that cannot be certified
    (ties, near-ties, non-finite costs) gets its two exact matrices from exact_pair(t) -> [2, N, M] and goes the exact mode's
    way (solve_pair_on_device), one such pairing at a time.  -> list of eight (row_ind, col_ind).
    STREAMED form (F4 = None; clouds whose four filter matrices do not fit in HBM together): build(t, out) -> the filter matrix of
    pairing t, written into the buffer `out`, with the SHORT side as its rows ([N, M] if N <= M, else [M, N]: pm_chi2_filter_pair with the roles swapped — the
    terms are symmetric in their two descriptors and each pairing's bin map is an involution, so that IS the transposed filter
    to within cost_delta), made on the pairing's own stream when its turn comes; at most in_flight pairings are resident, in
    buffers of dtype `storage` (default float64) allocated once."""
    torch = nat.torch_mod()
    if F4 is not None:
        n, m = shape if shape is not None else (F4.shape[1], F4.shape[2])
        if tuple(F4.shape[1:]) != (min(n, m), max(n, m)):
            raise ValueError("F4 must hold the filter matrices with the short side as rows: [4, %d, %d]" % (min(n, m), max(n, m)))
        device = F4.device
        in_flight = 4
    else:
        n, m = shape
    out = [None] * 8
    infos = [dict() for _ in range(8)]
    pairs = [(h, [t for t, s_ in TWINS.items() if s_ == h][0]) for h in range(4)]
    caller = torch.cuda.current_stream(device).cuda_stream
    exact_turn = threading.Lock()

    pin_base = _pin_base()

    def pair(t):
        pin_solver_thread(None if pin_base is None else pin_base + t)
        h, twin = pairs[t]
        stream = nat.side_stream(device, ("pair", caller, h))
        with torch.cuda.device(device), torch.cuda.stream(stream):
            got = [None, None]
            if min(n, m) >= DEVICE_MIN_ROWS:
                fetch, fetch_t = exact_entries(t), exact_entries_t(t)
                if n > m:                                       # the solve runs on the transpose: rows are fixed nuclei
                    f0, f0_t = fetch, fetch_t
                    fetch, fetch_t = (lambda rows, cols: f0(cols, rows)), (lambda rows, cols: f0_t(cols, rows))
                slot = None
                if F4 is not None:
                    W = DeviceMatrix(F4[t])
                else:
                    slot = pool.get()                           # one of the in_flight buffers, allocated once (22 ms per GB on this pool)
                    W = DeviceMatrix(build(t, slot))
                M = FilteredMatrix(W, fetch, cost_delta, fetch_t)
                sol = solve_core(M, infos[h])
                if sol is not None:
                    ok = certify_listed(M, *sol, exact_entries=fetch, cost_delta=cost_delta, infos=[infos[h], infos[twin]])
                    if len(ok) == 2 and all(ok):
                        got = [_answer(sol[2], n, m)] * 2
                        infos[h]["route"], infos[twin]["route"] = "device", "device (sibling's duals certified)"
                        for i in (infos[h], infos[twin]):
                            i["cost_mode"] = "filter (approximate matrix as selector, exact costs on %d listed entries)" % M.exact_evaluated
                del W, M
                if slot is not None:
                    stream.synchronize()                        # the solve's last passes have read the buffer
                    pool.put(slot)
            if (got[0] is None or got[1] is None) and F4 is None:
                deferred.append(t)      # streamed: the exact matrices are built once every filter matrix has been released
                stream.synchronize()
                return
            if got[0] is None or got[1] is None:
                exact(t)
            else:
                out[h], out[twin] = got
            stream.synchronize()

    def exact(t):
        h, twin = pairs[t]
        infos[h].clear()
        infos[twin].clear()
        with exact_turn:        # one pairing's exact matrices at a time: four filter matrices + two exact ones never exceed the exact mode's eight
            U2 = exact_pair(t)
            out[h], out[twin] = solve_pair_on_device(U2[0], U2[1], infos[h], infos[twin], allow_host, accept_near_ties)
            del U2
        for i in (infos[h], infos[twin]):
            i["cost_mode"] = "exact (built: the filtered solve did not certify)"

    deferred = []
    workers = max(1, min(4, int(in_flight)))
    pool = None
    if F4 is None:
        import queue
        pool = queue.Queue()
        with torch.cuda.device(device):
            for _ in range(workers):
                pool.put(big_empty((min(n, m), max(n, m)), storage or torch.float64, device))
    torch.cuda.current_stream(device).synchronize()       # the descriptors / F4 were produced on the caller's stream
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(pair, range(4)))
    pool = None                                           # (the buffers go back to the allocator before any exact matrix is built)
    for t in sorted(deferred):
        with torch.cuda.device(device):
            exact(t)
    if info is not None:
        info["routes"] = [i.get("route") for i in infos]
        info["details"] = infos
    return out


def _answer(c4r, n, m):
    """col4row of the (possibly transposed) problem -> SciPy's (row_ind ascending, col_ind) for the n x m matrix."""
    if n <= m:
        return np.arange(n, dtype=np.int64), c4r.astype(np.int64)
    order = np.argsort(c4r, kind="stable")
    return c4r[order].astype(np.int64), order.astype(np.int64)
