"""Linear sum assignment on the host without the GIL (pm_lsap_solve, platymatch_amd/csrc/pm_lsap.cpp).

The widget calls scipy.optimize.linear_sum_assignment eight times in a row (platymatch/_dock_widget.py:604-611); SciPy
keeps the GIL, so those solves cannot overlap.  pm_lsap_solve restates SciPy 1.15.3's solver operation for operation
(identical indices, ties included — tests/test_lsap.py checks it against SciPy itself) and is called through ctypes,
which releases the GIL: `solve_many` runs the eight solves on eight host threads.  Host code only; nothing here runs on
the GPU."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as nat


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
