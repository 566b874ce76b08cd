"""Host tables for get_shape_context called with its own binning arguments (shape_context.py:10-58).

The reference bins a neighbour through np.arccos, np.arctan2 and two float floor divisions.  Which float64 inputs fall into
which bin is a property of the HOST's NumPy / libm, so the tables are built with exactly those calls, on scalars as the
reference makes them:

  cos_steps[k]  the largest float64 c in [-1, 1] with  np.arccos(c) // (np.pi / n_thetabins) >= k + 1   (k = 0 .. K-1, decreasing);
                theta_index(c) = #{k : c <= cos_steps[k]} for EVERY c: arccos never has to be evaluated on the device;
  phi_steps[m]  the smallest float64 phi in [0, 2 pi] with  phi // (2 * np.pi / n_phibins) >= m + 1   (increasing);
                phi_index(phi) = #{m : phi >= phi_steps[m]} for every wrapped angle phi.

Both are found by bisection over the ordered float64 bit patterns (64 evaluations per step) and checked for a clean step within
64 ulps either side.  The kernel (pm_shape_context_neighbors_binned) leaves neighbours whose device atan2 lies within 2^-46 of a
phi step to bin_rows() below: the reference's own expressions on the host, for those rows only.  Not on the hot path: get_unary
always uses the default 5 x 6 x 12 binning, whose tables are compiled in (csrc/pm_bin_tables.h)."""
import functools
import struct

import numpy as np

MAX_STEPS = 4096


def _ordered(x):
    b = struct.unpack("<q", struct.pack("<d", float(x)))[0]
    return b if b >= 0 else -(b & 0x7FFFFFFFFFFFFFFF)


def _from_ordered(k):
    b = k if k >= 0 else ((-k) | (1 << 63))
    return struct.unpack("<d", struct.pack("<Q", b & 0xFFFFFFFFFFFFFFFF))[0]


def _theta_index(c, n_thetabins):
    with np.errstate(all="ignore"):
        return np.arccos(np.float64(c)) // (np.pi / n_thetabins)                 # shape_context.py:31, :51


def _phi_index(phi, n_phibins):
    with np.errstate(all="ignore"):
        return np.float64(phi) // (2 * np.pi / n_phibins)                        # :52


def _clean_step(f, at, lower_side_value, upper_side_value, span=64):
    """f is a step at ordered position `at`: f == lower_side_value for the `span` floats up to and including it, f == upper_side_value
    for the `span` floats above (values are compared as 'reaches the step or not' by the caller's predicate)."""
    for d in range(span):
        if not lower_side_value(f(_from_ordered(at - d))) or not upper_side_value(f(_from_ordered(at + 1 + d))):
            return False
    return True


@functools.lru_cache(maxsize=64)
def cos_steps(n_thetabins):
    n_thetabins = int(n_thetabins)
    top = float(_theta_index(-1.0, n_thetabins))                                   # theta = pi: the highest index there is
    if not (top == top) or top < 0 or top > MAX_STEPS:
        raise ValueError("n_thetabins = %r: unsupported" % (n_thetabins,))
    out = []
    lo_end, hi_end = _ordered(-1.0), _ordered(1.0)
    for k in range(1, int(top) + 1):
        reach = lambda v: v >= k
        if reach(_theta_index(1.0, n_thetabins)):
            out.append(1.0)
            continue
        lo, hi = lo_end, hi_end                                                    # reach(lo), not reach(hi); arccos decreases
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if reach(_theta_index(_from_ordered(mid), n_thetabins)):
                lo = mid
            else:
                hi = mid
        span = min(64, lo - lo_end, hi_end - hi)
        if not _clean_step(lambda c: _theta_index(c, n_thetabins), lo, reach, lambda v: not reach(v), span):
            raise RuntimeError("np.arccos is not monotone around c = %r: the threshold tables cannot represent it" % _from_ordered(lo))
        out.append(_from_ordered(lo))
    return np.array(out, dtype=np.float64)


@functools.lru_cache(maxsize=64)
def phi_steps(n_phibins):
    n_phibins = int(n_phibins)
    two_pi = 2 * np.pi
    top = float(_phi_index(two_pi, n_phibins))
    if not (top == top) or top < 0 or top > MAX_STEPS:
        raise ValueError("n_phibins = %r: unsupported" % (n_phibins,))
    out = []
    lo_end, hi_end = _ordered(0.0), _ordered(two_pi)
    for m in range(1, int(top) + 1):
        reach = lambda v: v >= m
        lo, hi = lo_end, hi_end                                                    # not reach(lo), reach(hi)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if reach(_phi_index(_from_ordered(mid), n_phibins)):
                hi = mid
            else:
                lo = mid
        span = min(64, lo - lo_end, hi_end - hi)
        if not _clean_step(lambda p: _phi_index(p, n_phibins), lo, lambda v: not reach(v), reach, span):
            raise RuntimeError("floor division is not monotone around phi = %r" % _from_ordered(hi))
        out.append(_from_ordered(hi))
    return np.array(out, dtype=np.float64)


def r_edges(r_inner, r_outer, n_rbins):
    with np.errstate(all="ignore"):
        return np.logspace(np.log10(r_inner), np.log10(r_outer), int(n_rbins))    # :24


def check_arguments(n_rbins, n_thetabins, n_phibins):
    for name, v in (("n_rbins", n_rbins), ("n_thetabins", n_thetabins), ("n_phibins", n_phibins)):
        if int(v) != v or v < 1 or v > MAX_STEPS:
            raise ValueError("%s = %r: a whole number in 1..%d expected" % (name, v, MAX_STEPS))
    if int(n_rbins) * int(n_thetabins) * int(n_phibins) > (1 << 24):
        raise ValueError("more than 2^24 bins")


def bin_rows(rows, mean_dist, edges, n_thetabins, n_phibins):
    """The reference's own expressions (shape_context.py:25-35, 46-58) for a FEW neighbours (rows [k, 3]: the ones the kernel
    listed) -> integer bin per row, -1 = not counted."""
    n_rbins = len(edges)
    n_bins = n_rbins * n_thetabins * n_phibins
    out = np.full(len(rows), -1, dtype=np.int64)
    with np.errstate(all="ignore"):
        for t, (x_, y_, z_) in enumerate(np.asarray(rows, dtype=np.float64)):
            r_ = np.linalg.norm(np.array([x_, y_, z_]))
            r = r_ / mean_dist
            theta = np.arccos(z_ / r_)
            phi = np.arctan2(y_, x_)
            if phi < 0:
                phi = 2 * np.pi + phi
            r_index = n_rbins - 1
            for ind, edge in enumerate(edges):
                if r < edge:
                    r_index = ind
                    break
            index = r_index * n_thetabins * n_phibins + theta // (np.pi / n_thetabins) * n_phibins + phi // (2 * np.pi / n_phibins)
            if index == index and 0 <= index < n_bins and index == np.floor(index):
                out[t] = int(index)
    return out
