"""Device-level wrappers: GPU tensors in, GPU tensors out, one C-ABI call each.

Every function validates shapes, dtypes and index ranges on the host before a kernel is
enqueued (a faulting kernel can take the whole node down), allocates outputs/workspace with
torch (whole matrices of 4 GiB and more: device_memory.big_empty), and launches on torch's current stream.  Clouds are float64 [3, N] contiguous.
"""
import os

import numpy as np

from . import _native as nat
from ._native import NBINS, ICP_NSUMS, check, ptr
from .device_memory import big_empty


def _t():
    return nat.torch_mod()


def _cloud(x, name="cloud"):
    torch = _t()
    if not (nat.is_torch(x) and x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.shape[0] == 3
            and x.is_contiguous()):
        raise ValueError("%s must be a contiguous float64 GPU tensor of shape [3, N]" % name)
    if x.shape[1] < 1:
        raise ValueError("%s is empty" % name)
    if x.shape[1] >= 2 ** 31:
        raise ValueError("%s too large" % name)
    _here(x, name)
    return x


def _here(x, name):
    """Kernels are enqueued on the current stream of the CURRENT device, and operands must live there (a launch on another GPU's
    stream against foreign memory faults).  Every public wrapper of this module runs under the device of its first GPU operand
    (_on_operand_device below: SURVEY.md §8b "set its device per call"), so this only fires when operands of ONE call live on
    different devices."""
    cur = _t().cuda.current_device()
    if x.device.index != cur:
        raise ValueError("%s lives on cuda:%s but the call's other operands (and its stream) are on cuda:%d: all operands of one call "
                         "must live on one device" % (name, x.device.index, cur))


def _vec(x, n, name):
    torch = _t()
    if not (nat.is_torch(x) and x.is_cuda and x.dtype == torch.float64 and x.numel() == n and x.is_contiguous()):
        raise ValueError("%s must be a contiguous float64 GPU tensor with %d elements" % (name, n))
    return x


def _idx(x, n, hi, name, trusted=False):
    """int32 GPU index vector of length n with values in [0, hi) — range-checked here, not in the kernel.
    `trusted`: the vector was produced by one of our own kernels (e.g. icp_nn); skip the two read-backs."""
    torch = _t()
    if not (nat.is_torch(x) and x.is_cuda and x.dtype == torch.int32 and x.is_contiguous() and x.numel() == n):
        raise ValueError("%s must be a contiguous int32 GPU tensor with %d elements" % (name, n))
    if n and not trusted and (int(x.min()) < 0 or int(x.max()) >= hi):
        raise IndexError("%s has entries outside [0, %d)" % (name, hi))
    return x


def centroid(xyz, sequential=False):
    """Mean of each coordinate row in NumPy's own summation order (the reference's bits): pairwise pieces for the 3 x N layout,
    one point after the other (sequential=True) for what np.mean does with an N x 3 array."""
    xyz = _cloud(xyz)
    out = _t().empty(3, dtype=_t().float64, device=xyz.device)
    if sequential:
        check(nat.load().pm_centroid_sequential(ptr(xyz), xyz.shape[1], ptr(out), nat.stream_ptr()))
    else:
        check(nat.load().pm_centroid(ptr(xyz), xyz.shape[1], ptr(out), 0, 0, nat.stream_ptr()))
    return out


def mean_distance(xyz):
    xyz = _cloud(xyz)
    n = xyz.shape[1]
    if n < 2:
        raise ValueError("mean distance needs at least two points")
    lib = nat.load()
    ws = nat.workspace(lib.pm_mean_distance_workspace(n), xyz.device)
    out = _t().empty(1, dtype=_t().float64, device=xyz.device)
    check(lib.pm_mean_distance(ptr(xyz), n, ptr(out), ptr(ws), ws.numel(), nat.stream_ptr()))
    return out


def mean_distance_partials(xyz, row_offset, row_stride):
    """Tile partial sums of the mean pairwise distance for the tile rows row_offset, row_offset + row_stride, ...
    (zeros elsewhere) -> float64 GPU vector; ranks sum these element-wise, then mean_distance_finish."""
    xyz = _cloud(xyz)
    n = xyz.shape[1]
    if n < 2:
        raise ValueError("mean distance needs at least two points")
    if not (0 <= int(row_offset) < int(row_stride)):
        raise ValueError("need 0 <= row_offset < row_stride")
    lib = nat.load()
    nbytes = lib.pm_mean_distance_workspace(n)
    part = _t().empty(nbytes // 8, dtype=_t().float64, device=xyz.device)
    check(lib.pm_mean_distance_rows(ptr(xyz), n, int(row_offset), int(row_stride), ptr(part), nbytes, nat.stream_ptr()))
    return part


def mean_distance_finish(partials, n):
    torch = _t()
    lib = nat.load()
    if not (nat.is_torch(partials) and partials.is_cuda and partials.dtype == torch.float64 and partials.is_contiguous()
            and n >= 2 and partials.numel() * 8 >= lib.pm_mean_distance_workspace(n)):
        raise ValueError("partials must be the float64 GPU vector mean_distance_partials returned for this n")
    out = torch.empty(1, dtype=torch.float64, device=partials.device)
    check(lib.pm_mean_distance_finish(ptr(partials), int(n), ptr(out), nat.stream_ptr()))
    return out


def pca_axis(xyz):
    xyz = _cloud(xyz)
    if xyz.shape[1] < 2:
        raise ValueError("PCA axis needs at least two points")
    out = _t().empty(3, dtype=_t().float64, device=xyz.device)
    check(nat.load().pm_pca_axis(ptr(xyz), xyz.shape[1], ptr(out), 0, 0, nat.stream_ptr()))
    return out


def pca_components(xyz):
    """-> [3, 3]: sklearn PCA(3).fit(X).components_ (rows by decreasing variance, sign convention of svd_flip)."""
    xyz = _cloud(xyz)
    if xyz.shape[1] < 2:
        raise ValueError("PCA needs at least two points")
    out = _t().empty((3, 3), dtype=_t().float64, device=xyz.device)
    check(nat.load().pm_pca_components(ptr(xyz), xyz.shape[1], ptr(out), nat.stream_ptr()))
    return out


def cdist(a, b, out=None):
    """Euclidean distance matrix of two [3, n], [3, m] clouds -> [n, m] (scipy cdist semantics)."""
    torch = _t()
    a, b = _cloud(a, "a"), _cloud(b, "b")
    n, m = a.shape[1], b.shape[1]
    if out is None:
        out = big_empty((n, m), torch.float64, a.device)
    elif not (out.is_cuda and out.dtype == torch.float64 and tuple(out.shape) == (n, m) and out.stride(1) == 1):
        raise ValueError("out must be float64 GPU [n, m] with unit column stride")
    check(nat.load().pm_cdist(ptr(a), n, ptr(b), m, ptr(out), out.stride(0), nat.stream_ptr()))
    return out


def row_argmin(U, return_values=False):
    """np.argmin(U, axis=-1) of a float64 GPU cost matrix [rows, cols] or stack [n_mat, rows, cols] -> int32 indices
    (first minimum; first NaN if the row holds one), optionally with the minima."""
    torch = _t()
    if not (nat.is_torch(U) and U.is_cuda and U.dtype == torch.float64 and U.dim() in (2, 3) and U.numel() > 0
            and U.stride(-1) == 1):
        raise ValueError("U must be a non-empty float64 GPU tensor [rows, cols] or [n_mat, rows, cols] with unit column stride")
    stack = U if U.dim() == 3 else U.unsqueeze(0)
    n_mat, rows, cols = stack.shape
    ld = stack.stride(1) if rows > 1 else cols
    ms = stack.stride(0) if n_mat > 1 else rows * ld
    if ld < cols or (n_mat > 1 and ms < (rows - 1) * ld + cols):
        raise ValueError("U rows / matrices must not overlap")
    idx = torch.empty((n_mat, rows), dtype=torch.int32, device=U.device)
    val = torch.empty((n_mat, rows), dtype=torch.float64, device=U.device) if return_values else None
    check(nat.load().pm_row_argmin(ptr(stack), n_mat, rows, cols, ld, ms, ptr(idx), ptr(val) if val is not None else None,
                                   nat.stream_ptr()))
    if U.dim() == 2:
        idx = idx[0]
        val = val[0] if val is not None else None
    return (idx, val) if return_values else idx


def label_moments(labels, n_labels=None):
    """labels: int32 GPU [Z, Y, X] -> (counts [L] int64, sums [3, L] int64 of z, y, x indices), L = max label + 1."""
    torch = _t()
    if not (nat.is_torch(labels) and labels.is_cuda and labels.dtype == torch.int32 and labels.dim() == 3
            and labels.is_contiguous() and labels.numel() > 0):
        raise ValueError("labels must be a contiguous int32 GPU tensor [Z, Y, X]")
    if n_labels is None:
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0:
            raise ValueError("labels must be non-negative")
        n_labels = max(hi + 1, 2)
    if n_labels > 2 ** 27:
        raise ValueError("label values above 2^27 are not supported")
    nz, ny, nx = labels.shape
    counts = torch.empty(n_labels, dtype=torch.int64, device=labels.device)
    sums = torch.empty((3, n_labels), dtype=torch.int64, device=labels.device)
    check(nat.load().pm_label_moments(ptr(labels), nz, ny, nx, n_labels, ptr(counts), ptr(sums), nat.stream_ptr()))
    if int(counts[0]) != 0:
        raise IndexError("label image has values outside [0, %d)" % n_labels)
    return counts, sums


def shape_context(xyz, centroid3, x0_3, mean_dist1, n_frames, row0=0, nrows=None, want_counts=False, want_hist=True, path="tiled"):
    """-> dict(hist=[F, nrows, 360] float64, counts=[F, nrows, 360] int32, totals=[F, nrows] int32, guard=int32 GPU [2] or None).
    guard (tiled path): how many (point, neighbour) pairs sit so close to a ring radius [0] / a sector plane or polar cone [1] — or
    coincide — that neither the tested agreement of the PCA axis with the reference's (1e-12) nor the reference's own rounding noise
    (np.linalg.inv behind its local coordinates) settles their bin (include/platymatch_hip.h).
    path: "tiled" (default: pm_shape_context_tiled) or "general" (pm_shape_context, one workgroup per point: the kernel the
    tiled call itself falls back to for tiles with neighbours on a sector edge) — identical outputs."""
    torch = _t()
    xyz = _cloud(xyz)
    n = xyz.shape[1]
    nrows = n - row0 if nrows is None else nrows
    if n_frames not in (2, 4) or row0 < 0 or nrows < 0 or row0 + nrows > n:
        raise ValueError("bad frame count or row block")
    c = _vec(centroid3, 3, "centroid")
    a = _vec(x0_3, 3, "x0")
    md = _vec(mean_dist1, 1, "mean_dist")
    hist = torch.empty((n_frames, nrows, NBINS), dtype=torch.float64, device=xyz.device) if want_hist else None
    counts = torch.empty((n_frames, nrows, NBINS), dtype=torch.int32, device=xyz.device) if want_counts else None
    totals = torch.empty((n_frames, nrows), dtype=torch.int32, device=xyz.device) if want_counts else None
    if not (want_hist or want_counts):
        raise ValueError("nothing requested")
    lib = nat.load()
    guard = None
    if path == "general":
        check(lib.pm_shape_context(ptr(xyz), n, row0, nrows, ptr(c), ptr(a), ptr(md), n_frames, ptr(counts), ptr(totals),
                                   ptr(hist), nat.stream_ptr()))
    elif path == "tiled":
        ws = nat.workspace(lib.pm_shape_context_workspace(nrows), xyz.device)
        guard = torch.zeros(2, dtype=torch.int32, device=xyz.device)
        check(lib.pm_shape_context_tiled(ptr(xyz), n, row0, nrows, ptr(c), ptr(a), ptr(md), n_frames, ptr(counts), ptr(totals),
                                         ptr(hist), ptr(guard), ptr(ws), ws.numel(), nat.stream_ptr()))
    else:
        raise ValueError("path must be 'tiled' or 'general'")
    return {"hist": hist, "counts": counts, "totals": totals, "guard": guard}


def shape_context_neighbors(nb, mean_dist):
    """nb: [n, 3] float64 GPU (frame coordinates) -> (hist[360] float64, counts[360] int32, total[1] int32)."""
    torch = _t()
    if not (nat.is_torch(nb) and nb.is_cuda and nb.dtype == torch.float64 and nb.dim() == 2 and nb.shape[1] == 3
            and nb.is_contiguous() and nb.shape[0] >= 1):
        raise ValueError("neighbors must be a contiguous float64 GPU tensor [n, 3]")
    hist = torch.empty(NBINS, dtype=torch.float64, device=nb.device)
    counts = torch.empty(NBINS, dtype=torch.int32, device=nb.device)
    total = torch.empty(1, dtype=torch.int32, device=nb.device)
    check(nat.load().pm_shape_context_neighbors(ptr(nb), nb.shape[0], float(mean_dist), ptr(counts), ptr(total), ptr(hist),
                                                nat.stream_ptr()))
    return hist, counts, total


def shape_context_neighbors_binned(nb, mean_dist, r_edges, cos_steps, phi_steps, n_thetabins, n_phibins):
    """get_shape_context with the caller's binning (pm_shape_context_neighbors_binned): nb [n, 3] float64 GPU; r_edges, cos_steps,
    phi_steps: float64 host arrays (estimate_transform/binning.py) -> (counts int64 host [n_bins], rows int64 host: the neighbours
    the kernel left to the host — within 2^-46 of a phi step)."""
    torch = _t()
    if not (nat.is_torch(nb) and nb.is_cuda and nb.dtype == torch.float64 and nb.dim() == 2 and nb.shape[1] == 3
            and nb.is_contiguous() and nb.shape[0] >= 1):
        raise ValueError("neighbors must be a contiguous float64 GPU tensor [n, 3]")
    r_edges, cos_steps, phi_steps = (np.ascontiguousarray(a, dtype=np.float64) for a in (r_edges, cos_steps, phi_steps))
    n_bins = int(r_edges.size) * int(n_thetabins) * int(n_phibins)
    tab = nat.to_dev(np.concatenate([r_edges, cos_steps, phi_steps, np.zeros(1)]), dev=nb.device)
    n = nb.shape[0]
    ints = torch.empty(n_bins + 2 + n, dtype=torch.int32, device=nb.device)     # counts | total | n_unsure | unsure rows
    base = tab.data_ptr()
    check(nat.load().pm_shape_context_neighbors_binned(
        ptr(nb), n, float(mean_dist), base, int(r_edges.size), base + 8 * r_edges.size, int(cos_steps.size),
        base + 8 * (r_edges.size + cos_steps.size), int(phi_steps.size), int(n_thetabins), int(n_phibins),
        ints.data_ptr(), ints.data_ptr() + 4 * n_bins, ints.data_ptr() + 4 * (n_bins + 2), ints.data_ptr() + 4 * (n_bins + 1),
        nat.stream_ptr()))
    host = ints.cpu().numpy()
    counts, n_unsure = host[:n_bins].astype(np.int64), int(host[n_bins + 1])
    return counts, np.sort(host[n_bins + 2:n_bins + 2 + n_unsure].astype(np.int64))


def _desc(x, name):
    torch = _t()
    if not (nat.is_torch(x) and x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.shape[1] == NBINS
            and x.is_contiguous() and x.shape[0] >= 1):
        raise ValueError("%s must be a contiguous float64 GPU tensor [N, 360]" % name)
    _here(x, name)
    return x


def chi2_cost(scA, scB, out=None):
    torch = _t()
    a, b = _desc(scA, "scA"), _desc(scB, "scB")
    nA, nB = a.shape[0], b.shape[0]
    if out is None:
        out = big_empty((nA, nB), torch.float64, a.device)
    elif not (out.is_cuda and out.dtype == torch.float64 and tuple(out.shape) == (nA, nB) and out.stride(1) == 1):
        raise ValueError("out must be float64 GPU [nA, nB] with unit column stride")
    check(nat.load().pm_chi2_cost(ptr(a), nA, ptr(b), nB, ptr(out), out.stride(0), nat.stream_ptr()))
    return out


def chi2_symmetry_flag(sc_m, sc_f):
    """int32 GPU tensor [1]: 0 if frames 2..4 are, bit for bit, the phi-sector permutations of frame 1 for every given
    row of both clouds, 1 otherwise.  No read-back: ranks of a sharded run max-reduce their flags before looking."""
    torch = _t()
    m = [_desc(sc_m[k], "sc_m[%d]" % k) for k in range(2)]
    f = [_desc(sc_f[k], "sc_f[%d]" % k) for k in range(4)]
    flag = torch.empty(1, dtype=torch.int32, device=sc_m.device)
    check(nat.load().pm_chi2_symmetry_check(ptr(m[0]), ptr(m[1]), m[0].shape[0], ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]),
                                            f[0].shape[0], ptr(flag), nat.stream_ptr()))
    return flag


def chi2_symmetric(sc_m, sc_f):
    """True if the permutation relation holds (one tiny kernel + a 4-byte read)."""
    return int(chi2_symmetry_flag(sc_m, sc_f).item()) == 0


# Measurement knob (tools/): False = the half-cost kernel divides every term (no term table).  Same bits either way.
TERM_TABLE = os.environ.get("PM_CHI2_TERM_TABLE", "1") != "0"


def chi2_cost8_frame1(sc_m1, sc_f1, out=None, info=None):
    """The eight matrices from the frame-1 descriptors alone ([nM, 360], [nF, 360]) by the half-cost kernel.  Only for
    descriptor sets whose frames 2..4 were verified (chi2_symmetry_flag == 0) to be permutations of frame 1.
    info (dict): receives which of the 30 (ring, theta) shells were served from the term table (synchronises)."""
    torch = _t()
    a, b = _desc(sc_m1, "sc_m1"), _desc(sc_f1, "sc_f1")
    nM, nF = a.shape[0], b.shape[0]
    out = _out8(out, nM, nF, a.device)
    lib = nat.load()
    if not TERM_TABLE:
        check(lib.pm_chi2_cost8_sym(ptr(a), nM, ptr(b), nF, ptr(out), out.stride(1), out.stride(0), nat.stream_ptr()))
        return out
    ws = _sym_workspace(lib, nM, nF, a.device)
    check(lib.pm_chi2_cost8_sym_ws(ptr(a), nM, ptr(b), nF, ptr(out), out.stride(1), out.stride(0), ptr(ws), ws.numel(), nat.stream_ptr()))
    if info is not None:
        import ctypes
        tabled, size = (ctypes.c_int32 * 30)(), ctypes.c_int32(0)
        check(lib.pm_chi2_sym_table_info(ptr(ws), ctypes.addressof(tabled), ctypes.addressof(size), nat.stream_ptr()))
        info["tabled_shells"] = [int(x) for x in tabled]
        info["table_size"] = int(size.value)
    return out


def chi2_cost8_relaxed(sc_m1, sc_f1, out=None, variant=1):
    """OPT-IN EXPERIMENT (never the default): the eight matrices from the frame-1 descriptors in relaxed float64 arithmetic —
    U = 0.5 (sum a + sum b) - 2 sum ab/(a+b), reciprocal with one Newton step, sums in any order, the twins identical
    (pm_chi2_cost8_relaxed; csrc/pm_chi2.hip: RELAX).  Every entry is within chi2_relaxed_delta() of the exact value, NOT bit-
    identical to it: an assignment taken from these matrices counts only with lsap.certify(min_eps = 2 min(N, M) delta).
    variant 0: every shell computed; 1: sparsely filled shells from a 94 x 94 term table; 2: 64 x 64 table, three waves per SIMD."""
    torch = _t()
    a, b = _desc(sc_m1, "sc_m1"), _desc(sc_f1, "sc_f1")
    nM, nF = a.shape[0], b.shape[0]
    out = _out8(out, nM, nF, a.device)
    lib = nat.load()
    ws = torch.empty(int(lib.pm_chi2_relaxed_workspace_bytes(nM, nF)), dtype=torch.uint8, device=a.device)
    check(lib.pm_chi2_cost8_relaxed(ptr(a), nM, ptr(b), nF, ptr(out), out.stride(1), out.stride(0), ptr(ws), ws.numel(), int(variant),
                                    nat.stream_ptr()))
    return out


def chi2_relaxed_delta():
    """Absolute per-entry error bound of chi2_cost8_relaxed against the exact cost (csrc/pm_chi2.hip: PM_CHI2_RELAX_DELTA)."""
    return float(nat.load().pm_chi2_relaxed_delta())


def chi2_filter4(sc_m1, sc_f1, out=None, dtype=None):
    """OPT-IN: the four pairings' cost matrices in packed float32 arithmetic -> [4, nM, nF] float64 (or float32: dtype / out's),
    every entry within chi2_filter_delta() of the exact cost (pm_chi2_filter4 / _f32).  A FILTER for the assignment solver
    (lsap.FilteredMatrix), never a result: matrix t stands for hypothesis PAIRINGS[t][0] and for its twin."""
    torch = _t()
    a, b = _desc(sc_m1, "sc_m1"), _desc(sc_f1, "sc_f1")
    nM, nF = a.shape[0], b.shape[0]
    if out is None:
        out = big_empty((4, nM, nF), dtype or torch.float64, a.device)
    if (tuple(out.shape) != (4, nM, nF) or out.dtype not in (torch.float64, torch.float32) or out.device != a.device or out.stride(2) != 1
            or out.stride(1) < nF or out.stride(0) < nM * out.stride(1)):
        raise ValueError("out must be a float64 or float32 tensor [4, nM, nF] on the descriptors' device with unit column stride")
    lib = nat.load()
    ws = torch.empty(int(lib.pm_chi2_filter_workspace_bytes(nM, nF)), dtype=torch.uint8, device=a.device)
    fn = lib.pm_chi2_filter4_f32 if out.dtype == torch.float32 else lib.pm_chi2_filter4
    check(fn(ptr(a), nM, ptr(b), nF, ptr(out), out.stride(1), out.stride(0), ptr(ws), ws.numel(), nat.stream_ptr()))
    return out


def chi2_filter_pair(sc_m1, sc_f1, pairing, out=None, dtype=None):
    """One pairing's filter matrix alone -> [nM, nF] float64 or float32 (pm_chi2_filter_pair / _f32: matrix `pairing` of
    chi2_filter4, the same values)."""
    torch = _t()
    a, b = _desc(sc_m1, "sc_m1"), _desc(sc_f1, "sc_f1")
    nM, nF = a.shape[0], b.shape[0]
    if out is None:
        out = big_empty((nM, nF), dtype or torch.float64, a.device)
    if (tuple(out.shape) != (nM, nF) or out.dtype not in (torch.float64, torch.float32) or out.device != a.device or out.stride(1) != 1
            or out.stride(0) < nF):
        raise ValueError("out must be a float64 or float32 tensor [nM, nF] on the descriptors' device with unit column stride")
    lib = nat.load()
    ws = torch.empty(int(lib.pm_chi2_filter_workspace_bytes(nM, nF)), dtype=torch.uint8, device=a.device)
    fn = lib.pm_chi2_filter_pair_f32 if out.dtype == torch.float32 else lib.pm_chi2_filter_pair
    check(fn(ptr(a), nM, ptr(b), nF, int(pairing), ptr(out), out.stride(0), ptr(ws), ws.numel(), nat.stream_ptr()))
    return out


def chi2_filter_delta():
    """Absolute per-entry error bound of chi2_filter4 against the exact cost (csrc/pm_chi2.hip: PM_CHI2_FILTER_DELTA)."""
    return float(nat.load().pm_chi2_filter_delta())


def chi2_entries(sc_m1, sc_f1, pairing, rows, cols, trusted=False):
    """Listed entries (rows[e], cols[e]) of pairing t's two EXACT matrices -> (natural-order values, rolled-order values), float64
    GPU tensors [len(rows)] carrying the bits of chi2_cost_pair's matrices (pm_chi2_entries_sym).  rows / cols: integer arrays
    (NumPy or torch) of equal length, every index in range (checked here: the kernel would answer NaN; trusted=True skips the
    check and its two read-backs — for index lists that come from the library's own kernels)."""
    torch = _t()
    a, b = _desc(sc_m1, "sc_m1"), _desc(sc_f1, "sc_f1")
    r = torch.as_tensor(np.ascontiguousarray(rows) if not nat.is_torch(rows) else rows).to(device=a.device, dtype=torch.int32).contiguous()
    c = torch.as_tensor(np.ascontiguousarray(cols) if not nat.is_torch(cols) else cols).to(device=a.device, dtype=torch.int32).contiguous()
    if r.dim() != 1 or r.shape != c.shape:
        raise ValueError("rows and cols must be one-dimensional and of equal length")
    k = int(r.numel())
    out = torch.empty((2, k), dtype=torch.float64, device=a.device)
    if k:
        if not trusted and (int(r.min()) < 0 or int(r.max()) >= a.shape[0] or int(c.min()) < 0 or int(c.max()) >= b.shape[0]):
            raise ValueError("entry index out of range")
        check(nat.load().pm_chi2_entries_sym(ptr(a), a.shape[0], ptr(b), b.shape[0], int(pairing), ptr(r), ptr(c), k, ptr(out[0]), ptr(out[1]),
                                             nat.stream_ptr()))
    return out[0], out[1]


def _sym_workspace(lib, nM, nF, device):
    """Device scratch of the half-cost kernel's term table (the integer counts behind the descriptor values; include/platymatch_hip.h)."""
    return _t().empty(int(lib.pm_chi2_sym_workspace_bytes(nM, nF)), dtype=_t().uint8, device=device)


from ._pairings import PAIRINGS  # noqa: E402  pairing t -> (hypothesis summed in natural order, its twin), widget numbering


def chi2_cost8_frame1_by_pairings(sc_m1, sc_f1, out=None):
    """chi2_cost8_frame1 as four launches, one per pairing (each writes its hypothesis and its twin into the eight-matrix
    buffer: the same bits), with an event after each -> (out [8, nM, nF], [event] * 4).  The assignment of pairing t can start
    when event t has fired, while the later pairings are still being built (lsap.solve_eight_on_device(ready=...))."""
    torch = _t()
    a, b = _desc(sc_m1, "sc_m1"), _desc(sc_f1, "sc_f1")
    nM, nF = a.shape[0], b.shape[0]
    out = _out8(out, nM, nF, a.device)
    lib = nat.load()
    ws = _sym_workspace(lib, nM, nF, a.device)          # one workspace: the launches are ordered on the stream
    events = []
    for t, (h, twin) in enumerate(PAIRINGS):
        check(lib.pm_chi2_cost_pair_sym_ws(ptr(a), nM, ptr(b), nF, t, ptr(out[h]), out.stride(1), (twin - h) * out.stride(0), ptr(ws),
                                           ws.numel(), nat.stream_ptr()))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(a.device))
        events.append(ev)
    return out, events


def chi2_cost_pair_into(sc_m1, sc_f1, pairing, out8):
    """The exact matrices of pairing t (hypothesis PAIRINGS[t][0] and its twin) written INTO their places of an eight-matrix
    buffer out8 [8, nM, nF] (pm_chi2_cost_pair_sym_ws with the twin's offset as matrix stride): what the relaxed mode falls back
    to for a pairing whose relaxed matrices do not certify (lsap.solve_eight_on_device(exact_rebuild=...))."""
    a, b = _desc(sc_m1, "sc_m1"), _desc(sc_f1, "sc_f1")
    nM, nF = a.shape[0], b.shape[0]
    out8 = _out8(out8, nM, nF, a.device)
    t = int(pairing)
    h, twin = PAIRINGS[t]
    lib = nat.load()
    ws = _sym_workspace(lib, nM, nF, a.device)
    check(lib.pm_chi2_cost_pair_sym_ws(ptr(a), nM, ptr(b), nF, t, ptr(out8[h]), out8.stride(1), (twin - h) * out8.stride(0), ptr(ws),
                                       ws.numel(), nat.stream_ptr()))
    return out8


def chi2_cost_pair(sc_m, sc_f, pairing, symmetric, out=None):
    """The two cost matrices of pairing t (PAIRINGS[t]: U11/U22, U12/U21, U13/U24, U14/U23) alone -> [2, nM, nF]: what a cloud
    whose eight matrices do not fit in HBM together is assigned from, two at a time.  symmetric (the frame-permutation
    relation was verified: chi2_symmetric): one launch of the half-cost kernel restricted to that pairing, sc_f may hold
    frame 1 only; otherwise two launches of the general kernel on the frames themselves.  Same bits as chi2_cost8's rows."""
    torch = _t()
    t = int(pairing)
    if t not in (0, 1, 2, 3):
        raise ValueError("pairing must be 0..3")
    nM, nF = sc_m.shape[1], sc_f.shape[1]
    if out is None:
        out = big_empty((2, nM, nF), torch.float64, sc_m.device)
    elif not (out.is_cuda and out.dtype == torch.float64 and tuple(out.shape) == (2, nM, nF) and out.is_contiguous()):
        raise ValueError("out must be a contiguous float64 GPU tensor [2, nM, nF]")
    if symmetric:
        a, b = _desc(sc_m[0], "sc_m[0]"), _desc(sc_f[0], "sc_f[0]")
        lib = nat.load()
        ws = _sym_workspace(lib, nM, nF, a.device)
        check(lib.pm_chi2_cost_pair_sym_ws(ptr(a), nM, ptr(b), nF, t, ptr(out), out.stride(1), out.stride(0), ptr(ws), ws.numel(),
                                           nat.stream_ptr()))
    else:
        if sc_f.shape[0] != 4 or sc_m.shape[0] != 2:
            raise ValueError("the general path needs all frames: sc_m [2, nM, 360], sc_f [4, nF, 360]")
        twin_fixed = (1, 0, 3, 2)[t]                     # U22, U21, U24, U23: moving frame 2 against fixed frame 2, 1, 4, 3
        chi2_cost(sc_m[0], sc_f[t], out=out[0])
        chi2_cost(sc_m[1], sc_f[twin_fixed], out=out[1])
    return out


def _out8(out, nM, nF, device):
    torch = _t()
    if out is None:
        return big_empty((8, nM, nF), torch.float64, device)
    if not (out.is_cuda and out.dtype == torch.float64 and tuple(out.shape) == (8, nM, nF) and out.stride(2) == 1
            and out.stride(0) >= nM * out.stride(1) and out.stride(1) >= nF):
        raise ValueError("out must be float64 GPU [8, nM, nF] with unit column stride")
    return out


def chi2_cost8(sc_m, sc_f, out=None, path="auto"):
    """sc_m: [2, nM, 360], sc_f: [4, nF, 360] -> out [8, nM, nF] in the widget's order 11..14, 21..24.
    path: 'auto' (verify the frame-permutation relation on the device, then take the half-cost kernel if it
    holds), 'general' (never assume it), 'symmetric' (caller has verified it) or 'symmetric-computed' (the same without the
    term table).  All paths give identical bits."""
    torch = _t()
    if not (nat.is_torch(sc_m) and sc_m.dim() == 3 and sc_m.shape[0] == 2 and nat.is_torch(sc_f) and sc_f.dim() == 3
            and sc_f.shape[0] == 4):
        raise ValueError("sc_m must be [2, nM, 360] and sc_f [4, nF, 360]")
    m = [_desc(sc_m[k], "sc_m[%d]" % k) for k in range(2)]
    f = [_desc(sc_f[k], "sc_f[%d]" % k) for k in range(4)]
    nM, nF = m[0].shape[0], f[0].shape[0]
    out = _out8(out, nM, nF, sc_m.device)
    if path not in ("auto", "general", "symmetric", "symmetric-computed"):
        raise ValueError("path must be 'auto', 'general', 'symmetric' or 'symmetric-computed'")
    sym = path.startswith("symmetric") or (path == "auto" and chi2_symmetric(sc_m, sc_f))
    if sym and path == "symmetric-computed":       # the half-cost kernel without its term table (every term divided out): a yardstick
        check(nat.load().pm_chi2_cost8_sym(ptr(m[0]), nM, ptr(f[0]), nF, ptr(out), out.stride(1), out.stride(0), nat.stream_ptr()))
    elif sym:
        chi2_cost8_frame1(m[0], f[0], out=out)
    else:
        check(nat.load().pm_chi2_cost8(ptr(m[0]), ptr(m[1]), nM, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), nF, ptr(out),
                                       out.stride(1), out.stride(0), nat.stream_ptr()))
    return out


def _pairs(mov, fix, rows, cols, trusted=False):
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    if (rows is None) != (cols is None):
        raise ValueError("rows and cols go together")
    if rows is None:
        if mov.shape[1] != fix.shape[1]:
            raise ValueError("identity pairing needs clouds of equal size")
        n = mov.shape[1]
    else:
        n = rows.numel()
        rows = _idx(rows, n, mov.shape[1], "rows", trusted=trusted)
        cols = _idx(cols, n, fix.shape[1], "cols", trusted=trusted)
    return mov, fix, rows, cols, n


def ransac_affine(mov, fix, rows, cols, samples, error):
    """samples: [trials, k] int32 GPU, k >= 4 -> (A [trials, 4, 4] float64, inliers [trials] int32, degenerate [trials] int32).
    Trials whose sample is (nearly) rank deficient come back flagged, with A = NaN and 0 inliers: refit those with pinv."""
    torch = _t()
    mov, fix, rows, cols, n = _pairs(mov, fix, rows, cols)
    if not (nat.is_torch(samples) and samples.dim() == 2 and samples.shape[1] >= 4):
        raise ValueError("samples must be [trials, k] with k >= 4 (fewer pairs are rank deficient by construction: host pinv)")
    trials, k = samples.shape
    if k > n:
        raise ValueError("more samples per trial than matched pairs")
    samples = _idx(samples.reshape(-1), trials * k, n, "samples")
    A = torch.empty((trials, 4, 4), dtype=torch.float64, device=mov.device)
    inl = torch.empty(trials, dtype=torch.int32, device=mov.device)
    deg = torch.empty(trials, dtype=torch.int32, device=mov.device)
    check(nat.load().pm_ransac_affine(ptr(mov), mov.shape[1], ptr(fix), fix.shape[1], ptr(rows), ptr(cols), n, ptr(samples),
                                      k, trials, float(error), ptr(A), ptr(inl), ptr(deg), nat.stream_ptr()))
    return A, inl, deg


def ransac_draw(n, k, trials, seed, run=0, device=None):
    """[trials, k] int32 GPU: k distinct indices of range(n) per trial from the device sampler (Philox-4x32-10 keyed by the
    64-bit `seed`, counter (trial, block, run); Floyd's subset algorithm): pm_ransac_draw."""
    torch = _t()
    n, k, trials = int(n), int(k), int(trials)
    if k > n or k <= 0 or trials <= 0:
        raise ValueError("need 0 < k <= n and trials > 0")
    out = torch.empty((trials, k), dtype=torch.int32, device=nat.device(device))
    _here(out, "samples")
    check(nat.load().pm_ransac_draw(n, k, trials, int(seed) & (2 ** 64 - 1), int(run) & 0xffffffff, ptr(out), nat.stream_ptr()))
    return out


def ransac_affine_draw(mov, fix, rows, cols, k, trials, seed, run, error, trusted=False):
    """ransac_affine with each trial's index set drawn on the device in front of its fit (pm_ransac_affine_draw)
    -> (samples [trials, k] int32, A [trials, 4, 4], inliers [trials] int32, degenerate [trials] int32).
    trusted: rows / cols were range-checked by the caller on the host (no read-back here: the launch does not wait for the stream)."""
    torch = _t()
    mov, fix, rows, cols, n = _pairs(mov, fix, rows, cols, trusted=trusted)
    k, trials = int(k), int(trials)
    if k < 4:
        raise ValueError("k >= 4 (fewer pairs are rank deficient by construction: host pinv)")
    if k > n:
        raise ValueError("more samples per trial than matched pairs")
    samples = torch.empty((trials, k), dtype=torch.int32, device=mov.device)
    A = torch.empty((trials, 4, 4), dtype=torch.float64, device=mov.device)
    inl = torch.empty(trials, dtype=torch.int32, device=mov.device)
    deg = torch.empty(trials, dtype=torch.int32, device=mov.device)
    check(nat.load().pm_ransac_affine_draw(ptr(mov), mov.shape[1], ptr(fix), fix.shape[1], ptr(rows), ptr(cols), n, k, trials,
                                           int(seed) & (2 ** 64 - 1), int(run) & 0xffffffff, float(error), ptr(samples), ptr(A),
                                           ptr(inl), ptr(deg), nat.stream_ptr()))
    return samples, A, inl, deg


def ransac_score(mov, fix, rows, cols, A, error):
    torch = _t()
    mov, fix, rows, cols, n = _pairs(mov, fix, rows, cols)
    if not (nat.is_torch(A) and A.is_cuda and A.dtype == torch.float64 and A.dim() == 3 and tuple(A.shape[1:]) == (4, 4)
            and A.is_contiguous() and A.shape[0] >= 1):
        raise ValueError("A must be a contiguous float64 GPU tensor [trials, 4, 4]")
    inl = torch.empty(A.shape[0], dtype=torch.int32, device=mov.device)
    check(nat.load().pm_ransac_score(ptr(mov), mov.shape[1], ptr(fix), fix.shape[1], ptr(rows), ptr(cols), n, ptr(A),
                                     A.shape[0], float(error), ptr(inl), nat.stream_ptr()))
    return inl


def apply_affine(A, xyz, out=None):
    xyz = _cloud(xyz)
    A = _vec(A, 16, "A")
    out = _t().empty_like(xyz) if out is None else _cloud(out, "out")
    if out.shape != xyz.shape:
        raise ValueError("out shape mismatch")
    check(nat.load().pm_apply_affine(ptr(A), ptr(xyz), xyz.shape[1], ptr(out), nat.stream_ptr()))
    return out


def fit_affine(mov, fix, nn=None, status=None):
    """-> A [4, 4].  status: optional int32 GPU tensor [1], set to 1 if the moving points are (nearly) coplanar (the
    caller must then use pinv, as the reference does), else 0."""
    torch = _t()
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    n = mov.shape[1]
    if nn is None:
        if fix.shape[1] != n:
            raise ValueError("moving and fixed must pair up one to one")
    else:
        nn = _idx(nn, n, fix.shape[1], "nn")
    lib = nat.load()
    ws = nat.workspace(lib.pm_fit_affine_workspace(n), mov.device)
    A = torch.empty((4, 4), dtype=torch.float64, device=mov.device)
    check(lib.pm_fit_affine(ptr(mov), n, ptr(fix), fix.shape[1], ptr(nn), ptr(A), ptr(_status(status)), ptr(ws), ws.numel(),
                            nat.stream_ptr()))
    return A


def _status(status):
    torch = _t()
    if status is not None and not (nat.is_torch(status) and status.is_cuda and status.dtype == torch.int32 and status.numel() == 1):
        raise ValueError("status must be an int32 GPU tensor with one element")
    return status


def icp_grid(fix):
    """Bin a fixed cloud for repeated nearest-neighbour queries -> opaque workspace tensor (valid while `fix` is unchanged)."""
    fix = _cloud(fix, "fixed")
    lib = nat.load()
    ws = nat.workspace(lib.pm_icp_grid_workspace(fix.shape[1]), fix.device)
    check(lib.pm_icp_grid_build(ptr(fix), fix.shape[1], ptr(ws), ws.numel(), nat.stream_ptr()))
    return ws


def icp_nn(mov, fix, want_dist=True, grid=None, brute=False):
    """Nearest fixed point of every moving point -> (nn [n] int32, dist [n] float64 or None).
    grid: workspace from icp_grid(fix) to skip re-binning; brute=True evaluates all N*M pairs (same answer)."""
    torch = _t()
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    n, m = mov.shape[1], fix.shape[1]
    lib = nat.load()
    nn = torch.empty(n, dtype=torch.int32, device=mov.device)
    dist = torch.empty(n, dtype=torch.float64, device=mov.device) if want_dist else None
    if brute:
        ws = nat.workspace(lib.pm_icp_nn_brute_workspace(n, m), mov.device)
        check(lib.pm_icp_nn_brute(ptr(mov), n, ptr(fix), m, ptr(nn), ptr(dist), ptr(ws), ws.numel(), nat.stream_ptr()))
    elif grid is not None:
        if not (nat.is_torch(grid) and grid.is_cuda and grid.numel() >= lib.pm_icp_grid_workspace(m)):
            raise ValueError("grid must be the tensor icp_grid(fix) returned for this fixed cloud")
        check(lib.pm_icp_grid_nn(ptr(mov), n, m, ptr(grid), grid.numel(), ptr(nn), ptr(dist), nat.stream_ptr()))
    else:
        ws = nat.workspace(lib.pm_icp_nn_workspace(n, m), mov.device)
        check(lib.pm_icp_nn(ptr(mov), n, ptr(fix), m, ptr(nn), ptr(dist), ptr(ws), ws.numel(), nat.stream_ptr()))
    return nn, dist


def icp_accumulate(mov, fix, nn, origin6, out=None, nn_trusted=False):
    """-> sums [24] (written into `out` if given).  nn_trusted: nn comes straight from icp_nn (no range read-back)."""
    torch = _t()
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    n = mov.shape[1]
    nn = None if nn is None else _idx(nn, n, fix.shape[1], "nn", trusted=nn_trusted)
    origin6 = _vec(origin6, 6, "origin")
    lib = nat.load()
    ws = nat.workspace(lib.pm_icp_accumulate_workspace(n), mov.device)
    sums = torch.empty(ICP_NSUMS, dtype=torch.float64, device=mov.device) if out is None else _vec(out, ICP_NSUMS, "out")
    check(lib.pm_icp_accumulate(ptr(mov), n, ptr(fix), fix.shape[1], ptr(nn), ptr(origin6), ptr(sums), ptr(ws), ws.numel(),
                                nat.stream_ptr()))
    return sums


def icp_update(sums, origin6, mov, fix, nn, A_icp, parts_out=None, nn_trusted=False, status=None):
    """In place: mov <- A_est mov, A_icp <- A_est A_icp.  -> (A_est [4,4], residual_parts [2] = (sum, n)).
    status: optional int32 GPU tensor [1], set to 1 (never cleared) when the moment matrix is (nearly) singular."""
    torch = _t()
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    n = mov.shape[1]
    nn = None if nn is None else _idx(nn, n, fix.shape[1], "nn", trusted=nn_trusted)
    sums, origin6 = _vec(sums, ICP_NSUMS, "sums"), _vec(origin6, 6, "origin")
    A_icp = _vec(A_icp, 16, "A_icp")
    lib = nat.load()
    ws = nat.workspace(lib.pm_icp_update_workspace(n), mov.device)
    A_est = torch.empty((4, 4), dtype=torch.float64, device=mov.device)
    parts = torch.empty(2, dtype=torch.float64, device=mov.device) if parts_out is None else _vec(parts_out, 2, "parts_out")
    check(lib.pm_icp_update(ptr(sums), ptr(origin6), ptr(mov), n, ptr(fix), fix.shape[1], ptr(nn), ptr(A_icp), ptr(A_est),
                            ptr(parts), ptr(_status(status)), ptr(ws), ws.numel(), nat.stream_ptr()))
    return A_est, parts


def icp_apply(A_est, mov, fix, nn, A_icp, nn_trusted=False):
    """A step fitted elsewhere: mov <- A_est mov (rows 0-2), A_icp <- A_est A_icp (all four rows) -> residual parts [2]."""
    torch = _t()
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    n = mov.shape[1]
    nn = None if nn is None else _idx(nn, n, fix.shape[1], "nn", trusted=nn_trusted)
    A_est, A_icp = _vec(A_est, 16, "A_est"), _vec(A_icp, 16, "A_icp")
    lib = nat.load()
    ws = nat.workspace(lib.pm_icp_update_workspace(n), mov.device)
    parts = torch.empty(2, dtype=torch.float64, device=mov.device)
    check(lib.pm_icp_apply(ptr(A_est), ptr(mov), n, ptr(fix), fix.shape[1], ptr(nn), ptr(A_icp), ptr(parts), ptr(ws),
                           ws.numel(), nat.stream_ptr()))
    return parts


def get_error(a, b):
    a, b = _cloud(a, "moving"), _cloud(b, "fixed")
    if a.shape != b.shape:
        raise ValueError("clouds must have equal shape")
    lib = nat.load()
    ws = nat.workspace(lib.pm_get_error_workspace(a.shape[1]), a.device)
    out = _t().empty(1, dtype=_t().float64, device=a.device)
    check(lib.pm_get_error(ptr(a), ptr(b), a.shape[1], ptr(out), ptr(ws), ws.numel(), nat.stream_ptr()))
    return out


_ICP_LOCKS = {}
_ICP_LOCKS_GUARD = __import__("threading").Lock()


def _icp_lock(device):
    with _ICP_LOCKS_GUARD:
        return _ICP_LOCKS.setdefault(device.index, __import__("threading").Lock())


def icp(mov, fix, iters, want_nn=False, ws=None, status=None, one_launch=False):
    """Affine ICP loop on the device.  `mov` is updated IN PLACE.
    -> (A_icp [4,4], residuals [iters], nn_all [iters, n] or None).
    one_launch=False (default): one launch per iteration (pm_icp), nothing awaited.  one_launch=True: iterations 1 .. iters-1
    in ONE launch of persistent workgroups (pm_icp_one_launch) — identical results, measured slower on MI355X
    (estimate_transform/perform_icp.py: ONE_LAUNCH); at most one such launch may be in flight per device, so the call holds a
    per-device lock until its stream has drained (other threads' ordinary kernels overlap as before).
    status: optional int32 GPU tensor [1]: 0, or 1 if some iteration met a (nearly) planar moving cloud — the results are
    then meaningless and the loop must be rerun with pinv fits (estimate_transform.perform_icp does)."""
    torch = _t()
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    n, m = mov.shape[1], fix.shape[1]
    if iters < 0:
        raise ValueError("iterations must be >= 0")
    lib = nat.load()
    need = lib.pm_icp_workspace(n, m)
    if ws is None or ws.numel() < need:
        ws = nat.workspace(need, mov.device)
    A = torch.empty((4, 4), dtype=torch.float64, device=mov.device)
    res = torch.empty(max(iters, 1), dtype=torch.float64, device=mov.device)
    nn_all = torch.empty((max(iters, 1), n), dtype=torch.int32, device=mov.device) if want_nn else None
    if one_launch and iters >= 3:
        with _icp_lock(mov.device):
            check(lib.pm_icp_one_launch(ptr(mov), n, ptr(fix), m, iters, ptr(A), ptr(res), ptr(nn_all), ptr(_status(status)), ptr(ws),
                                        ws.numel(), nat.stream_ptr()))
            torch.cuda.current_stream(mov.device).synchronize()
    else:
        check(lib.pm_icp(ptr(mov), n, ptr(fix), m, iters, ptr(A), ptr(res), ptr(nn_all), ptr(_status(status)), ptr(ws), ws.numel(),
                         nat.stream_ptr()))
    return A, res[:iters], (nn_all[:iters] if want_nn else None)


def similar_moments(mov, fix, nn=None, mov_sequential=False, fix_sequential=True, ws=None):
    """The seventeen O(N) numbers of get_similar_transform in NumPy's arithmetic (pm_similar_moments) -> float64 GPU tensor [17]:
    com_source[3], com_target[3], Sxx .. Szz [9], D, Sp.  Pairs (mov[:, i], fix[:, nn[i]]) (nn None: one to one)."""
    torch = _t()
    mov, fix = _cloud(mov, "moving"), _cloud(fix, "fixed")
    n, m = mov.shape[1], fix.shape[1]
    if nn is None:
        if n != m:
            raise ValueError("moving and fixed must pair up one to one")
    else:
        nn = _idx(nn, n, m, "nn")
    lib = nat.load()
    need = lib.pm_similar_workspace(n)
    if ws is None or ws.numel() < need:
        ws = nat.workspace(need, mov.device)
    out = torch.empty(17, dtype=torch.float64, device=mov.device)
    check(lib.pm_similar_moments(ptr(mov), n, ptr(fix), m, ptr(nn), int(bool(mov_sequential)), int(bool(fix_sequential)), ptr(out),
                                 ptr(ws), ws.numel(), nat.stream_ptr()))
    return out


def similar_apply(A, mov, fix=None, nn=None, want_residual=True, ws=None):
    """mov <- (A . [mov; 1])[:3] IN PLACE as np.matmul rounds it; -> np.mean(np.linalg.norm(mov - fix[:, nn], axis=0)) as a GPU
    tensor [1] (None if not wanted): pm_similar_apply."""
    torch = _t()
    mov = _cloud(mov, "moving")
    A = _vec(A, 16, "A")
    n = mov.shape[1]
    res = None
    lib = nat.load()
    if want_residual:
        fix = _cloud(fix, "fixed")
        if nn is None:
            if fix.shape[1] != n:
                raise ValueError("moving and fixed must pair up one to one")
        else:
            nn = _idx(nn, n, fix.shape[1], "nn")
        need = lib.pm_similar_workspace(n)
        if ws is None or ws.numel() < need:
            ws = nat.workspace(need, mov.device)
        res = torch.empty(1, dtype=torch.float64, device=mov.device)
    check(lib.pm_similar_apply(ptr(A), ptr(mov), n, ptr(fix) if want_residual else None, fix.shape[1] if want_residual else 0,
                               ptr(nn) if want_residual else None, ptr(res), ptr(ws) if want_residual else None,
                               ws.numel() if want_residual else 0, nat.stream_ptr()))
    return res


# ---- every public wrapper runs on the device of its first GPU operand -------------------------------------------------------------
def _on_operand_device(fn):
    """The binding sets the device per call (VERDICT r04 missing #5): a napari worker thread — or any caller on a
    multi-GPU host — need not have made the tensors' device current.  No-op (one integer comparison) when it already is."""
    import functools

    @functools.wraps(fn)
    def run(*args, **kw):
        dev = None
        for a in args:
            if nat.is_torch(a) and a.is_cuda:
                dev = a.device
                break
        if dev is None:
            return fn(*args, **kw)
        cuda = _t().cuda
        if dev.index == cuda.current_device():
            return fn(*args, **kw)
        with cuda.device(dev):
            return fn(*args, **kw)
    return run


for _name, _fn in list(globals().items()):
    if not _name.startswith("_") and getattr(_fn, "__module__", None) == __name__ and type(_fn).__name__ == "function":
        globals()[_name] = _on_operand_device(_fn)
del _name, _fn

