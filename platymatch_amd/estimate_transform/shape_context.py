"""Shape-context descriptors, chi-square distance and RANSAC with the reference's names
(platymatch/estimate_transform/shape_context.py).  NumPy or torch in, same kind out.

Like the reference module (`from apply_transform import *`, shape_context.py:4) this one also
re-exports apply_affine_transform / apply_similar_transform.
"""
import numpy as np

from .. import _kernels as K
from .. import _native as nat
from .apply_transform import apply_affine_transform, apply_similar_transform  # noqa: F401
from .find_transform import get_affine_transform, get_similar_transform  # noqa: F401

HYPOTHESES = ("11", "12", "13", "14", "21", "22", "23", "24")   # widget order, _dock_widget.py:547-611

# ---- descriptors that remember where they came from ---------------------------------------------------------------
# The widget never asks for a cost matrix: it calls get_unary_distance(unary_11[i], unary_21[j]) inside eight Python
# double loops (_dock_widget.py:547-602).  For that UNCHANGED code to be fast, the arrays get_unary returns for NumPy
# input are ndarrays of a subclass whose row views unary[i] remember (descriptor set, frame, row).  The first
# get_unary_distance between rows of two descriptor sets builds the whole N x M matrix (all eight at once for a
# moving x fixed combination: one chi2_cost8 launch) and keeps it on the host; every later call is a table lookup.
# Anything else — copies, slices, arithmetic results, foreign arrays, torch tensors — takes the per-pair launch as before.
MATRIX_CACHE_BYTES = 4 << 30          # all matrices of a combination are built at once below this, one at a time above


class _DescriptorSet:
    """What one get_unary call produced: the device histograms [F, N, 360], their host copy, and the cost matrices
    already built against other sets."""
    __slots__ = ("hist", "host", "tables", "fast", "rows", "lock", "__weakref__")

    def __init__(self, hist, host):
        import threading
        import weakref
        self.hist, self.host = hist, host
        self.tables = weakref.WeakKeyDictionary()        # other set -> {(frame_a, frame_b): N x M ndarray}  (or False: do not cache)
        self.fast = {}                                   # (id(other set), frame_a, frame_b) -> that ndarray: the per-call path
        self.rows = {}                                   # frame -> list of tagged row views (made once, handed out by a[i])
        self.lock = threading.Lock()


class UnaryArray(np.ndarray):
    """(N, 360) float64 descriptors as get_unary returns them — an ordinary ndarray in every respect; `a[i]` with an
    integer i additionally carries (set, frame, row) so that get_unary_distance can look the pair up."""

    def __array_finalize__(self, obj):
        self._pm_set = None          # never inherited: only __getitem__(int) and get_unary tag an array
        self._pm_frame = -1
        self._pm_row = -1

    def __getitem__(self, idx):
        dset = self._pm_set
        if dset is not None and self._pm_row < 0 and self.ndim == 2 and isinstance(idx, (int, np.integer)):
            rows = dset.rows.get(self._pm_frame)
            if rows is None:         # the widget asks for every row N (or M) times: make the tagged views once
                rows = []
                for r in range(self.shape[0]):
                    v = super().__getitem__(r)
                    v._pm_set, v._pm_frame, v._pm_row = dset, self._pm_frame, r
                    rows.append(v)
                dset.rows[self._pm_frame] = rows
            return rows[idx]         # (IndexError for an out-of-range row, like ndarray; negative rows count from the end)
        return super().__getitem__(idx)

    def __reduce__(self):            # pickling / copying yields plain data
        return np.asarray(self).__reduce__()


def _cost_table(sa, sb, fa, fb):
    """The N x M chi-square matrix between frame fa of descriptor set sa and frame fb of set sb, built on first use.
    -> ndarray, or None if the sets may not be cached (their host arrays were modified after get_unary returned them)."""
    with sa.lock:
        tabs = sa.tables.get(sb)
        if tabs is None:
            # the host arrays are writable like the reference's: trust the device copy only if they still agree bit for bit
            same = all(np.array_equal(x.host.view(np.uint64), x.hist.cpu().numpy().view(np.uint64)) for x in (sa, sb))
            tabs = sa.tables[sb] = {} if same else False
        if tabs is False:
            return None
        t = tabs.get((fa, fb))
        if t is None:
            na, nb = sa.hist.shape[1], sb.hist.shape[1]
            if sa.hist.shape[0] == 2 and sb.hist.shape[0] == 4 and 64 * na * nb <= MATRIX_CACHE_BYTES:
                U = K.chi2_cost8(sa.hist, sb.hist).cpu().numpy()                 # the widget's eight matrices in one launch
                for h, name in enumerate(HYPOTHESES):
                    tabs[(int(name[0]) - 1, int(name[1]) - 1)] = U[h]
            else:
                if 8 * na * nb * (len(tabs) + 1) > MATRIX_CACHE_BYTES:           # large clouds: the widget fills one matrix at a time
                    tabs.clear()
                tabs[(fa, fb)] = K.chi2_cost(sa.hist[fa], sb.hist[fb]).cpu().numpy()
            t = tabs[(fa, fb)]
            # the per-call index (keyed by id: dropped again when the other set dies, so that a recycled id cannot alias)
            import weakref
            if not any(k[0] == id(sb) for k in sa.fast):
                weakref.finalize(sb, _forget, weakref.ref(sa), id(sb))
            sa.fast = {k: v for k, v in sa.fast.items() if k[0] != id(sb)}
            for (a, b), tab in tabs.items():
                sa.fast[(id(sb), a, b)] = tab
        return t


def _forget(sa_ref, other_id):
    sa = sa_ref()
    if sa is not None:
        sa.fast = {k: v for k, v in sa.fast.items() if k[0] != other_id}


def get_Y(z, x):
    """shape_context.py:6-8: unit(z x x).  Three-vector helper; the per-point frames of get_unary are
    built inside the shape-context kernel, this is kept for API compatibility."""
    y = np.cross(np.asarray(z, dtype=np.float64), np.asarray(x, dtype=np.float64))
    return y / np.linalg.norm(y)


def get_shape_context(neighbors, mean_dist, r_inner=1 / 8, r_outer=2, n_rbins=5, n_thetabins=6, n_phibins=12):
    """shape_context.py:10-42: histogram of neighbours already expressed in the local frame
    ((N-1) x 3 rows of x_, y_, z_), normalised by the number counted."""
    if (neighbors.numel() if nat.is_torch(neighbors) else np.asarray(neighbors).size) == 0:
        # no neighbour at all: the reference's loops run zero times and sc / sc.sum() is 0 / 0 in every bin (:37-41)
        sc = np.full(int(n_rbins) * int(n_thetabins) * int(n_phibins), np.nan)
        return nat.torch_mod().as_tensor(sc, device=neighbors.device) if nat.is_torch(neighbors) else sc
    nb = nat.to_dev(neighbors)
    if nb.dim() != 2 or nb.shape[1] != 3:
        raise ValueError("neighbors must be (N-1) x 3")
    if (r_inner, r_outer, n_rbins, n_thetabins, n_phibins) == (1 / 8, 2, 5, 6, 12):
        hist, _, _ = K.shape_context_neighbors(nb.contiguous(), float(mean_dist))
        return nat.like_input(hist, neighbors)
    # Any other binning (get_unary never asks for one; this function's signature allows it): the ring edges, the arccos steps and
    # the floor-division steps of the azimuth are tabulated on the host with the reference's own NumPy calls (binning.py), the
    # neighbours are binned on the device against those tables, and the few whose device atan2 lies within 2^-46 of an azimuth
    # step (neighbours exactly on a sector plane) are binned by the reference's expressions themselves.
    from . import binning as B
    B.check_arguments(n_rbins, n_thetabins, n_phibins)
    n_rbins, n_thetabins, n_phibins = int(n_rbins), int(n_thetabins), int(n_phibins)
    edges = B.r_edges(r_inner, r_outer, n_rbins)
    counts, rows = K.shape_context_neighbors_binned(nb.contiguous(), float(mean_dist), edges, B.cos_steps(n_thetabins),
                                                    B.phi_steps(n_phibins), n_thetabins, n_phibins)
    if rows.size:
        idx = B.bin_rows(nb[nat.torch_mod().as_tensor(rows, device=nb.device)].cpu().numpy(), float(mean_dist), edges, n_thetabins, n_phibins)
        np.add.at(counts, idx[idx >= 0], 1)
    sc = counts.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        sc = sc / sc.sum()                                                       # :41 (0 / 0 = NaN as in the reference)
    return nat.to_dev(sc, dev=nb.device) if nat.is_torch(neighbors) else sc


def get_bin_index(r, theta, phi, r_edges, n_rbins, n_thetabins, n_phibins):
    """shape_context.py:46-58 on explicit (r, theta, phi) lists -> list of float bin indices.
    Legacy helper on host lists, kept for API compatibility: the device path never materialises
    angles (pm_binning.h bins by comparison), so nothing on the hot path calls this."""
    r, theta, phi = (np.asarray(v, dtype=np.float64) for v in (r, theta, phi))
    edges = np.asarray(r_edges, dtype=np.float64)
    below = r[:, None] < edges[None, :]
    r_index = np.where(below.any(1), below.argmax(1), n_rbins - 1).astype(np.float64)
    theta_index = theta // (np.pi / n_thetabins)
    phi_index = phi // (2 * np.pi / n_phibins)
    return list(r_index * n_thetabins * n_phibins + theta_index * n_phibins + phi_index)


def transform(detection, x_vector, y_vector, z_vector, neighbors):
    """shape_context.py:61-84: express `neighbors` ((N-1) x 3) in the frame (x, y, z) at `detection`.
    The 4 x 4 map T = B . inv(A) is formed on the host exactly as the reference does (one 4 x 4
    inverse) and applied to the points on the device."""
    det = np.asarray(nat.to_dev(detection).cpu().numpy(), dtype=np.float64).reshape(3)
    vx, vy, vz = (np.asarray(nat.to_dev(v).cpu().numpy(), dtype=np.float64).reshape(3) for v in (x_vector, y_vector, z_vector))
    A = np.ones((4, 4))
    A[0, :3], A[1, :3], A[2, :3], A[3, :3] = det, det + vx, det + vy, det + vz
    B = np.array([[0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [1, 1, 1, 1]], dtype=np.float64)
    T = np.matmul(B, np.linalg.inv(A.T))
    nb = nat.to_dev(neighbors)
    if nb.dim() != 2 or nb.shape[1] != 3:
        raise ValueError("neighbors must be (N-1) x 3")
    T[3, :] = [0, 0, 0, 1]
    out = K.apply_affine(nat.to_dev(T, dev=nb.device).reshape(16), nb.t().contiguous())
    return nat.like_input(out.t().contiguous(), neighbors)


def get_unary_distance(sc1, sc2):
    """shape_context.py:88-99 for one pair of descriptors.  Rows of the arrays get_unary returned (the widget's
    unary_11[i], unary_21[j]) are answered from the cost matrix of their two descriptor sets, built on the device at
    the first call (see UnaryArray); any other input is one launch per pair.  Whole clouds: unary_distance_matrix /
    unary_distance_matrices."""
    try:        # the per-call path of the widget's loops: two attribute reads, one dict lookup, one element
        return sc1._pm_set.fast[(id(sc2._pm_set), sc1._pm_frame, sc2._pm_frame)][sc1._pm_row, sc2._pm_row]
    except (AttributeError, KeyError, TypeError):
        pass
    sa, sb = getattr(sc1, "_pm_set", None), getattr(sc2, "_pm_set", None)
    if sa is not None and sb is not None and sc1._pm_row >= 0 and sc2._pm_row >= 0:
        t = _cost_table(sa, sb, sc1._pm_frame, sc2._pm_frame)
        if t is not None:
            return t[sc1._pm_row, sc2._pm_row]          # np.float64, as the reference's 0.5 * dist
    a, b = nat.to_dev(sc1).reshape(1, -1), nat.to_dev(sc2).reshape(1, -1)
    d = K.chi2_cost(a.contiguous(), b.contiguous())
    return d[0, 0] if nat.is_torch(sc1) else float(d.item())


def unary_distance_matrix(scA, scB):
    """U[i, j] = get_unary_distance(scA[i], scB[j]) for all pairs (_dock_widget.py:547-602), bit-identical
    to the scalar loop."""
    a, b = nat.to_dev(scA), nat.to_dev(scB)
    return nat.like_input(K.chi2_cost(a.contiguous(), b.contiguous()), scA)


def unary_distance_matrices(unary_moving, unary_fixed, out=None):
    """The widget's eight cost matrices in one launch: unary_moving = (sc, sc2) from
    get_unary(..., 'moving'), unary_fixed = (sc, sc2, sc3, sc4) from get_unary(..., 'fixed').
    -> [8, N, M] in the order U11, U12, U13, U14, U21, U22, U23, U24."""
    torch = nat.torch_mod()
    m = torch.stack([nat.to_dev(unary_moving[k]) for k in range(2)]) if not _stacked(unary_moving, 2) else unary_moving
    f = torch.stack([nat.to_dev(unary_fixed[k]) for k in range(4)]) if not _stacked(unary_fixed, 4) else unary_fixed
    U = K.chi2_cost8(m.contiguous(), f.contiguous(), out=out)
    return U if (nat.is_torch(unary_moving[0]) or out is not None) else U.cpu().numpy()


def _stacked(x, k):
    return nat.is_torch(x) and x.dim() == 3 and x.shape[0] == k and x.is_cuda


def draw_ransac_samples(n, min_samples, trials, rng=None):
    """The index sets do_ransac draws: one np.random.choice(n, min_samples, replace=False) per trial
    from NumPy's global RNG, in trial order (shape_context.py:122) — so np.random.seed(s) before a
    call reproduces the reference's sets exactly.  `rng`: a np.random.RandomState to draw from instead of
    the global one (RandomState(s) yields what np.random.seed(s) + the global generator would)."""
    src = np.random if rng is None else rng
    state = src.get_state()
    if state[0] != 'MT19937' or trials == 0 or n > 2 ** 31 - 1 or min_samples > n:
        # (min_samples > n raises inside np.random.choice exactly as in the reference)
        return np.stack([src.choice(n, min_samples, replace=False) for _ in range(trials)]).astype(np.int32)
    # Same generator outputs, consumed in C (pm_legacy_choice restates RandomState.choice -> permutation -> shuffle ->
    # random_interval on MT19937); the advanced state is handed back so the stream continues as NumPy's would.
    import ctypes
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos = ctypes.c_int(int(state[2]))
    out = np.empty((trials, int(min_samples)), dtype=np.int32)
    nat.check(nat.load().pm_legacy_choice(key.ctypes.data, ctypes.byref(pos), int(n), int(min_samples), int(trials),
                                          out.ctypes.data))
    src.set_state(('MT19937', key, pos.value, state[3], state[4]))
    return out


# Where do_ransac's index sets come from when the caller passes neither `samples` nor `device_seed`:
#   "numpy"  (default) NumPy's global generator, call for call as the reference consumes it (shape_context.py:122): after
#            np.random.seed(s) the reference's own sets, hence its own result — what the fixture tests rely on;
#   "device" the device sampler (pm_ransac_affine_draw), keyed by 64 bits taken from NumPy's global generator: for callers
#            that never seed (the widget: SURVEY.md §5) — the same distribution of results without 8 x 8 000 host shuffles.
SAMPLER = "numpy"


def fresh_device_seed(rng=None):
    """64 bits for the device sampler, taken from NumPy's global generator (or `rng`): a caller who seeded it gets
    repeatable — though not the reference's — index sets; otherwise NumPy's own entropy-seeded start-up state decides."""
    src = np.random if rng is None else rng
    lo, hi = (int(v) for v in src.randint(0, 2 ** 32, size=2, dtype=np.uint64))
    return (hi << 32) | lo


def refit_affine_winner(deferred):
    """The model of a RANSAC winner by the reference's own expression, [fixed; 1] . pinv([moving; 1]) of its sample
    (find_transform.py:11-17), on the host.  deferred: what do_ransac(defer=...) left behind (the sample's 2 k points, on the
    device) -> 4 x 4 GPU tensor."""
    from .find_transform import affine_pinv_host
    torch = nat.torch_mod()
    pts, k = deferred["pts"], deferred["k"]
    h = pts.cpu().numpy()
    return torch.as_tensor(affine_pinv_host(h[:, :k], h[:, k:]), device=pts.device)


def ransac_prelaunch(moving_all, fixed_all, rows, cols, min_samples, trials, error, device_seed, run):
    """Enqueue do_ransac's fused draw + fit + score launch for one hypothesis WITHOUT waiting for it (Affine, device sampler,
    min_samples >= 4) -> an opaque handle for do_ransac(prelaunched=...).  The driver enqueues all eight hypotheses' launches
    back to back and only then starts reading results: the GPU runs them without the host's read-backs in between."""
    torch = nat.torch_mod()
    m, f = nat.to_dev(moving_all), nat.to_dev(fixed_all)
    m, f = m[:3, :].contiguous(), f[:3, :].contiguous()
    r_h, c_h = np.asarray(rows), np.asarray(cols)         # range-checked HERE, on the host: the launch must not wait for the stream
    if r_h.size and (r_h.min() < 0 or r_h.max() >= m.shape[1] or c_h.min() < 0 or c_h.max() >= f.shape[1]):
        raise IndexError("matched pair indices outside the clouds")
    rows = nat.to_dev(r_h, dtype=torch.int32, dev=m.device)
    cols = nat.to_dev(c_h, dtype=torch.int32, dev=m.device)
    fused = K.ransac_affine_draw(m, f, rows, cols, int(min_samples), int(trials), device_seed, run, float(error), trusted=True)
    return {"fused": fused, "rows": rows, "cols": cols, "key": (int(min_samples), int(trials), float(error), int(device_seed), int(run))}


def do_ransac(moving_all, fixed_all, min_samples=4, trials=500, error=5, transform='Affine', rows=None, cols=None,
              samples=None, device_seed=None, run=0, defer=None, prelaunched=None):
    """shape_context.py:103-139 -> (A_best 4 x 4, inliers_best).

    The host draws the index sets (same RNG calls as the reference); one kernel launch fits and
    scores every trial (any min_samples >= 4; samples that are rank deficient — and every sample when
    min_samples < 4 — get the reference's pinv fit on the host and are scored on the device).
    The first trial with strictly more inliers than all before it wins;
    with no inliers at all A_best stays np.ones((4, 4)), as in the reference (:119-120, 136-138).
    `rows`/`cols` (optional) select matched pairs without gathering on the host:
    pairs are (moving_all[:, rows[k]], fixed_all[:, cols[k]]).  `samples` (optional, [trials, min_samples] int32):
    index sets already drawn with draw_ransac_samples (pipeline.estimate_transform draws them ahead of time).
    `device_seed` (optional, 64-bit int): draw the sets on the device instead (Philox stream `run` of that seed; see SAMPLER).
    `defer` (optional dict, 'Affine' only): the winner's host refit (below) is left to the caller — the dict receives what
    refit_affine_winner needs and the device's own fit of the winner is returned (pipeline.estimate_transform refits only the
    hypothesis it goes on with: one read-back instead of eight)."""
    torch = nat.torch_mod()
    m, f = nat.to_dev(moving_all), nat.to_dev(fixed_all)
    if m.dim() != 2 or f.dim() != 2:
        raise ValueError("clouds must be 3 x N or 4 x N")
    if m.shape[0] == 4 or f.shape[0] == 4:
        m, f = m[:3, :], f[:3, :]
    m, f = m.contiguous(), f.contiguous()
    if prelaunched is not None:               # the launch is already in flight (ransac_prelaunch): same arguments, checked
        if transform != 'Affine' or prelaunched["key"] != (int(min_samples), int(trials), float(error), int(device_seed), int(run)):
            raise ValueError("prelaunched does not belong to this call")
        rows, cols = prelaunched["rows"], prelaunched["cols"]
    if rows is not None:
        rows = nat.to_dev(rows, dtype=torch.int32, dev=m.device)
        cols = nat.to_dev(cols, dtype=torch.int32, dev=m.device)
        n = rows.numel()
    else:
        n = f.shape[1]
    trials = int(trials)
    ones = np.ones((4, 4))
    if trials <= 0:
        return (torch.as_tensor(ones, device=m.device) if nat.is_torch(moving_all) else ones), 0
    if transform not in ('Affine', 'Similar'):
        raise ValueError("transform must be 'Affine' or 'Similar'")
    if samples is None and device_seed is None and SAMPLER == "device" and int(min_samples) <= n:
        device_seed = fresh_device_seed()
    fused = None
    if samples is not None:
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        if samples.shape != (trials, int(min_samples)):
            raise ValueError("samples must be [trials, min_samples]")
    elif device_seed is None or int(min_samples) > n:       # (min_samples > n raises inside np.random.choice, as in the reference)
        samples = draw_ransac_samples(n, int(min_samples), trials)
    elif transform == 'Affine' and int(min_samples) >= 4:
        # the draw is fused in front of each trial's fit; the sets stay on the device (fetched only for trials the host refits)
        fused = prelaunched["fused"] if prelaunched is not None else K.ransac_affine_draw(m, f, rows, cols, int(min_samples), trials, device_seed, run, float(error))
        samples = _DeviceSamples(fused[0])
    else:
        samples = K.ransac_draw(n, int(min_samples), trials, device_seed, run, device=m.device).cpu().numpy()
    if transform == 'Affine':
        from .find_transform import affine_pinv_host, affine_pinv_host_batch
        k = int(min_samples)
        hosts = {}

        def host_pairs():                 # matched clouds on the host, fetched only if some trial needs a pinv fit
            if not hosts:
                mh, fh = m.cpu().numpy(), f.cpu().numpy()
                if rows is not None:
                    mh, fh = mh[:, rows.cpu().numpy()], fh[:, cols.cpu().numpy()]
                hosts["m"], hosts["f"] = mh, fh
            return hosts["m"], hosts["f"]

        if fused is not None:
            _, A, inl, deg = fused
            redo = np.flatnonzero(deg.cpu().numpy())
        elif k >= 4:
            A, inl, deg = K.ransac_affine(m, f, rows, cols, nat.to_dev(samples, dtype=torch.int32, dev=m.device), float(error))
            redo = np.flatnonzero(deg.cpu().numpy())
        else:                             # fewer than four pairs: rank deficient by construction, every fit is pinv's
            A = torch.empty((trials, 4, 4), dtype=torch.float64, device=m.device)
            inl = torch.zeros(trials, dtype=torch.int32, device=m.device)
            redo = np.arange(trials)
        if redo.size:
            # (nearly) coplanar / repeated sample points: the reference's pinv answer (find_transform.py:17), fitted on the
            # host for just those trials and scored on the device like the others
            mh, fh = host_pairs()
            sel = samples[redo]
            A_h = affine_pinv_host_batch(np.moveaxis(mh[:, sel], 0, 1), np.moveaxis(fh[:, sel], 0, 1))
            A_d = nat.to_dev(A_h, dev=m.device)
            idx = torch.as_tensor(redo, device=m.device)
            A[idx] = A_d
            inl[idx] = K.ransac_score(m, f, rows, cols, A_d, float(error))
        inl_h = inl.cpu().numpy()
        best = int(np.argmax(inl_h))          # first maximum == first strictly-better trial
        if inl_h[best] <= 0:
            return (torch.as_tensor(ones, device=m.device) if nat.is_torch(moving_all) else ones), 0
        # The winning trial's model by the reference's own expression, [fixed; 1] . pinv([moving; 1]) of ITS sample
        # (find_transform.py:11-17), on the host: the device's fit equals it to ~1e-12, but ICP starts from this matrix applied to
        # the moving cloud, and on lattice-like data the last bit of a moved point can decide its first correspondence
        # (tests/probes/soak_parity.py).  Only the sample's 2 k points travel.
        if isinstance(samples, _DeviceSamples):
            sel = samples.dev[best].long()
        else:
            sel = torch.as_tensor(np.asarray(samples[best], dtype=np.int64), device=m.device)
        mi, fi = (rows[sel].long(), cols[sel].long()) if rows is not None else (sel, sel)
        pts = torch.cat((m[:, mi], f[:, fi]), dim=1)
        if defer is not None:
            defer.update(pts=pts, k=k)
            A_best = A[best]
        else:
            A_best = refit_affine_winner({"pts": pts, "k": k})
        return (A_best if nat.is_torch(moving_all) else A_best.cpu().numpy()), int(inl_h[best])
    elif transform == 'Similar':
        mh, fh = m.cpu().numpy(), f.cpu().numpy()
        if rows is not None:
            mh, fh = mh[:, rows.cpu().numpy()], fh[:, cols.cpu().numpy()]
        from .find_transform import similar_fit_batch, similar_transform_host
        A_h = similar_fit_batch(np.moveaxis(mh[:, samples], 0, 1), np.moveaxis(fh[:, samples], 0, 1))     # [trials, 3, k] each
        A = nat.to_dev(A_h, dev=m.device)
        inl = K.ransac_score(m, f, rows, cols, A, float(error))
        inl_h = inl.cpu().numpy()
        best = int(np.argmax(inl_h))
        if inl_h[best] <= 0:
            return (torch.as_tensor(ones, device=m.device) if nat.is_torch(moving_all) else ones), 0
        # the winning trial refitted by the reference's own call sequence (the batch agrees with it to rounding, but
        # what follows in this mode amplifies the last bit: find_transform.similar_transform_host)
        s = samples[best]
        A_best = similar_transform_host(mh[:, s], fh[:, s])
        return (torch.as_tensor(A_best, device=m.device) if nat.is_torch(moving_all) else A_best), int(inl_h[best])


class _DeviceSamples:
    """Index sets that were drawn on the device and stay there: rows are fetched only when the host needs them (the pinv
    refit of flagged trials, the winner's set)."""

    def __init__(self, dev):
        self.dev = dev

    def __getitem__(self, idx):
        torch = nat.torch_mod()
        if isinstance(idx, (int, np.integer)):
            return self.dev[int(idx)].cpu().numpy()
        return self.dev[torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.dev.device)].cpu().numpy()


def pca_view(detections, transposed=False):
    """The N x 3 array the reference hands to sklearn's PCA (shape_context.py:151-165), made from the CALLER'S OWN array by the
    reference's own steps — transpose unless `transposed`, drop a 4th column — as views: the array's memory layout is part of what
    BLAS sees and hence of the axis's last bits.  Torch tensors come over as host arrays; non-float64 input is converted
    (sklearn itself would keep a float32 cloud in float32: not mirrored)."""
    a = detections.detach().cpu().numpy() if nat.is_torch(detections) else np.asarray(detections)
    if a.ndim != 2:
        raise ValueError("detections must be 2-D")
    if not transposed:
        a = a.transpose()
    if a.shape[1] == 4:
        a = a[:, :3]
    if a.shape[1] != 3:
        raise ValueError("detections must be 3 x N (N x 3 with transposed=True)")
    return a if a.dtype == np.float64 else a.astype(np.float64)


def pca_components_host(X):
    """sklearn.decomposition.PCA(n_components=3).fit(X).components_ (3 x 3; shape_context.py:162-165 uses row 0, the widget's
    PCA-only alignment, _dock_widget.py:722-731, all three) restated in NumPy, call for call as scikit-learn 1.7 computes it
    (sklearn/decomposition/_pca.py: _fit_full; svd_solver='auto'): for n_samples >= 10 n_features the eigen-decomposition of the
    Gram matrix X.T @ X minus n mean mean^T (solver 'covariance_eigh'), otherwise LAPACK's SVD of the centred data ('full');
    every component flipped so that its largest-magnitude entry is positive (svd_flip, u_based_decision=False).  HOST code on
    purpose (round 4): the result hangs on BLAS's accumulation order inside X.T @ X and on LAPACK's eigh — only the same NumPy
    calls on the same array reproduce the reference's axes to the BIT (verified against sklearn itself on 3 500 random clouds of
    4 .. 3 000 points, both solver branches and both memory orders: tests/golden/gen_pca_axis.py); the device kernels
    (pm_pca_axis, pm_pca_components) get within 1e-13 .. 1e-11.  The work is O(N) on 24 N bytes."""
    X = np.asarray(X)
    n, f = X.shape
    mean = np.mean(X, axis=0)
    if f <= 1000 and n >= 10 * f:
        C = X.T @ X
        C -= n * np.reshape(mean, (-1, 1)) * np.reshape(mean, (1, -1))
        C /= n - 1
        w, V = np.linalg.eigh(C)
        Vt = np.flip(np.asarray(V), axis=1).T              # rows by decreasing eigenvalue (eigh returns them ascending)
    else:
        _, _, Vt = np.linalg.svd(X - mean, full_matrices=False)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), np.argmax(np.abs(Vt), axis=1)])
    return Vt * signs[:, None]


def pca_axis_host(X):
    """The first principal axis, PCA(n_components=3).fit(X).components_[0]: what get_unary orients every local frame by."""
    return pca_components_host(X)[0]


def get_unary(centroid, mean_distance, detections, type, transposed=False, x0=None):
    """shape_context.py:144-188 -> (sc, sc2, sc3, sc4), each (N, 360) float64; sc3 and sc4 are empty
    (shape (0,)) unless type == 'fixed', as in the reference.

    centroid: 3 x 1 (or 1 x 3 with transposed=True); detections: 3 x N (or N x 3); a 4th
    row/column is dropped (:156-157).  The first PCA axis the reference gets from sklearn (:162-165) is sklearn's own sequence
    of NumPy calls on the caller's array (pca_axis_host: the reference's bits) unless `x0` is given."""
    torch = nat.torch_mod()
    d = nat.to_dev(detections)
    if d.dim() != 2:
        raise ValueError("detections must be 2-D")
    if transposed:
        d = d.t()
    if d.shape[0] not in (3, 4):
        raise ValueError("detections must be 3 x N (N x 3 with transposed=True)")
    xyz = d[:3, :].contiguous()
    c = nat.to_dev(centroid, dev=xyz.device).reshape(-1)[:3].contiguous()
    md = nat.to_dev(mean_distance, dev=xyz.device).reshape(1)
    axis = nat.to_dev(pca_axis_host(pca_view(detections, transposed)) if x0 is None else x0, dev=xyz.device).reshape(3).contiguous()
    nf = 4 if type == 'fixed' else 2
    hist = K.shape_context(xyz, c, axis, md, nf)["hist"]
    if nat.is_torch(detections):
        outs = [hist[k] for k in range(nf)]
    else:
        host = hist.cpu().numpy()
        dset = _DescriptorSet(hist, host)
        outs = []
        for k in range(nf):
            a = host[k].view(UnaryArray)
            a._pm_set, a._pm_frame = dset, k
            outs.append(a)
    if nf == 2:
        empty = torch.empty(0, dtype=torch.float64, device=xyz.device) if nat.is_torch(detections) else np.array([])
        outs += [empty, empty]
    return tuple(outs)
