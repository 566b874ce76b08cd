"""CPU restatement of the reference's estimate_transform path (the checker).

TEST INFRASTRUCTURE ONLY — never imported by platymatch_amd.  Parity status:
PINNED against outputs of the unmodified reference (tests/golden/*.npz, made by
tests/golden/gen_golden.py; checked by tests/test_oracle_golden.py).

The O(N^2) loops live in pm_oracle.c (scalar C, one rounding per operation);
this module holds the thin NumPy/SciPy parts and mirrors the reference's
function names, argument order and array conventions (3 x N float64, rows
z, y, x) so the parity tests read like the reference's own tests.
File:line citations are relative to the reference checkout.
"""
import ctypes
import os
import subprocess

import numpy as np
from scipy.optimize import linear_sum_assignment

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpm_oracle.so")
_lib = None

HYPOTHESES = ("11", "12", "13", "14", "21", "22", "23", "24")  # _dock_widget.py:547-611 order


def build(force=False):
    """Compile pm_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def set_threads(t):
    """Threads for the oracle's row loops (results do not depend on it). -> the count now in force."""
    return int(lib().pmo_set_threads(ctypes.c_int(int(t))))


def host_threads():
    """Cores this process may run on (its affinity mask, not the machine's socket count)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                   # a container's CPU quota (cgroup v2 cpu.max: "<quota> <period>" or "max <period>")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _cloud3(detections, transposed):
    """-> contiguous 3 x N (4th row/column dropped: shape_context.py:156-157, utils.py:69-70)."""
    d = np.asarray(detections, dtype=np.float64)
    if transposed:
        d = d.T
    return _f64(d[:3, :])


# --------------------------------------------------------------------------- utils/utils.py
def get_centroid(detections, transposed=True):
    """utils/utils.py:48-56."""
    d = np.asarray(detections)
    if transposed:
        return np.mean(d[:, :3], 0, keepdims=True)
    return np.mean(d[:3, :], 1, keepdims=True)


def get_mean_distance(detections, transposed=True):
    """utils/utils.py:58-75 (mean over all unordered pairs)."""
    x = _cloud3(detections, transposed)
    out = ctypes.c_double()
    lib().pmo_mean_distance(_p(x), ctypes.c_int(x.shape[1]), ctypes.byref(out))
    return np.float64(out.value)


def get_error(moving_landmarks, fixed_landmarks):
    """utils/utils.py:77-88."""
    if moving_landmarks is None and fixed_landmarks is None:
        return None
    return np.mean(np.linalg.norm(np.asarray(moving_landmarks) - np.asarray(fixed_landmarks), axis=0))


# --------------------------------------------------------------------------- apply_transform.py
def apply_affine_transform(moving, affine_transform_matrix):
    """apply_transform.py:3-17."""
    moving = np.asarray(moving)
    if moving.shape[0] == 4:
        moving = moving[:3, :]
    hom = np.vstack((moving, np.ones((1, moving.shape[1]))))
    return np.matmul(affine_transform_matrix, hom)[:3, :]


def apply_similar_transform(source, scale, rotation, translation, with_ones=False):
    """apply_transform.py:19-33."""
    if with_ones:
        source = source[:3, :]
    return scale * np.matmul(rotation, source) + translation


# --------------------------------------------------------------------------- find_transform.py
def get_affine_transform(moving, fixed, with_ones=False):
    """find_transform.py:4-17: [fixed;1] . pinv([moving;1])."""
    moving, fixed = np.asarray(moving), np.asarray(fixed)
    if not with_ones:
        ones = np.ones((1, moving.shape[1]))
        moving = np.vstack((moving, ones))
        fixed = np.vstack((fixed, ones))
    return np.matmul(fixed, np.linalg.pinv(moving))


def get_similar_transform(moving, fixed):
    """find_transform.py:21-99 (Horn's quaternion method).  Faithful to the
    reference's quirk at :60-66: after sorting, q is ROW 0 of the eigenvector
    matrix, not column 0."""
    moving, fixed = np.asarray(moving, dtype=np.float64), np.asarray(fixed, dtype=np.float64)
    ct = np.mean(fixed, 1, keepdims=True)
    cs = np.mean(moving, 1, keepdims=True)
    Y = fixed[:3, :] - ct[:3, :]
    P = moving[:3, :] - cs[:3, :]
    S = np.array([[np.sum(P[a] * Y[b]) for b in range(3)] for a in range(3)])  # S[a][b] = sum P_a Y_b  (:43-53)
    (Sxx, Sxy, Sxz), (Syx, Syy, Syz), (Szx, Szy, Szz) = S
    N = [[Sxx + Syy + Szz, Syz - Szy, -Sxz + Szx, Sxy - Syx],
         [-Szy + Syz, Sxx - Szz - Syy, Sxy + Syx, Sxz + Szx],
         [Szx - Sxz, Syx + Sxy, Syy - Szz - Sxx, Syz + Szy],
         [-Syx + Sxy, Szx + Sxz, Szy + Syz, Szz - Syy - Sxx]]
    w, V = np.linalg.eig(N)
    order = w.argsort()[::-1]
    V = V[:, order]
    q0, q1, q2, q3 = V[0]
    Qbar = [[q0, -q1, -q2, -q3], [q1, q0, q3, -q2], [q2, -q3, q0, q1], [q3, q2, -q1, q0]]
    Q = [[q0, -q1, -q2, -q3], [q1, q0, -q3, q2], [q2, q3, q0, -q1], [q3, -q2, q1, q0]]
    R = np.matmul(np.transpose(Qbar), Q)[1:, 1:]
    D = Sp = 0
    for i in range(Y.shape[1]):
        D += np.matmul(np.transpose(Y[:, i]), Y[:, i])
        Sp += np.matmul(np.transpose(P[:, i]), P[:, i])
    s = np.sqrt(D / Sp)
    t = ct[:3, :] - s * np.matmul(R, cs[:3, :])
    A = np.zeros((4, 4))
    A[:3, :3] = s * R
    A[:3, 3:4] = t
    A[3, 3] = 1
    return A


# --------------------------------------------------------------------------- shape_context.py
def pca_axis(detections_nx3):
    """First principal axis as sklearn.decomposition.PCA(3).fit(X).components_[0] gives it (shape_context.py:162-165), by
    scikit-learn 1.7's own sequence of NumPy calls (sklearn/decomposition/_pca.py: _fit_full, svd_solver='auto'): Gram matrix
    X.T @ X minus n mean mean^T and np.linalg.eigh for n >= 30 points, LAPACK's SVD of the centred data below; the component's
    largest-|.| entry made positive (svd_flip, u_based_decision=False).  Bit-identical to sklearn on the
    fixtures and on tests/golden/gen_pca_axis.py's 3 500 random clouds (round 4; rounds 1-3 used the centred covariance: 9e-14)."""
    X = np.asarray(detections_nx3, dtype=np.float64)
    n, f = X.shape
    mean = np.mean(X, axis=0)
    if f <= 1000 and n >= 10 * f:
        C = X.T @ X
        C -= n * np.reshape(mean, (-1, 1)) * np.reshape(mean, (1, -1))
        C /= n - 1
        w, V = np.linalg.eigh(C)
        Vt = np.flip(np.asarray(V), axis=1).T
    else:
        _, _, Vt = np.linalg.svd(X - mean, full_matrices=False)
    row = Vt[0]
    return row * np.sign(row[np.argmax(np.abs(row))])


def shape_context_counts(centroid, mean_distance, detections, type, transposed=False, x0=None):
    """Integer histograms behind get_unary: counts [F][N][360] int32 and totals [F][N]."""
    x = _cloud3(detections, transposed)
    n = x.shape[1]
    c = _f64(np.asarray(centroid, dtype=np.float64).reshape(-1)[:3])
    if x0 is None:
        # the array the reference hands to sklearn (shape_context.py:151-165): the CALLER's array, transposed and cut as views —
        # its memory layout reaches BLAS and decides the axis's last bits (a C- and an F-ordered copy of one cloud differ there)
        view = np.asarray(detections)
        view = view if transposed else view.transpose()
        view = view[:, :3] if view.shape[1] == 4 else view
        x0 = pca_axis(view)
    x0 = _f64(x0)
    nf = 4 if type == "fixed" else 2
    counts = np.zeros((nf, n, 360), dtype=np.int32)
    totals = np.zeros((nf, n), dtype=np.int32)
    lib().pmo_shape_context_counts(_p(x), ctypes.c_int(n), _p(c), _p(x0), ctypes.c_double(float(mean_distance)),
                                   ctypes.c_int(nf), _p(counts), _p(totals))
    return counts, totals


def shape_context_counts_rows(centroid, mean_distance, detections, type, rows, x0):
    """shape_context_counts for the listed query points only: counts [F][len(rows)][360], totals [F][len(rows)]."""
    x = _cloud3(detections, False)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    c = _f64(np.asarray(centroid, dtype=np.float64).reshape(-1)[:3])
    nf = 4 if type == "fixed" else 2
    counts = np.zeros((nf, rows.size, 360), dtype=np.int32)
    totals = np.zeros((nf, rows.size), dtype=np.int32)
    rc = lib().pmo_shape_context_rows(_p(x), ctypes.c_int(x.shape[1]), _p(rows), ctypes.c_int(rows.size), _p(c), _p(_f64(x0)),
                                      ctypes.c_double(float(mean_distance)), ctypes.c_int(nf), _p(counts), _p(totals))
    if rc != 0:
        raise IndexError("row outside the cloud")
    return counts, totals


def normalise_counts(counts, totals):
    """sc = sc / sc.sum() (shape_context.py:41); a row with nothing counted becomes NaN, as 0/0 does there."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return counts.astype(np.float64) / totals.astype(np.float64)[..., None]


def get_unary(centroid, mean_distance, detections, type, transposed=False, x0=None):
    """shape_context.py:144-188 -> (sc, sc2, sc3, sc4); sc3/sc4 are empty for type != 'fixed'."""
    counts, totals = shape_context_counts(centroid, mean_distance, detections, type, transposed, x0)
    sc = normalise_counts(counts, totals)
    if type == "fixed":
        return sc[0], sc[1], sc[2], sc[3]
    return sc[0], sc[1], np.array([]), np.array([])


def get_bin_index_direct(neighbors, mean_dist, projected=False):
    """Float bin index of each (already frame-expressed) neighbour: shape_context.py:25-36 + 46-58.  Default: the reference's
    arithmetic on an explicit list to the bit (:29 = np.linalg.norm's fused chain); projected=True: the step as
    shape_context_counts takes it on its directly projected neighbours (:29 in the plain order — pm_oracle.c: bin_index)."""
    nb = _f64(neighbors)
    out = np.empty(nb.shape[0], dtype=np.float64)
    fn = lib().pmo_bin_index_projected if projected else lib().pmo_bin_index
    fn(_p(nb), ctypes.c_int(nb.shape[0]), ctypes.c_double(float(mean_dist)), _p(out))
    return out


def get_shape_context(neighbors, mean_dist, r_inner=1 / 8, r_outer=2, n_rbins=5, n_thetabins=6, n_phibins=12):
    """shape_context.py:10-42.  The default binning (the only one get_unary ever uses) goes through the C port;
    any other parameter set is the reference's own sequence of NumPy calls, neighbour by neighbour (a Python loop: small
    cases only) — np.linalg.norm, np.arccos, np.arctan2 and the float floor divisions of get_bin_index (:46-58)."""
    if (r_inner, r_outer, n_rbins, n_thetabins, n_phibins) == (1 / 8, 2, 5, 6, 12):
        idx = get_bin_index_direct(neighbors, mean_dist)
        sc = np.zeros(360)
        for v in idx:
            if v >= 0 and v < 360 and v == np.floor(v):
                sc[int(v)] += 1
        with np.errstate(invalid="ignore", divide="ignore"):
            return sc / sc.sum()
    nb = _f64(neighbors)
    n_bins = n_rbins * n_thetabins * n_phibins
    with np.errstate(all="ignore"):
        r_edges = np.logspace(np.log10(r_inner), np.log10(r_outer), n_rbins)            # :24
        sc = np.zeros(n_bins)
        for x_, y_, z_ in nb:
            r_ = np.linalg.norm(np.array([x_, y_, z_]))                                  # :29
            r = r_ / mean_dist                                                           # :30
            theta = np.arccos(z_ / r_)                                                   # :31
            phi = np.arctan2(y_, x_)                                                     # :32-35
            if phi < 0:
                phi = 2 * np.pi + phi
            r_index = n_rbins - 1                                                        # :49
            for ind, edge in enumerate(r_edges):                                         # :53-56
                if r < edge:
                    r_index = ind
                    break
            index = r_index * n_thetabins * n_phibins + theta // (np.pi / n_thetabins) * n_phibins + phi // (2 * np.pi / n_phibins)
            if index == index and 0 <= index < n_bins and index == np.floor(index):       # :38-40: index.count(i), i = 0..n_bins-1
                sc[int(index)] += 1
        return sc / sc.sum()                                                             # :41


def unary_distance_matrix(scA, scB):
    """get_unary_distance for every (i, j) (_dock_widget.py:547-602) -> (N, M)."""
    a, b = _f64(scA), _f64(scB)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
    lib().pmo_chi2(_p(a), ctypes.c_int(a.shape[0]), _p(b), ctypes.c_int(b.shape[0]), _p(out))
    return out


def get_unary_distance(sc1, sc2):
    """shape_context.py:88-99."""
    return unary_distance_matrix(np.asarray(sc1)[None, :], np.asarray(sc2)[None, :])[0, 0]


def ransac_score(moving, fixed, A_batch, error):
    """Inlier counts of a batch of 4x4 transforms (shape_context.py:130-135)."""
    mv, fx = _f64(np.asarray(moving)[:3]), _f64(np.asarray(fixed)[:3])
    A = _f64(A_batch).reshape(-1, 16)
    inl = np.zeros(A.shape[0], dtype=np.int32)
    lib().pmo_ransac_score(_p(mv), _p(fx), ctypes.c_int(mv.shape[1]), _p(A), ctypes.c_int(A.shape[0]),
                           ctypes.c_double(float(error)), _p(inl))
    return inl


def draw_ransac_samples(n, min_samples, trials):
    """The index sets do_ransac draws, one np.random.choice per trial from the
    global RNG (shape_context.py:122)."""
    return np.stack([np.random.choice(n, min_samples, replace=False) for _ in range(trials)]).astype(np.int32)


def do_ransac(moving_all, fixed_all, min_samples=4, trials=500, error=5, transform="Affine", samples=None):
    """shape_context.py:103-139.  First strictly-better trial wins; A_best starts as ones((4,4)).
    `samples`: the index sets, if the caller has already drawn them with draw_ransac_samples (same RNG calls)."""
    moving_all, fixed_all = np.asarray(moving_all), np.asarray(fixed_all)
    if moving_all.shape[0] == 4 or fixed_all.shape[0] == 4:
        moving_all, fixed_all = moving_all[:3, :], fixed_all[:3, :]
    if samples is None:
        samples = draw_ransac_samples(fixed_all.shape[1], min_samples, trials)
    fit = get_affine_transform if transform == "Affine" else get_similar_transform
    A = np.stack([fit(moving_all[:, s], fixed_all[:, s]) for s in samples]) if trials else np.zeros((0, 4, 4))
    inl = ransac_score(moving_all, fixed_all, A, error)
    if trials == 0 or inl.max() <= 0:
        return np.ones((4, 4)), 0
    best = int(np.argmax(inl))
    return A[best], int(inl[best])


# --------------------------------------------------------------------------- perform_icp.py
def nn_argmin(moving, fixed):
    """perform_icp.py:15-16 -> (index of the nearest fixed point per moving point, that distance)."""
    mv, fx = _f64(np.asarray(moving)[:3]), _f64(np.asarray(fixed)[:3])
    idx = np.empty(mv.shape[1], dtype=np.int32)
    dist = np.empty(mv.shape[1], dtype=np.float64)
    lib().pmo_nn_argmin(_p(mv), ctypes.c_int(mv.shape[1]), _p(fx), ctypes.c_int(fx.shape[1]), _p(idx), _p(dist))
    return idx, dist


def perform_icp(moving, fixed, icp_iterations=50, transform="Affine", log=None):
    """perform_icp.py:7-26.  `log`, if a dict, receives per-iteration NN indices and residuals."""
    moving, fixed = np.asarray(moving, dtype=np.float64), np.asarray(fixed, dtype=np.float64)
    if moving.shape[0] == 4:
        moving = moving[:3, :]
    if fixed.shape[0] == 4:
        fixed = fixed[:3, :]
    A_icp = np.identity(4)
    nn_log, res_log = [], []
    for _ in range(icp_iterations):
        i2, _d = nn_argmin(moving, fixed)
        if transform == "Affine":
            A_est = get_affine_transform(moving, fixed[:, i2])
        else:
            A_est = get_similar_transform(moving, fixed[:, i2])
        moving = apply_affine_transform(moving, A_est)
        nn_log.append(i2)
        res_log.append(get_error(moving, fixed[:, i2]))
        A_icp = np.matmul(A_est, A_icp)
    if log is not None:
        log["nn"] = np.stack(nn_log) if nn_log else np.zeros((0, moving.shape[1]), np.int32)
        log["residuals"] = np.array(res_log)
    return A_icp


# --------------------------------------------------------------------------- _dock_widget.py:526-718
def estimate_transform(moving, fixed, *, transform="Affine", mode="unsupervised", ransac_samples=4,
                       ransac_trials=8000, ransac_error=16, icp_iterations=50, keypoints=None, seed=None,
                       details=None):
    """Headless restatement of EstimateTransform._click_run's shape-context branch
    (_dock_widget.py:526-718) -> (A_sc, A_icp, inliers[8]).  `seed`, if given, is passed to
    np.random.seed immediately before the eight RANSAC runs."""
    moving = _f64(np.asarray(moving)[:3])
    fixed = _f64(np.asarray(fixed)[:3])
    inliers = np.zeros(8, dtype=np.int64)
    if mode == "unsupervised":
        cm, cf = get_centroid(moving, transposed=False), get_centroid(fixed, transposed=False)      # 526-527
        mdm, mdf = get_mean_distance(moving, transposed=False), get_mean_distance(fixed, transposed=False)  # 531-532
        um = get_unary(cm, mdm, moving, "moving")                                                    # 540-542
        uf = get_unary(cf, mdf, fixed, "fixed")                                                      # 543-545
        lsa = []
        for h in HYPOTHESES:                                                                         # 547-611
            U = unary_distance_matrix(um[int(h[0]) - 1], uf[int(h[1]) - 1])
            lsa.append(linear_sum_assignment(U))
        if seed is not None:
            np.random.seed(seed)
        A_h = []
        for k, (r, c) in enumerate(lsa):                                                             # 622-675
            A, inl = do_ransac(moving[:, r], fixed[:, c], min_samples=ransac_samples, trials=ransac_trials,
                               error=ransac_error, transform=transform)
            A_h.append(A)
            inliers[k] = inl
        A_sc = A_h[int(np.argmax(inliers))]                                                          # 683-703
        if details is not None:
            details.update(lsa=lsa, ransac_A=np.stack(A_h))
    else:                                                                                            # 707-711
        kp_m, kp_f = keypoints
        A_sc = get_affine_transform(kp_m, kp_f) if transform == "Affine" else get_similar_transform(kp_m, kp_f)
    moved = apply_affine_transform(moving, A_sc)                                                     # 714
    A_icp = perform_icp(moved, fixed, icp_iterations, transform, log=details)                        # 715-717
    return A_sc, A_icp, inliers


# --------------------------------------------------------------------------- rows "next" 
def pca_components(detections_nx3):
    """sklearn.decomposition.PCA(3).fit(X).components_ (the widget's PCA-only alignment, _dock_widget.py:722-731) by scikit-learn
    1.7's own sequence of NumPy calls (see pca_axis): bit-identical to sklearn (round 4)."""
    X = np.asarray(detections_nx3, dtype=np.float64)
    n, f = X.shape
    mean = np.mean(X, axis=0)
    if f <= 1000 and n >= 10 * f:
        C = X.T @ X
        C -= n * np.reshape(mean, (-1, 1)) * np.reshape(mean, (1, -1))
        C /= n - 1
        w, V = np.linalg.eigh(C)
        Vt = np.flip(np.asarray(V), axis=1).T
    else:
        _, _, Vt = np.linalg.svd(X - mean, full_matrices=False)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), np.argmax(np.abs(Vt), axis=1)])
    return Vt * signs[:, None]


def cdist(a, b):
    """scipy.spatial.distance.cdist(a.T, b.T) — the call EvaluateMetrics makes (_dock_widget.py:1032,1038,1050)."""
    from scipy.spatial.distance import cdist as _cdist
    return _cdist(np.asarray(a).transpose(), np.asarray(b).transpose())


def calculate_metrics(moving_keypoints, moving_keypoint_ids, moving_detections, moving_ids, fixed_keypoints,
                      fixed_keypoint_ids, fixed_detections, fixed_ids, transform_matrix_1, transform_matrix_2):
    """EvaluateMetrics._calculate_metrics (_dock_widget.py:1030-1080) -> (matching accuracy, average registration error)."""
    r, c = linear_sum_assignment(cdist(moving_keypoints, moving_detections))
    moving_dictionary = {moving_keypoint_ids[i]: moving_ids[c[i]] for i in r}
    r, c = linear_sum_assignment(cdist(fixed_keypoints, fixed_detections))
    fixed_dictionary = {fixed_keypoint_ids[i]: fixed_ids[c[i]] for i in r}
    moved = apply_affine_transform(apply_affine_transform(moving_detections, transform_matrix_1), transform_matrix_2)
    row_indices, col_indices = linear_sum_assignment(cdist(moved, fixed_detections))
    row_ids, col_ids = np.asarray(moving_ids)[row_indices], np.asarray(fixed_ids)[col_indices]
    hits = 0
    for key in moving_dictionary.keys():
        if key in fixed_dictionary.keys():
            got = col_ids[np.where(row_ids == moving_dictionary[key])]
            if got.size == 1 and got[0] == fixed_dictionary[key]:
                hits += 1
    accuracy = hits / len(fixed_dictionary.keys())
    combined = np.matmul(transform_matrix_2, transform_matrix_1)
    tmk = apply_affine_transform(moving_keypoints, combined)
    distance = 0
    for i in range(tmk.shape[1]):
        distance += np.linalg.norm([np.asarray(fixed_keypoints).transpose()[np.where(np.asarray(fixed_keypoint_ids) == moving_keypoint_ids[i]), :]
                                    - tmk.transpose()[i, :]])
    return accuracy, distance / len(moving_dictionary.keys())


def label_centroids(label_image, anisotropy=1.0):
    """The widget's label-image branch (_dock_widget.py:497-521), literally: per label np.where + np.mean."""
    data = np.asarray(label_image)
    ids = np.unique(data)
    ids = ids[ids != 0]
    cents, sizes = [], []
    for i in ids:
        z, y, x = np.where(data == i)
        cents.append([np.mean(z), np.mean(y), np.mean(x)])
        sizes.append(float(anisotropy) * len(z))
    return np.asarray(cents).transpose(), np.asarray(sizes), ids


def ransac_error_from_sizes(moving_nucleus_size, fixed_nucleus_size):
    """_dock_widget.py:613-618."""
    if len(moving_nucleus_size) == 0 or len(fixed_nucleus_size) == 0:
        return 16
    return 0.5 * (np.average(moving_nucleus_size) ** (1 / 3) + np.average(fixed_nucleus_size) ** (1 / 3))
