#!/usr/bin/env python3
"""Generate golden fixtures by running the UNMODIFIED reference (juglab/PlatyMatch).

Runs only in the build container, where the reference is mounted read-only at
/root/reference.  Nothing here is used at test time: the tests read the .npz
files this script writes next to itself.  The fixtures are data only (inputs
and the reference's outputs); no reference source travels with them.

Import recipe: `platymatch/__init__.py` pulls in napari/Qt,
which are not installed, so a bare package stub with the right __path__ is
registered first, plus a stub `qtpy.QtWidgets` (utils/utils.py:3 imports
QFileDialog at module top).  The hot-path modules then import and run as is.

The widget's orchestration (`_dock_widget.py:526-718`) cannot be imported
(Qt); `run_pipeline` below drives the reference's own functions in the same
stage order with the same arguments (the reference's test-suite does the same,
`_tests/test_estimate_transform.py:11-72`).

Usage:  python tests/golden/gen_golden.py [scenario ...]
"""
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    pkg = types.ModuleType("platymatch")
    pkg.__path__ = [os.path.join(REF, "platymatch")]
    sys.modules["platymatch"] = pkg
    qtpy = types.ModuleType("qtpy")
    qtw = types.ModuleType("qtpy.QtWidgets")
    qtw.QFileDialog = type("QFileDialog", (), {})
    qtpy.QtWidgets = qtw
    sys.modules["qtpy"] = qtpy
    sys.modules["qtpy.QtWidgets"] = qtw
    from platymatch.estimate_transform import (apply_transform, find_transform,
                                               perform_icp, shape_context)
    from platymatch.utils import utils
    return shape_context, find_transform, apply_transform, perform_icp, utils


# Ground-truth affine used by the reference's own tests
# (data literal at _tests/test_estimate_transform.py:88-91 and 156-159).
A_GT = np.array([[9.08173020e-01, -2.58092254e-01, 2.21387350e-01, 4.98532315e+00],
                 [-2.85490902e-02, 5.66865806e-01, 7.60292965e-01, -2.13218259e+02],
                 [-2.53059848e-01, -7.49475117e-01, 4.48778146e-01, 5.56203489e+02],
                 [1.73472348e-17, 2.42861287e-17, -4.16333634e-17, 1.00000000e+00]])


def load_asset(name):
    """`id x y z` rows, space separated -> 3 x N in (z, y, x) order, as the
    reference's tests load it (test_estimate_transform.py:17-21)."""
    import pandas as pd
    arr = pd.read_csv(os.path.join(REF, "platymatch/_tests/assets", name),
                      header=None, delimiter=" ").to_numpy()
    return np.ascontiguousarray(np.flip(arr[:, 1:4], 1).transpose().astype(np.float64))


def counts_from_sc(sc):
    """Recover the integer histogram and its total from a normalised descriptor
    row set, and prove counts/total regenerates the reference floats bit for bit."""
    n = sc.shape[0]
    counts = np.zeros(sc.shape, dtype=np.int32)
    totals = np.zeros(n, dtype=np.int32)
    for i in range(n):
        row = sc[i]
        if np.isnan(row).any():
            totals[i] = 0
            continue
        ok = False
        for t in range(n + 2, 0, -1):  # N-1 unless neighbours were dropped
            c = np.rint(row * t)
            if c.sum() == t and np.array_equal(c / c.sum(), row):
                counts[i] = c.astype(np.int32)
                totals[i] = t
                ok = True
                break
        if not ok:
            raise RuntimeError("row %d: no integer total reproduces the descriptor" % i)
    return counts, totals


def run_pipeline(ref, moving, fixed, ransac_trials, ransac_error, icp_iters, u_row_step, tag, seed=0):
    sc_mod, ft, at, icp_mod, utils = ref
    from scipy.optimize import linear_sum_assignment
    from sklearn.decomposition import PCA
    out = {"moving": moving, "fixed": fixed}
    t0 = time.time()
    cm = utils.get_centroid(moving, transposed=False)
    cf = utils.get_centroid(fixed, transposed=False)
    mdm = utils.get_mean_distance(moving, transposed=False)
    mdf = utils.get_mean_distance(fixed, transposed=False)
    out.update(centroid_m=cm, centroid_f=cf, mean_dist_m=np.float64(mdm), mean_dist_f=np.float64(mdf))
    # the PCA axis get_unary derives internally (shape_context.py:162-165)
    out["x0_m"] = PCA(n_components=3).fit(moving.T).components_[0].copy()
    out["x0_f"] = PCA(n_components=3).fit(fixed.T).components_[0].copy()

    um = sc_mod.get_unary(cm, mean_distance=mdm, detections=moving, type="moving", transposed=False)
    uf = sc_mod.get_unary(cf, mean_distance=mdf, detections=fixed, type="fixed", transposed=False)
    assert um[2].shape == (0,) and um[3].shape == (0,)
    for k in range(2):
        c, t = counts_from_sc(um[k])
        out["counts_m%d" % (k + 1)] = c.astype(np.int16)
        out["total_m%d" % (k + 1)] = t
    for k in range(4):
        c, t = counts_from_sc(uf[k])
        out["counts_f%d" % (k + 1)] = c.astype(np.int16)
        out["total_f%d" % (k + 1)] = t
    print("[%s] descriptors %.1fs" % (tag, time.time() - t0), flush=True)

    n, m = moving.shape[1], fixed.shape[1]
    names = ["11", "12", "13", "14", "21", "22", "23", "24"]
    U = {}
    t0 = time.time()
    for nm in names:
        a = um[int(nm[0]) - 1]
        b = uf[int(nm[1]) - 1]
        mat = np.zeros((n, m))
        for i in range(n):
            for j in range(m):
                mat[i, j] = sc_mod.get_unary_distance(a[i], b[j])
        U[nm] = mat
    print("[%s] chi2 %.1fs" % (tag, time.time() - t0), flush=True)
    rows_kept = np.arange(0, n, u_row_step)
    out["U_rows"] = rows_kept
    out["U"] = np.stack([U[nm][rows_kept] for nm in names])          # 8 x R x M, exact float64
    out["U_sum"] = np.array([U[nm].sum() for nm in names])
    out["U_rowmin_idx"] = np.stack([U[nm].argmin(1) for nm in names]).astype(np.int32)

    lsa = [linear_sum_assignment(U[nm]) for nm in names]
    out["lsa_rows"] = np.stack([r for r, _ in lsa]).astype(np.int32)
    out["lsa_cols"] = np.stack([c for _, c in lsa]).astype(np.int32)

    t0 = time.time()
    np.random.seed(seed)
    A_r, inl = [], []
    for r, c in lsa:
        A, k = sc_mod.do_ransac(moving[:, r], fixed[:, c], min_samples=4, trials=ransac_trials,
                                error=ransac_error, transform="Affine")
        A_r.append(np.asarray(A, dtype=np.float64))
        inl.append(k)
    out["ransac_seed"] = np.int64(seed)
    out["ransac_trials"] = np.int64(ransac_trials)
    out["ransac_error"] = np.float64(ransac_error)
    out["ransac_A"] = np.stack(A_r)
    out["ransac_inliers"] = np.array(inl, dtype=np.int64)
    best = int(np.argmax(out["ransac_inliers"]))
    A_sc = A_r[best]
    out["best_hypothesis"] = np.int64(best)
    out["A_sc"] = A_sc
    print("[%s] ransac %.1fs inliers %s" % (tag, time.time() - t0, inl), flush=True)

    # ICP, with the reference's own distance_matrix/get_error calls observed
    nn_log, res_log = [], []
    real_dm, real_err = icp_mod.distance_matrix, icp_mod.get_error

    def dm_spy(a, b):
        d = real_dm(a, b)
        nn_log.append(np.argmin(d, 1).astype(np.int32))
        return d

    def err_spy(a, b):
        e = real_err(a, b)
        res_log.append(e)
        return e

    icp_mod.distance_matrix, icp_mod.get_error = dm_spy, err_spy
    try:
        moved = at.apply_affine_transform(moving.copy(), A_sc)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            A_icp = icp_mod.perform_icp(moved, fixed.copy(), icp_iters, "Affine")
    finally:
        icp_mod.distance_matrix, icp_mod.get_error = real_dm, real_err
    out["icp_iters"] = np.int64(icp_iters)
    out["icp_nn"] = np.stack(nn_log)
    out["icp_residuals"] = np.array(res_log)
    out["A_icp"] = A_icp
    out["A_final"] = A_icp @ A_sc
    return out


def synth_cloud(n, seed):
    rng = np.random.default_rng(seed)
    moving = rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0
    return rng, moving


def scenario(ref, name):
    at = ref[2]
    if name == "insitu02_identity":
        mv = load_asset("02-insitu.csv")
        return run_pipeline(ref, mv, at.apply_affine_transform(mv, np.identity(4)), 2000, 16, 50, 16, name), np.identity(4)
    if name == "insitu02_affine":
        mv = load_asset("02-insitu.csv")
        return run_pipeline(ref, mv, at.apply_affine_transform(mv, A_GT), 2000, 16, 50, 16, name), A_GT
    if name == "insitu04_affine":
        mv = load_asset("04-insitu.csv")
        return run_pipeline(ref, mv, at.apply_affine_transform(mv, A_GT), 2000, 16, 50, 16, name), A_GT
    if name == "synth128":
        rng, mv = synth_cloud(128, 0)
        fx = at.apply_affine_transform(mv, A_GT) + rng.normal(scale=1.0, size=mv.shape)
        fx = np.ascontiguousarray(fx[:, rng.permutation(128)])
        return run_pipeline(ref, mv, fx, 1000, 16, 30, 1, name), A_GT
    if name == "synth96x128":
        rng, mv = synth_cloud(128, 1)
        fx = at.apply_affine_transform(mv, A_GT) + rng.normal(scale=0.5, size=mv.shape)
        fx = np.ascontiguousarray(fx[:, rng.permutation(128)])
        mv = np.ascontiguousarray(mv[:, :96])
        return run_pipeline(ref, mv, fx, 500, 16, 20, 1, name), A_GT
    if name == "synth1000":
        rng, mv = synth_cloud(1000, 2)
        fx = at.apply_affine_transform(mv, A_GT) + rng.normal(scale=1.0, size=mv.shape)
        fx = np.ascontiguousarray(fx[:, rng.permutation(1000)])
        return run_pipeline(ref, mv, fx, 300, 16, 20, 50, name), A_GT
    raise KeyError(name)


def micro(ref):
    """Unit-level known answers for the binning edge cases and small helpers."""
    sc_mod, ft, at, icp_mod, utils = ref
    out = {}
    # every integer vector in {-2..2}^3: exact ring edges, theta = 0, pi/2, pi, phi on bin edges, the zero vector
    g = np.arange(-2, 3, dtype=np.float64)
    grid = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    out["grid_neighbors"] = grid
    with np.errstate(all="ignore"):
        out["grid_sc_md1"] = sc_mod.get_shape_context(grid, 1.0)
        out["grid_sc_md3"] = sc_mod.get_shape_context(grid, 3.0)
    rng = np.random.default_rng(7)
    nb = rng.normal(size=(500, 3)) * 40.0
    out["rand_neighbors"] = nb
    out["rand_sc"] = sc_mod.get_shape_context(nb, 55.0)
    # raw bin indices for the random neighbours (get_bin_index called exactly as get_shape_context does)
    r = [np.linalg.norm(v) / 55.0 for v in nb]
    th = [np.arccos(v[2] / np.linalg.norm(v)) for v in nb]
    ph = [(2 * np.pi + np.arctan2(v[1], v[0])) if np.arctan2(v[1], v[0]) < 0 else np.arctan2(v[1], v[0]) for v in nb]
    edges = np.logspace(np.log10(1 / 8), np.log10(2), 5)
    out["r_edges"] = edges
    out["rand_bin_index"] = np.array(sc_mod.get_bin_index(r, th, ph, edges, 5, 6, 12))
    # chi2 on two descriptor rows incl. equal and zero bins
    a = out["rand_sc"]
    b = sc_mod.get_shape_context(rng.normal(size=(300, 3)) * 30.0, 40.0)
    out["chi2_a"], out["chi2_b"] = a, b
    out["chi2_ab"] = np.float64(sc_mod.get_unary_distance(a, b))
    out["chi2_aa"] = np.float64(sc_mod.get_unary_distance(a, a))
    # transform fitting / application
    P = rng.normal(size=(3, 40)) * 50 + 100
    Q = at.apply_affine_transform(P, A_GT) + rng.normal(scale=0.3, size=(3, 40))
    out["fit_moving"], out["fit_fixed"] = P, Q
    out["fit_affine"] = ft.get_affine_transform(P, Q)
    out["fit_affine4"] = ft.get_affine_transform(P[:, :4], Q[:, :4])
    out["fit_similar"] = ft.get_similar_transform(P, Q)
    out["fit_similar4"] = ft.get_similar_transform(P[:, :4], Q[:, :4])
    out["apply_affine"] = at.apply_affine_transform(P, A_GT)
    out["apply_affine_4row"] = at.apply_affine_transform(np.vstack([P, np.ones((1, 40))]), A_GT)
    out["apply_similar"] = at.apply_similar_transform(P, 1.3, A_GT[:3, :3], A_GT[:3, 3:4])
    out["error_PQ"] = np.float64(utils.get_error(P, Q))
    out["centroid_T"] = utils.get_centroid(P.T, transposed=True)
    out["centroid_F"] = utils.get_centroid(P, transposed=False)
    out["mean_distance"] = np.float64(utils.get_mean_distance(P, transposed=False))
    cube = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [1, 0, 1], [1, 1, 0], [1, 1, 1], [0, 1, 1], [0, 1, 0]], dtype=np.float64)
    out["cube"] = cube
    out["cube_centroid"] = utils.get_centroid(cube, transposed=True)  # _tests/test_utils.py:5-8 expects 0.5,0.5,0.5
    # degenerate cloud: integer coordinates, symmetric about the origin (centroid is exactly a point),
    # one duplicated pair of points -> NaN row and dropped neighbours
    half = rng.integers(-60, 61, size=(3, 20)).astype(np.float64)
    deg = np.concatenate([half, -half, np.zeros((3, 1)), half[:, :1], -half[:, :1]], axis=1)
    out["degenerate_cloud"] = deg
    with np.errstate(all="ignore"):
        c = utils.get_centroid(deg, transposed=False)
        md = utils.get_mean_distance(deg, transposed=False)
        d = sc_mod.get_unary(c, md, deg, "fixed", transposed=False)
    out["degenerate_centroid"] = c
    out["degenerate_mean_dist"] = np.float64(md)
    for k in range(4):
        out["degenerate_sc%d" % (k + 1)] = d[k]
    # ICP on its own (supervised tail, _dock_widget.py:707-717)
    kp_m = P[:, :10]
    kp_f = Q[:, :10]
    out["sup_A_sc"] = ft.get_affine_transform(kp_m, kp_f)
    return out


def next_rows(ref):
    """Known answers for the SURVEY.md §8f rows: sklearn PCA components, the evaluation metrics (scipy cdist +
    linear_sum_assignment in the call order of _dock_widget.py:1030-1080 — the method itself lives in a Qt class and cannot
    be imported) and the label-image centroid loop (_dock_widget.py:497-521)."""
    sc_mod, ft, at, icp_mod, utils = ref
    from scipy.optimize import linear_sum_assignment
    from scipy.spatial.distance import cdist
    from sklearn.decomposition import PCA
    out = {}
    for tag, asset in (("02", "02-insitu.csv"), ("04", "04-insitu.csv")):
        X = load_asset(asset)
        out["pca_cloud_" + tag] = X
        out["pca_components_" + tag] = PCA(n_components=3).fit((X - utils.get_centroid(X, transposed=False)).transpose()).components_
    # evaluation metrics on a noisy synthetic pair with ids
    rng, mv = synth_cloud(220, 3)
    T1 = A_GT.copy()
    T1[3] = [0, 0, 0, 1]
    th = 0.01
    T2 = np.array([[np.cos(th), -np.sin(th), 0, 0.4], [np.sin(th), np.cos(th), 0, -0.3], [0, 0, 1, 0.2], [0, 0, 0, 1.0]])
    fx_all = at.apply_affine_transform(at.apply_affine_transform(mv, T1), T2) + rng.normal(scale=2.5, size=mv.shape)
    perm = rng.permutation(220)[:200]
    fx = np.ascontiguousarray(fx_all[:, perm])
    m_ids = np.arange(1000, 1220)
    f_ids = (np.arange(5000, 5220))[perm]
    kp_sel = rng.choice(200, 14, replace=False)
    kp_ids = np.arange(1, 15)
    m_kp = mv[:, perm[kp_sel]] + rng.normal(scale=1.0, size=(3, 14))
    f_kp = fx[:, kp_sel] + rng.normal(scale=1.0, size=(3, 14))
    f_kp_ids = kp_ids.copy()
    f_kp_ids[-2:] = [99, 98]                                   # two keypoints without a partner
    out.update(ev_moving=mv, ev_fixed=fx, ev_moving_ids=m_ids, ev_fixed_ids=f_ids, ev_moving_kp=m_kp, ev_fixed_kp=f_kp,
               ev_moving_kp_ids=kp_ids, ev_fixed_kp_ids=f_kp_ids, ev_T1=T1, ev_T2=T2)
    r, c = linear_sum_assignment(cdist(m_kp.transpose(), mv.transpose()))
    md = {kp_ids[i]: m_ids[c[i]] for i in r}
    r, c = linear_sum_assignment(cdist(f_kp.transpose(), fx.transpose()))
    fd = {f_kp_ids[i]: f_ids[c[i]] for i in r}
    moved = at.apply_affine_transform(at.apply_affine_transform(mv, T1), T2)
    cost = cdist(moved.transpose(), fx.transpose())
    ri, ci = linear_sum_assignment(cost)
    row_ids, col_ids = m_ids[ri], f_ids[ci]
    hits = 0
    for key in md.keys():
        if key in fd.keys():
            got = col_ids[np.where(row_ids == md[key])]
            if got.size == 1 and got[0] == fd[key]:
                hits += 1
    out["ev_accuracy"] = np.float64(hits / len(fd.keys()))
    tmk = at.apply_affine_transform(m_kp, np.matmul(T2, T1))
    dist = 0
    for i in range(tmk.shape[1]):
        dist += np.linalg.norm([f_kp.transpose()[np.where(f_kp_ids == kp_ids[i]), :] - tmk.transpose()[i, :]])
    out["ev_registration_error"] = np.float64(dist / len(md.keys()))
    out["ev_cdist_rows"] = cost[::20]
    out["ev_lsa_cols"] = ci.astype(np.int32)
    # label image: ellipsoids with non-contiguous ids, touching the borders, one voxel-sized label
    lab = np.zeros((40, 48, 56), dtype=np.uint16)
    zz, yy, xx = np.meshgrid(np.arange(40), np.arange(48), np.arange(56), indexing="ij")
    lid = 3
    for _ in range(30):
        c = rng.uniform([0, 0, 0], [40, 48, 56])
        rad = rng.uniform(2, 6, size=3)
        mask = ((zz - c[0]) / rad[0]) ** 2 + ((yy - c[1]) / rad[1]) ** 2 + ((xx - c[2]) / rad[2]) ** 2 <= 1
        lab[mask] = lid
        lid += int(rng.integers(1, 40))
    lab[39, 47, 55] = 60000
    out["lab_image"] = lab
    ids = np.unique(lab)
    ids = ids[ids != 0]
    cents, sizes = [], []
    for i in ids:
        z, y, x = np.where(lab == i)
        cents.append([np.mean(z), np.mean(y), np.mean(x)])
        sizes.append(float(2.0) * len(z))
    out["lab_ids"] = ids
    out["lab_centroids"] = np.asarray(cents).transpose()
    out["lab_sizes_aniso2"] = np.asarray(sizes)
    return out


def similar_mode(ref):
    """transform='Similar' end to end on a small pair: the per-trial fits of do_ransac on fancy-indexed samples (the memory
    layout the reference hands to get_similar_transform matters: np.mean reduces an F-ordered 3 x k array sequentially),
    the seeded do_ransac result, and a Similar-mode ICP."""
    sc_mod, ft, at, icp_mod, utils = ref
    rng = np.random.default_rng(21)
    n = 150
    mv = rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0
    th = 0.3
    Rz = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    fx = 1.2 * Rz @ mv + np.array([[10.0], [-20.0], [5.0]]) + rng.normal(scale=0.5, size=(3, n))
    out = {"moving": mv, "fixed": fx}
    for k in (4, 6, 9, 20):
        np.random.seed(100 + k)
        sets = np.stack([np.random.choice(n, k, replace=False) for _ in range(120)])
        out["samples_k%d" % k] = sets.astype(np.int32)
        out["fits_k%d" % k] = np.stack([ft.get_similar_transform(mv[:, s], fx[:, s]) for s in sets])
    for k, trials, err in ((4, 400, 3.0), (9, 150, 3.0)):
        np.random.seed(7)
        A, inl = sc_mod.do_ransac(mv, fx, min_samples=k, trials=trials, error=err, transform='Similar')
        out["ransac_A_k%d" % k], out["ransac_inliers_k%d" % k] = A, np.int64(inl)
        out["ransac_args_k%d" % k] = np.array([k, trials, err, 7], dtype=np.float64)
    A_icp = icp_mod.perform_icp(at.apply_affine_transform(mv, out["ransac_A_k4"]), fx, 12, 'Similar')
    out["icp_A"] = A_icp
    return out


def ransac_k(ref):
    """do_ransac with min_samples != 4 (the widget exposes the field, _dock_widget.py:327), get_affine_transform on
    rank-deficient input (pinv's minimum-norm answer, find_transform.py:17), a planar cloud through RANSAC and ICP, and
    shape_context.transform() on its own (shape_context.py:61-84)."""
    sc_mod, ft, at, icp_mod, utils = ref
    import contextlib
    import io
    rng = np.random.default_rng(33)
    n = 160
    mv = rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0
    fx = at.apply_affine_transform(mv, A_GT) + rng.normal(scale=0.8, size=(3, n))
    out = {"moving": mv, "fixed": fx}
    for k in (1, 2, 3, 5, 8, 13):
        np.random.seed(200 + k)
        sets = np.stack([np.random.choice(n, k, replace=False) for _ in range(80)])
        out["samples_k%d" % k] = sets.astype(np.int32)
        out["fits_k%d" % k] = np.stack([ft.get_affine_transform(mv[:, s], fx[:, s]) for s in sets])
    for k, trials, err in ((3, 300, 6.0), (5, 300, 3.0), (8, 200, 3.0)):
        np.random.seed(9)
        A, inl = sc_mod.do_ransac(mv, fx, min_samples=k, trials=trials, error=err, transform='Affine')
        out["ransac_A_k%d" % k], out["ransac_inliers_k%d" % k] = np.asarray(A, dtype=np.float64), np.int64(inl)
        out["ransac_args_k%d" % k] = np.array([k, trials, err, 9], dtype=np.float64)
    # planar moving cloud (one coordinate constant: 2-D data embedded in 3-D), non-planar and planar targets
    pl = mv.copy()
    pl[0, :] = 37.5
    T = A_GT.copy()
    T[3] = [0, 0, 0, 1]
    pf = at.apply_affine_transform(pl, T) + rng.normal(scale=0.5, size=(3, n))
    out["planar_moving"], out["planar_fixed"] = pl, pf
    out["planar_fit"] = ft.get_affine_transform(pl, pf)
    out["planar_fit_4"] = ft.get_affine_transform(pl[:, :4], pf[:, :4])
    out["planar_fit_3"] = ft.get_affine_transform(pl[:, :3], pf[:, :3])
    np.random.seed(4)
    A, inl = sc_mod.do_ransac(pl, pf, min_samples=4, trials=200, error=3.0, transform='Affine')
    out["planar_ransac_A"], out["planar_ransac_inliers"] = np.asarray(A, dtype=np.float64), np.int64(inl)
    start = at.apply_affine_transform(pl, A)
    nn_log = []
    real_dm = icp_mod.distance_matrix

    def dm_spy(a, b):
        d = real_dm(a, b)
        nn_log.append(np.argmin(d, 1).astype(np.int32))
        return d

    icp_mod.distance_matrix = dm_spy
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out["planar_icp_A"] = icp_mod.perform_icp(start, pf.copy(), 6, 'Affine')
    finally:
        icp_mod.distance_matrix = real_dm
    out["planar_icp_start"] = start
    out["planar_icp_nn"] = np.stack(nn_log)
    # half of the points on a plane: many 4-samples are coplanar
    hp = mv.copy()
    hp[2, : n // 2] = 150.0
    hf = at.apply_affine_transform(hp, T) + rng.normal(scale=0.5, size=(3, n))
    out["halfplane_moving"], out["halfplane_fixed"] = hp, hf
    np.random.seed(12)
    sets = np.stack([np.random.choice(n // 2, 4, replace=False) for _ in range(40)])          # all coplanar
    out["halfplane_samples"] = sets.astype(np.int32)
    out["halfplane_fits"] = np.stack([ft.get_affine_transform(hp[:, s], hf[:, s]) for s in sets])
    np.random.seed(13)
    A, inl = sc_mod.do_ransac(hp, hf, min_samples=4, trials=300, error=3.0, transform='Affine')
    out["halfplane_ransac_A"], out["halfplane_ransac_inliers"] = np.asarray(A, dtype=np.float64), np.int64(inl)
    # repeated points in a sample cannot happen (replace=False) but repeated COORDINATES can: duplicate nuclei
    dup = mv[:, :12].copy()
    dup[:, 1] = dup[:, 0]
    dupf = fx[:, :12].copy()
    out["dup_moving"], out["dup_fixed"] = dup, dupf
    out["dup_fit_4"] = ft.get_affine_transform(dup[:, :4], dupf[:, :4])
    # transform() by itself: the frame of one point as get_unary builds it (shape_context.py:169-177)
    det = mv[:, 5]
    c = utils.get_centroid(mv, transposed=False)[:, 0]
    z = (det - c) / np.linalg.norm(det - c)
    x0 = np.array([0.3, -0.5, 0.81])
    x = x0 - z * np.dot(x0, z)
    x = x / np.linalg.norm(x)
    y = sc_mod.get_Y(z, x)
    nb = np.delete(mv.T, 5, 0)
    out["tf_detection"], out["tf_x"], out["tf_y"], out["tf_z"], out["tf_neighbors"] = det, x, y, z, nb
    out["tf_out"] = sc_mod.transform(det[None, :], x[None, :], y[None, :], z[None, :], nb)
    out["tf_out_1d"] = sc_mod.transform(det, x, y, z, nb)
    out["get_Y"] = y
    return out


ALL = ["next_rows", "micro", "similar_mode", "ransac_k", "synth128", "synth96x128", "insitu02_identity", "insitu02_affine", "insitu04_affine"]

if __name__ == "__main__":
    todo = sys.argv[1:] or ALL
    ref = import_reference()
    import scipy
    import sklearn
    versions = "numpy %s scipy %s sklearn %s python %s" % (np.__version__, scipy.__version__, sklearn.__version__, sys.version.split()[0])
    for name in todo:
        t0 = time.time()
        if name == "micro":
            res = micro(ref)
        elif name == "next_rows":
            res = next_rows(ref)
        elif name == "similar_mode":
            res = similar_mode(ref)
        elif name == "ransac_k":
            res = ransac_k(ref)
        else:
            res, a_gt = scenario(ref, name)
            res["A_gt"] = a_gt
        res["versions"] = np.array(versions)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        print("wrote %s (%.1f KB) in %.1fs" % (path, os.path.getsize(path) / 1024, time.time() - t0), flush=True)
