"""Pins the oracle (oracle/) to the unmodified reference: every fixture under tests/golden/ was
produced by running the reference itself (gen_golden.py).  CPU only."""
import numpy as np
from scipy.optimize import linear_sum_assignment

from conftest import load_golden


def test_micro_binning_known_answers(oracle, micro):
    # get_bin_index on 500 random neighbours and get_shape_context on the integer lattice (exact ring,
    # theta and phi edges, the zero vector) — reference shape_context.py:10-58
    assert np.array_equal(oracle.get_bin_index_direct(micro["rand_neighbors"], 55.0), micro["rand_bin_index"])
    assert np.array_equal(oracle.get_shape_context(micro["rand_neighbors"], 55.0), micro["rand_sc"])
    for md, key in ((1.0, "grid_sc_md1"), (3.0, "grid_sc_md3")):
        assert np.array_equal(oracle.get_shape_context(micro["grid_neighbors"], md), micro[key])


def test_micro_chi2_and_fits(oracle, micro):
    assert oracle.get_unary_distance(micro["chi2_a"], micro["chi2_b"]) == micro["chi2_ab"]
    assert oracle.get_unary_distance(micro["chi2_a"], micro["chi2_a"]) == micro["chi2_aa"] == 0.0
    P, Q = micro["fit_moving"], micro["fit_fixed"]
    assert np.array_equal(oracle.get_affine_transform(P, Q), micro["fit_affine"])
    assert np.array_equal(oracle.get_affine_transform(P[:, :4], Q[:, :4]), micro["fit_affine4"])
    assert np.array_equal(oracle.get_similar_transform(P, Q), micro["fit_similar"])
    assert np.array_equal(oracle.get_similar_transform(P[:, :4], Q[:, :4]), micro["fit_similar4"])
    A = micro["fit_affine"]
    assert np.array_equal(oracle.apply_affine_transform(P, A), oracle.apply_affine_transform(np.vstack([P, np.ones((1, 40))]), A))
    assert oracle.get_error(P, Q) == micro["error_PQ"]
    assert np.array_equal(oracle.get_centroid(P, transposed=False), micro["centroid_F"])
    assert np.array_equal(oracle.get_centroid(P.T, transposed=True), micro["centroid_T"])
    np.testing.assert_array_almost_equal(oracle.get_centroid(micro["cube"], transposed=True), [[0.5, 0.5, 0.5]])  # _tests/test_utils.py
    assert oracle.get_mean_distance(P, transposed=False) == micro["mean_distance"]          # NumPy's summation order, BLAS's fused norm: the bits


def test_micro_degenerate_cloud(oracle, micro):
    """Point == centroid gives a NaN row; exactly (anti)parallel / duplicated neighbours are decided by
    rounding noise inside the reference's 4x4 inverse (transform, shape_context.py:61-84) and are not
    reproducible by any restatement — so only the (ring, theta) marginals of unaffected rows are compared."""
    cloud = micro["degenerate_cloud"]
    with np.errstate(all="ignore"):
        u = oracle.get_unary(micro["degenerate_centroid"], micro["degenerate_mean_dist"], cloud, "fixed")
    ref = micro["degenerate_sc1"]
    assert np.isnan(ref[40]).all() and np.isnan(u[0][40]).all()          # the origin is the centroid
    for k in range(4):
        r = micro["degenerate_sc%d" % (k + 1)]
        assert np.array_equal(np.isnan(r).any(1), np.isnan(u[k]).any(1))
    dup = {0, 41, 20, 42}                                                    # +-p0 appear twice
    for i in range(cloud.shape[1]):
        if i == 40 or i in dup:
            continue
        tot = round(1.0 / ref[i][ref[i] > 0].min() * round(ref[i][ref[i] > 0].min() * 42))
        a = np.rint(ref[i] * 42).reshape(30, 12).sum(1)
        b = np.rint(u[0][i] * 42).reshape(30, 12).sum(1)
        assert np.array_equal(a, b), (i, tot)


def test_scenario_statistics(oracle, scenario):
    name, d = scenario
    mv, fx = d["moving"], d["fixed"]
    assert np.array_equal(oracle.get_centroid(mv, transposed=False), d["centroid_m"])
    assert oracle.get_mean_distance(mv, transposed=False) == d["mean_dist_m"]              # bit for bit (round 3)
    assert oracle.get_mean_distance(fx, transposed=False) == d["mean_dist_f"]
    assert np.array_equal(oracle.pca_axis(mv.T), d["x0_m"])              # sklearn's PCA axis, bit for bit (round 4: its own NumPy calls restated)
    assert np.array_equal(oracle.pca_axis(fx.T), d["x0_f"])


def test_scenario_histograms_exact(oracle, scenario):
    name, d = scenario
    cm, tm = oracle.shape_context_counts(d["centroid_m"], d["mean_dist_m"], d["moving"], "moving", x0=d["x0_m"])
    cf, tf = oracle.shape_context_counts(d["centroid_f"], d["mean_dist_f"], d["fixed"], "fixed", x0=d["x0_f"])
    for k in range(2):
        assert np.array_equal(cm[k], d["counts_m%d" % (k + 1)]) and np.array_equal(tm[k], d["total_m%d" % (k + 1)])
    for k in range(4):
        assert np.array_equal(cf[k], d["counts_f%d" % (k + 1)]) and np.array_equal(tf[k], d["total_f%d" % (k + 1)])
    # the oracle's own statistics lead to the same integer histograms
    cm2, _ = oracle.shape_context_counts(oracle.get_centroid(d["moving"], False), oracle.get_mean_distance(d["moving"], False),
                                         d["moving"], "moving")
    assert np.array_equal(cm2, cm)


def test_scenario_costs_bit_exact_and_assignment(oracle, scenario):
    name, d = scenario
    um = oracle.normalise_counts(*oracle.shape_context_counts(d["centroid_m"], d["mean_dist_m"], d["moving"], "moving", x0=d["x0_m"]))
    uf = oracle.normalise_counts(*oracle.shape_context_counts(d["centroid_f"], d["mean_dist_f"], d["fixed"], "fixed", x0=d["x0_f"]))
    for h, nm in enumerate(oracle.HYPOTHESES):
        U = oracle.unary_distance_matrix(um[int(nm[0]) - 1], uf[int(nm[1]) - 1])
        assert np.array_equal(U[d["U_rows"]], d["U"][h])                  # float64, bit for bit
        assert U.sum() == d["U_sum"][h]
        assert np.array_equal(U.argmin(1), d["U_rowmin_idx"][h])
        r, c = linear_sum_assignment(U)
        assert np.array_equal(r, d["lsa_rows"][h]) and np.array_equal(c, d["lsa_cols"][h])


def test_scenario_end_to_end(oracle, scenario):
    name, d = scenario
    det = {}
    A_sc, A_icp, inl = oracle.estimate_transform(d["moving"], d["fixed"], ransac_trials=int(d["ransac_trials"]),
                                                 ransac_error=float(d["ransac_error"]), icp_iterations=int(d["icp_iters"]),
                                                 seed=int(d["ransac_seed"]), details=det)
    assert np.array_equal(inl, d["ransac_inliers"])
    assert int(np.argmax(inl)) == int(d["best_hypothesis"])
    assert np.array_equal(det["ransac_A"], d["ransac_A"]) and np.array_equal(A_sc, d["A_sc"])
    assert np.array_equal(det["nn"], d["icp_nn"])
    assert np.array_equal(det["residuals"], d["icp_residuals"])
    assert np.array_equal(A_icp, d["A_icp"]) and np.array_equal(A_icp @ A_sc, d["A_final"])
    if name.startswith("insitu"):
        # the reference's own assertion (_tests/test_estimate_transform.py:72,140,208), decimal 6
        np.testing.assert_array_almost_equal(d["A_gt"], A_icp @ A_sc)


def test_similar_mode_matches_reference(oracle):
    """transform='Similar' (similar_mode.npz): per-sample fits, seeded do_ransac and a Similar-mode ICP, all bit for bit."""
    from conftest import GOLDEN
    import os
    d = np.load(os.path.join(GOLDEN, "similar_mode.npz"))
    mv, fx = d["moving"], d["fixed"]
    for k in (4, 6, 9, 20):
        got = np.stack([oracle.get_similar_transform(mv[:, s], fx[:, s]) for s in d["samples_k%d" % k]])
        assert np.array_equal(got, d["fits_k%d" % k]), k
    for k in (4, 9):
        kk, trials, err, seed = d["ransac_args_k%d" % k]
        np.random.seed(int(seed))
        A, inl = oracle.do_ransac(mv, fx, min_samples=int(kk), trials=int(trials), error=err, transform="Similar")
        assert np.array_equal(A, d["ransac_A_k%d" % k]) and inl == int(d["ransac_inliers_k%d" % k])
    A_icp = oracle.perform_icp(oracle.apply_affine_transform(mv, d["ransac_A_k4"]), fx, 12, "Similar")
    assert np.array_equal(A_icp, d["icp_A"])


def test_oracle_reproduces_ransac_k_fixture(oracle):
    """min_samples != 4, rank-deficient fits (pinv's minimum-norm answer), planar cloud through RANSAC and ICP:
    the oracle against what the reference produced (tests/golden/ransac_k.npz), bit for bit."""
    d = load_golden("ransac_k")
    for k in (1, 2, 3, 5, 8, 13):
        fits = np.stack([oracle.get_affine_transform(d["moving"][:, s], d["fixed"][:, s]) for s in d["samples_k%d" % k]])
        assert np.array_equal(fits, d["fits_k%d" % k]), k
    for k in (3, 5, 8):
        kk, trials, err, seed = d["ransac_args_k%d" % k]
        np.random.seed(int(seed))
        A, inl = oracle.do_ransac(d["moving"], d["fixed"], min_samples=int(kk), trials=int(trials), error=float(err))
        assert inl == int(d["ransac_inliers_k%d" % k]) and np.array_equal(A, d["ransac_A_k%d" % k])
    assert np.array_equal(oracle.get_affine_transform(d["planar_moving"], d["planar_fixed"]), d["planar_fit"])
    np.random.seed(4)
    A, inl = oracle.do_ransac(d["planar_moving"], d["planar_fixed"], 4, 200, 3.0)
    assert inl == int(d["planar_ransac_inliers"]) and np.array_equal(A, d["planar_ransac_A"])
    log = {}
    A = oracle.perform_icp(d["planar_icp_start"], d["planar_fixed"], 6, "Affine", log=log)
    assert np.array_equal(A, d["planar_icp_A"]) and np.array_equal(log["nn"], d["planar_icp_nn"])
    np.random.seed(13)
    A, inl = oracle.do_ransac(d["halfplane_moving"], d["halfplane_fixed"], 4, 300, 3.0)
    assert inl == int(d["halfplane_ransac_inliers"]) and np.array_equal(A, d["halfplane_ransac_A"])


# ------------------------------------------------------------------------------------------------ random small clouds (round 3)
def _random_small():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "random_small.npz"))


def test_oracle_reproduces_the_reference_on_36_random_small_clouds(oracle):
    """tests/golden/random_small.npz (gen_random_small.py: the unmodified reference on 36 generic pairs of 8-40 points): centroid,
    mean distance, every histogram, seeded do_ransac and perform_icp in both modes — bit for bit; the two stored chi-square
    matrices bit for bit except where libm's pow(x, 2.0) behind the reference's `** 2` is not x * x: at most one
    ulp, in 12 of the 39 992 entries."""
    d = _random_small()
    off = entries = 0
    for k in range(int(d["cases"][0])):
        p = "c%02d_" % k
        mv, fx = d[p + "moving"], d[p + "fixed"]
        n, m = mv.shape[1], fx.shape[1]
        assert np.array_equal(np.asarray(oracle.get_centroid(mv, False)), d[p + "centroid_m"])
        assert np.array_equal(np.asarray(oracle.get_centroid(fx, False)), d[p + "centroid_f"])
        assert oracle.get_mean_distance(mv, False) == d[p + "mean_dist"][0] and oracle.get_mean_distance(fx, False) == d[p + "mean_dist"][1]
        cm, tm = oracle.shape_context_counts(oracle.get_centroid(mv, False), oracle.get_mean_distance(mv, False), mv, "moving")
        cf, tf = oracle.shape_context_counts(oracle.get_centroid(fx, False), oracle.get_mean_distance(fx, False), fx, "fixed")
        assert np.array_equal(cm, d[p + "counts_m"]) and np.array_equal(cf, d[p + "counts_f"]), k
        um, uf = oracle.normalise_counts(cm, tm), oracle.normalise_counts(cf, tf)
        for name, a, b in (("U11", um[0], uf[0]), ("U24", um[1], uf[3])):
            got, want = np.asarray(oracle.unary_distance_matrix(a, b)), d[p + name]
            diff = got != want
            entries += want.size
            off += int(diff.sum())
            assert np.all(np.abs(got[diff].view(np.int64) - want[diff].view(np.int64)) <= 1), (k, name)     # one ulp at most
        pq = min(n, m)
        for tr in ("Affine", "Similar"):
            np.random.seed(int(d[p + "seed"][0]))
            A, inl = oracle.do_ransac(mv[:, :pq], fx[:, :pq], 4, 30, 10.0, tr)
            assert int(inl) == int(d[p + "ransac_inl_" + tr][0]) and np.array_equal(np.asarray(A), d[p + "ransac_A_" + tr]), (k, tr)
            assert np.array_equal(np.asarray(oracle.perform_icp(mv, fx, 5, tr)), d[p + "icp_" + tr], equal_nan=True), (k, tr)
    assert entries == 39992 and off == 12, (off, entries)          # deterministic data: exactly these twelve, one ulp each


def test_oracle_end_to_end_on_24_random_small_pairs(oracle):
    """tests/golden/random_e2e.npz (gen_random_e2e.py: the unmodified reference end to end on generic, voxel, lattice and
    asset-like pairs of 14-35 points).  Generic pairs: the eight assignment vectors, inlier counts, A_sc, every ICP correspondence
    and A_final bit for bit.  Pairs with lattice structure: the same wherever no neighbour sits on a bin boundary (the oracle has no
    guard of its own: the count of such pairs that agree is reported; the HIP path's guard is tested on the GPU)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "random_e2e.npz"))
    agree = {0: [0, 0], 1: [0, 0], 2: [0, 0], 3: [0, 0]}
    for k in range(int(d["cases"][0])):
        p = "c%02d_" % k
        kind = int(d[p + "kind"][0])
        det = {}
        got = oracle.estimate_transform(d[p + "moving"], d[p + "fixed"], ransac_trials=60, ransac_error=12.0, icp_iterations=4, seed=0, details=det)
        same = (all(np.array_equal(det["lsa"][h][0], d[p + "lsa_rows"][h]) and np.array_equal(det["lsa"][h][1], d[p + "lsa_cols"][h]) for h in range(8))
                and np.array_equal(got[2], d[p + "ransac_inliers"]) and np.array_equal(got[0], d[p + "A_sc"])
                and np.array_equal(np.asarray(det["nn"]), d[p + "icp_nn"]) and np.array_equal(got[1] @ got[0], d[p + "A_final"]))
        agree[kind][0] += same
        agree[kind][1] += 1
        if kind == 0:
            assert same, k
    assert agree[0] == [6, 6]
    print("pairs equal to the reference end to end / pairs: generic %s, voxel %s, lattice %s, asset-like %s" % tuple(agree[k] for k in range(4)))


# ------------------------------------------------------------------------------------------------ lopsided lattice pairs (round 4)
LOPSIDED_ORACLE_EQUALS_REFERENCE = (30009, 30019, 30034)       # end to end; the other thirteen differ in a histogram that matters


def test_oracle_against_the_reference_on_the_lopsided_lattice_pairs(oracle):
    """tests/golden/lopsided.npz (gen_lopsided.py): the unmodified reference on sixteen pairs of one tiny (4-12 points) and one
    larger cloud, both on a half-integer lattice with duplicates — the family on which the HIP path and this oracle disagreed eight
    times in round 3's soak.  What holds on all sixteen: centroid-independent statistics (mean distances) bit for bit.  What the
    reference's verdict is: on every pair at least one histogram differs from the oracle's — every neighbour of such a cloud sits
    ON a ring radius / sector plane, where the reference bins by the rounding noise of its np.linalg.inv — and
    only three pairs come out equal end to end.  Neither the oracle nor the HIP path is "right" on the others: the reference's
    own result there is a property of its LAPACK build (the GPU test asserts that the edge guard says so for every such pair)."""
    d = load_golden("lopsided")
    equal = []
    for seed in d["seeds"]:
        p = "s%d_" % seed
        mv, fx = d[p + "moving"], d[p + "fixed"]
        assert oracle.get_mean_distance(mv, False) == d[p + "mean_dist_m"] and oracle.get_mean_distance(fx, False) == d[p + "mean_dist_f"]
        det = {}
        got = oracle.estimate_transform(mv, fx, transform="Affine", ransac_trials=80, ransac_error=float(d[p + "ransac_error"][0]),
                                        icp_iterations=4, seed=int(d[p + "ransac_seed"][0]), details=det)
        same = (all(np.array_equal(det["lsa"][h][0], d[p + "lsa_rows"][h]) and np.array_equal(det["lsa"][h][1], d[p + "lsa_cols"][h]) for h in range(8))
                and np.array_equal(got[2], d[p + "ransac_inliers"]))
        if same:
            equal.append(int(seed))
            if np.isfinite(d[p + "A_sc"]).all() and np.linalg.cond(d[p + "A_sc"]) < 1e8:
                assert np.array_equal(got[0], d[p + "A_sc"]) and np.array_equal(np.asarray(det["nn"]), d[p + "icp_nn"]), seed
    assert tuple(equal) == LOPSIDED_ORACLE_EQUALS_REFERENCE, equal



def _pca_views(d, k):
    """The N x 3 array the reference would hand to sklearn for fixture case k of pca_axis.npz (gen_pca_axis.py)."""
    cloud, kind = d["c%02d_cloud" % k], int(d["c%02d_kind" % k][0])
    return cloud if kind == 2 else (cloud.transpose()[:, :3] if kind == 3 else cloud.transpose())


def test_pca_axis_is_sklearns_bit_for_bit(oracle):
    """tests/golden/pca_axis.npz (gen_pca_axis.py): sklearn.decomposition.PCA(3).fit(X).components_[0] for 40 clouds of 4 .. 1 500
    points in the layouts the reference meets (transposed views, C-ordered N x 3, voxel coordinates, a sliced 4 x N array), both
    solver branches.  The oracle's restatement and the PRODUCT's (shape_context.pca_axis_host, host NumPy by design: the axis hangs
    on BLAS's accumulation order and LAPACK's eigh) both reproduce it bit for bit; pca_view builds the reference's view from the
    caller's array."""
    from platymatch_amd.estimate_transform import shape_context as sc
    d = load_golden("pca_axis")
    for k in range(int(d["cases"][0])):
        X, want = _pca_views(d, k), d["c%02d_axis" % k]
        assert np.array_equal(oracle.pca_axis(X), want), k
        assert np.array_equal(sc.pca_axis_host(X), want), k
        kind = int(d["c%02d_kind" % k][0])
        view = sc.pca_view(d["c%02d_cloud" % k], transposed=(kind == 2))
        assert np.array_equal(sc.pca_axis_host(view), want), k


def test_explicit_neighbour_lists_take_numpys_fused_norm(oracle):
    """shape_context.py:29 on an explicit list: a neighbour exactly on a ring radius under np.linalg.norm's fused chain (one ulp
    inside it under the unfused form) lands in the reference's ring."""
    from conftest import ring_edge_neighbours
    for nb, md, want in ring_edge_neighbours():
        assert np.array_equal(oracle.get_shape_context(nb, md), want)
