"""The reference widget's worker, UNCHANGED in shape, on top of the drop-in: after platymatch_amd.install_as_platymatch()
the names the widget binds at _dock_widget.py:15-21 are imported from `platymatch.*` and then called in the order, with the
argument forms and with the Python double loops of EstimateTransform._click_run (_dock_widget.py:526-718): get_centroid,
get_mean_distance, get_unary x 2, eight `for i: for j: U[i, j] = get_unary_distance(a[i], b[j])` loops, SciPy's
linear_sum_assignment as the widget imports it (:10), eight do_ransac calls on fancy-indexed host arrays,
np.argmax(inliers), apply_affine_transform, perform_icp, A_icp @ A_sc (:428).  Checked against the reference's own results
for that cloud (tests/golden/insitu02_affine.npz, made by running the reference)."""
import time

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


def _widget_body(moving_detections, fixed_detections, ransac_samples, ransac_iterations, ransac_error, icp_iterations,
                 transform_kind, seed):
    import platymatch_amd
    platymatch_amd.install_as_platymatch()
    # the widget's imports (_dock_widget.py:10, 15-21), resolved through the reference's module paths
    from scipy.optimize import linear_sum_assignment
    from platymatch.estimate_transform.apply_transform import apply_affine_transform
    from platymatch.estimate_transform.perform_icp import perform_icp
    from platymatch.estimate_transform.shape_context import get_unary, get_unary_distance, do_ransac
    from platymatch.utils.utils import get_centroid, get_mean_distance

    moving_centroid = get_centroid(moving_detections, transposed=False)                                       # :526
    fixed_centroid = get_centroid(fixed_detections, transposed=False)                                         # :527
    moving_mean_distance = get_mean_distance(moving_detections, transposed=False)                             # :531
    fixed_mean_distance = get_mean_distance(fixed_detections, transposed=False)                               # :532
    moving_copy, fixed_copy = moving_detections.copy(), fixed_detections.copy()                               # :533-534
    unary_11, unary_12, _, _ = get_unary(moving_centroid, mean_distance=moving_mean_distance,                 # :540-542
                                         detections=moving_detections, type='moving', transposed=False)
    unary_21, unary_22, unary_23, unary_24 = get_unary(fixed_centroid, mean_distance=fixed_mean_distance,     # :543-545
                                                       detections=fixed_detections, type='fixed', transposed=False)
    n1, n2 = moving_detections.shape[1], fixed_detections.shape[1]
    t_loops = time.perf_counter()
    U = []
    for a, b in ((unary_11, unary_21), (unary_11, unary_22), (unary_11, unary_23), (unary_11, unary_24),      # :547-602
                 (unary_12, unary_21), (unary_12, unary_22), (unary_12, unary_23), (unary_12, unary_24)):
        Uab = np.zeros((n1, n2))
        for i in range(Uab.shape[0]):
            for j in range(Uab.shape[1]):
                unary_i = a[i]
                unary_j = b[j]
                Uab[i, j] = get_unary_distance(unary_i, unary_j)
        U.append(Uab)
    t_loops = time.perf_counter() - t_loops
    lsa = [linear_sum_assignment(Uab) for Uab in U]                                                            # :604-611
    np.random.seed(seed)             # the reference's RANSAC draws from NumPy's global generator (shape_context.py:122)
    results = [do_ransac(moving_copy[:, r], fixed_copy[:, c], min_samples=ransac_samples, trials=ransac_iterations,
                         error=ransac_error, transform=transform_kind) for r, c in lsa]                        # :622-675
    inliers = np.array([k for _, k in results])                                                                # :683-686
    transform_matrix_sc = results[int(np.argmax(inliers))][0]                                                  # :688-703
    transformed = apply_affine_transform(moving_copy, transform_matrix_sc)                                     # :714
    transform_matrix_icp = perform_icp(transformed, fixed_copy, icp_iterations, transform_kind)                # :715-717
    return dict(U=U, lsa=lsa, inliers=inliers, A_sc=transform_matrix_sc, A_icp=transform_matrix_icp, loop_seconds=t_loops,
                unaries=(unary_11, unary_12, unary_21, unary_22, unary_23, unary_24))


def test_unchanged_widget_body_on_the_drop_in_matches_the_reference():
    import torch
    from platymatch_amd import _native as nat
    from platymatch_amd.build import build_native
    from platymatch_amd.estimate_transform import perform_icp as pi
    build_native()
    nat.load()
    assert torch.cuda.is_available()
    pi.VERBOSE = False
    d = load_golden("insitu02_affine")
    t0 = time.perf_counter()
    out = _widget_body(d["moving"].copy(), d["fixed"].copy(), 4, int(d["ransac_trials"]), float(d["ransac_error"]),
                       int(d["icp_iters"]), 'Affine', int(d["ransac_seed"]))
    wall = time.perf_counter() - t0
    print("widget replay, 331 x 331 nuclei: %.2f s wall, of which the eight get_unary_distance double loops %.2f s "
          "(%d calls)" % (wall, out["loop_seconds"], 8 * 331 * 331))
    # descriptors: ordinary float64 (N, 360) arrays, bit-identical to the reference's
    for a, key in zip(out["unaries"], ("m1", "m2", "f1", "f2", "f3", "f4")):
        want = d["counts_" + key].astype(np.float64) / d["total_" + key].astype(np.float64)[:, None]
        assert isinstance(a, np.ndarray) and a.dtype == np.float64 and np.array_equal(a, want)
    # the matrices the loops filled: the reference's own floats (stored rows), and its np.argmin per row
    for h in range(8):
        assert np.array_equal(out["U"][h][d["U_rows"]].view(np.uint64), d["U"][h].view(np.uint64)), h
        assert np.array_equal(out["U"][h].argmin(1), d["U_rowmin_idx"][h])
        assert np.array_equal(out["lsa"][h][0], d["lsa_rows"][h]) and np.array_equal(out["lsa"][h][1], d["lsa_cols"][h])
    assert np.array_equal(out["inliers"], d["ransac_inliers"])
    assert relerr(out["A_sc"], d["A_sc"]) < 1e-8
    A_final = out["A_icp"] @ out["A_sc"]                                                                       # :428
    assert relerr(A_final, d["A_final"]) < 1e-9
    np.testing.assert_array_almost_equal(d["A_gt"], A_final)          # the reference test's own assertion (decimal 6)
    assert wall < 60.0 and out["loop_seconds"] < 30.0                 # the reference needs ~90 s for these loops alone (typically 3 s here; generous: shared host cores)


def test_get_unary_distance_lookup_equals_per_pair_launch_and_survives_foreign_inputs():
    """The cached-matrix answer is the per-pair kernel's answer; copies, slices, modified arrays and plain ndarrays take
    the per-pair path; a descriptor set modified in place after get_unary is never answered from the device copy."""
    import platymatch_amd
    platymatch_amd.install_as_platymatch()
    from platymatch.estimate_transform import shape_context as sc
    from platymatch.utils.utils import get_centroid, get_mean_distance
    d = load_golden("synth96x128")
    mv, fx = d["moving"], d["fixed"]
    um = sc.get_unary(get_centroid(mv, transposed=False), get_mean_distance(mv, transposed=False), mv, 'moving')
    uf = sc.get_unary(get_centroid(fx, transposed=False), get_mean_distance(fx, transposed=False), fx, 'fixed')
    assert um[2].shape == (0,) and um[3].shape == (0,)
    for (a, b) in ((um[0], uf[2]), (um[1], uf[0]), (uf[1], uf[3]), (um[0], um[1]), (uf[0], um[1])):
        for i, j in ((0, 0), (5, 77), (a.shape[0] - 1, b.shape[0] - 1), (-1, -2)):
            fast = sc.get_unary_distance(a[i], b[j])
            slow = sc.get_unary_distance(np.array(a[i]), np.array(b[j]))          # plain copies: one launch per pair
            assert isinstance(fast, float) and fast == slow
    # derived arrays are foreign: they must not be mistaken for rows of the set
    a, b = um[0], uf[0]
    assert sc.get_unary_distance(a[3] * 1.0, b[4]) == sc.get_unary_distance(np.array(a[3]), np.array(b[4]))
    assert sc.get_unary_distance(a[2:4][1], b[4]) == sc.get_unary_distance(np.array(a[3]), np.array(b[4]))
    # in-place modification after get_unary: a fresh pair of sets, modified before the first distance call
    um2 = sc.get_unary(get_centroid(mv, transposed=False), get_mean_distance(mv, transposed=False), mv, 'moving')
    um2[0][7, :] = um2[0][8, :]
    assert sc.get_unary_distance(um2[0][7], uf[0][1]) == sc.get_unary_distance(np.array(um2[0][8]), np.array(uf[0][1]))
