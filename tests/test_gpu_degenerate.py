"""What the widget can reach beyond its defaults, against known answers produced by the reference
(tests/golden/ransac_k.npz, made by tests/golden/gen_golden.py::ransac_k):
  * do_ransac with min_samples != 4 (_dock_widget.py:327 exposes the field; shape_context.py:121-127);
  * rank-deficient fits — fewer than four pairs, coplanar samples, a planar cloud — where the reference's
    fixed . pinv(moving) (find_transform.py:17) returns a minimum-norm answer;
  * shape_context.transform() and get_Y called on their own (shape_context.py:6-8, 61-84)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


@pytest.fixture(scope="module")
def rk():
    import torch
    from platymatch_amd import _native as nat
    from platymatch_amd.build import build_native
    from platymatch_amd.estimate_transform import perform_icp as pi
    build_native()
    nat.load()
    assert torch.cuda.is_available()
    pi.VERBOSE = False
    return load_golden("ransac_k")


def _dev(x, dtype=None):
    import torch
    from platymatch_amd import _native as nat
    return nat.to_dev(x, dtype=dtype or torch.float64)


@pytest.mark.parametrize("k", [5, 8, 13])
def test_per_trial_fits_with_more_than_four_samples_match_the_reference(rk, k):
    import torch
    from platymatch_amd import _kernels as K
    s = rk["samples_k%d" % k]
    A, inl, deg = K.ransac_affine(_dev(rk["moving"]), _dev(rk["fixed"]), None, None, _dev(s, torch.int32), 3.0)
    assert int(deg.sum()) == 0
    A = A.cpu().numpy()
    for t in range(len(s)):
        assert relerr(A[t], rk["fits_k%d" % k][t]) < 1e-9, t


@pytest.mark.parametrize("k", [3, 5, 8])
def test_seeded_do_ransac_with_other_sample_counts_matches_the_reference(rk, k):
    from platymatch_amd.estimate_transform.shape_context import do_ransac
    kk, trials, err, seed = rk["ransac_args_k%d" % k]
    np.random.seed(int(seed))
    A, inl = do_ransac(rk["moving"], rk["fixed"], min_samples=int(kk), trials=int(trials), error=float(err), transform='Affine')
    assert inl == int(rk["ransac_inliers_k%d" % k])
    assert relerr(A, rk["ransac_A_k%d" % k]) < 1e-9                   # k = 3: pinv's minimum-norm matrix, last row and all


@pytest.mark.parametrize("k", [1, 2, 3])
def test_fits_through_fewer_than_four_pairs_are_pinv_minimum_norm(rk, k, oracle):
    from platymatch_amd.estimate_transform.find_transform import get_affine_transform
    mv, fx = rk["moving"], rk["fixed"]
    for t, s in enumerate(rk["samples_k%d" % k][:25]):
        A = get_affine_transform(np.ascontiguousarray(mv[:, s]), np.ascontiguousarray(fx[:, s]))
        assert relerr(A, rk["fits_k%d" % k][t]) < 1e-10, t


def test_planar_cloud_fit_ransac_and_icp_match_the_reference(rk, oracle):
    """2-D data embedded in 3-D (one coordinate constant): the reference keeps working through pinv."""
    from platymatch_amd.estimate_transform.find_transform import get_affine_transform
    from platymatch_amd.estimate_transform.perform_icp import perform_icp
    from platymatch_amd.estimate_transform.shape_context import do_ransac
    pl, pf = rk["planar_moving"], rk["planar_fixed"]
    assert relerr(get_affine_transform(pl, pf), rk["planar_fit"]) < 1e-9
    assert relerr(get_affine_transform(pl[:, :4].copy(), pf[:, :4].copy()), rk["planar_fit_4"]) < 1e-9
    assert relerr(get_affine_transform(pl[:, :3].copy(), pf[:, :3].copy()), rk["planar_fit_3"]) < 1e-9
    np.random.seed(4)
    A, inl = do_ransac(pl, pf, min_samples=4, trials=200, error=3.0, transform='Affine')        # every sample is coplanar
    assert inl == int(rk["planar_ransac_inliers"]) and relerr(A, rk["planar_ransac_A"]) < 1e-9
    log = {}
    A_icp = perform_icp(rk["planar_icp_start"], pf, 6, 'Affine', log=log)
    assert np.array_equal(log["nn"], rk["planar_icp_nn"])                                       # every iteration's correspondences
    assert relerr(A_icp, rk["planar_icp_A"]) < 1e-8
    # the torch-in / torch-out form takes the same route
    A_t = perform_icp(_dev(rk["planar_icp_start"]), _dev(pf), 6, 'Affine')
    assert relerr(A_t.cpu().numpy(), rk["planar_icp_A"]) < 1e-8


def test_coplanar_samples_inside_a_3d_cloud_are_refitted_like_the_reference(rk, oracle):
    import torch
    from platymatch_amd import _kernels as K
    from platymatch_amd.estimate_transform.shape_context import do_ransac
    hp, hf = rk["halfplane_moving"], rk["halfplane_fixed"]
    s = rk["halfplane_samples"]                                       # all four points of every sample lie in one plane
    A, inl, deg = K.ransac_affine(_dev(hp), _dev(hf), None, None, _dev(s, torch.int32), 3.0)
    assert int(deg.sum()) == len(s) and int(inl.sum()) == 0 and bool(torch.isnan(A).all())
    got = do_ransac(hp, hf, min_samples=4, trials=len(s), error=3.0, transform='Affine', samples=s)
    want = oracle.do_ransac(hp, hf, min_samples=4, trials=len(s), error=3.0, transform='Affine', samples=s)
    assert got[1] == want[1] and relerr(got[0], want[0]) < 1e-9
    best = int(np.argmax(oracle.ransac_score(hp, hf, rk["halfplane_fits"], 3.0)))
    assert relerr(got[0], rk["halfplane_fits"][best]) < 1e-9
    np.random.seed(13)                                                # mixed: about 1 sample in 16 is coplanar
    A, k = do_ransac(hp, hf, min_samples=4, trials=300, error=3.0, transform='Affine')
    assert k == int(rk["halfplane_ransac_inliers"]) and relerr(A, rk["halfplane_ransac_A"]) < 1e-9
    # duplicate nuclei inside a sample
    from platymatch_amd.estimate_transform.find_transform import get_affine_transform
    assert relerr(get_affine_transform(rk["dup_moving"][:, :4].copy(), rk["dup_fixed"][:, :4].copy()), rk["dup_fit_4"]) < 1e-9


def test_estimate_transform_on_a_planar_pair_equals_the_oracle(rk, oracle):
    """End to end with a flat moving cloud: descriptors, costs and assignments exact; RANSAC and ICP through the pinv route."""
    from platymatch_amd import pipeline as P
    pl, pf = rk["planar_moving"], rk["planar_fixed"]
    kw = dict(ransac_trials=150, ransac_error=3.0, icp_iterations=5, seed=2)
    dg, do = {}, {}
    with np.errstate(all="ignore"):
        A_sc, A_icp, inl = P.estimate_transform(pl, pf, details=dg, **kw)
        o_sc, o_icp, o_inl = oracle.estimate_transform(pl, pf, details=do, **kw)
    for h in range(8):
        assert np.array_equal(dg["lsa"][h][1], do["lsa"][h][1])
    assert np.array_equal(inl, o_inl) and np.array_equal(dg["nn"], do["nn"])
    assert relerr(A_sc, o_sc) < 1e-9 and relerr(A_icp @ A_sc, o_icp @ o_sc) < 1e-8


def test_transform_and_get_Y_on_their_own(rk):
    from platymatch_amd.estimate_transform import shape_context as sc
    got = sc.transform(rk["tf_detection"][None, :], rk["tf_x"][None, :], rk["tf_y"][None, :], rk["tf_z"][None, :], rk["tf_neighbors"])
    assert got.shape == rk["tf_out"].shape and np.abs(got - rk["tf_out"]).max() < 1e-10
    got1 = sc.transform(rk["tf_detection"], rk["tf_x"], rk["tf_y"], rk["tf_z"], rk["tf_neighbors"])
    assert np.abs(got1 - rk["tf_out_1d"]).max() < 1e-10
    # the frame coordinates feed get_shape_context exactly as in get_unary (shape_context.py:177-187)
    h_ref = sc.get_shape_context(rk["tf_out"], 55.0)
    h_got = sc.get_shape_context(got, 55.0)
    assert np.array_equal(h_ref, h_got)
    assert np.allclose(sc.get_Y(rk["tf_z"], rk["tf_x"]), rk["get_Y"], rtol=0, atol=1e-15)
    t_in = _dev(rk["tf_neighbors"])
    t_out = sc.transform(rk["tf_detection"], rk["tf_x"], rk["tf_y"], rk["tf_z"], t_in)                # torch in -> torch out
    assert t_out.is_cuda and np.abs(t_out.cpu().numpy() - rk["tf_out"]).max() < 1e-10


def test_host_arithmetic_self_check_passes_on_this_box():
    """platymatch_amd.self_check: NumPy's buffered pairwise summation, BLAS ddot's and dgemm's fused multiply-adds —
    the host properties the kernels restate — hold on the GPU box's host, so no HostArithmeticWarning is raised and the three
    verdicts are True; estimate_transform runs the check once per process."""
    import warnings
    import platymatch_amd
    with warnings.catch_warnings():
        warnings.simplefilter("error", platymatch_amd.HostArithmeticWarning)
        verdicts = platymatch_amd.self_check(force=True)
    assert len(verdicts) == 3 and all(verdicts.values()), verdicts
