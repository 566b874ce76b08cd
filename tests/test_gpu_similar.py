"""transform='Similar' with its O(N) arithmetic on the device (pm_similar_moments / pm_similar_apply).

The reference's fit hangs on the last bit of a 4 x 4 matrix (find_transform.py:55-66), so the device must produce NumPy's bits:
every number it hands to the host is compared here with the reference's own NumPy expression evaluated on the host of the
machine the test runs on — centroids in both memory orders, the nine sums, D and Sp, the quaternion matrix, np.matmul's moved
cloud, the residual — and the ICP chain with the oracle's literal loop."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import torch
    from platymatch_amd import _kernels as K, _native as nat
    from platymatch_amd.estimate_transform import perform_icp as pi
    nat.load()
    assert torch.cuda.is_available()
    pi.VERBOSE = False

    class G:
        pass
    G.K, G.nat, G.t, G.dev = K, nat, torch, torch.device("cuda:0")
    return G


def reference_numbers(moving, fixed):
    """find_transform.py:27-53, 86-91 literally (host NumPy) -> the seventeen numbers."""
    ct = np.mean(fixed, 1, keepdims=True)
    cs = np.mean(moving, 1, keepdims=True)
    Y = fixed[:3, :] - ct[:3, :]
    P = moving[:3, :] - cs[:3, :]
    Px, Py, Pz = P[0, :], P[1, :], P[2, :]
    Yx, Yy, Yz = Y[0, :], Y[1, :], Y[2, :]
    S = [np.sum(Yx * Px), np.sum(Px * Yy), np.sum(Px * Yz), np.sum(Py * Yx), np.sum(Py * Yy), np.sum(Py * Yz),
         np.sum(Pz * Yx), np.sum(Pz * Yy), np.sum(Pz * Yz)]
    D = Sp = 0
    for i in range(Y.shape[1]):
        D += np.matmul(np.transpose(Y[:, i]), Y[:, i])
        Sp += np.matmul(np.transpose(P[:, i]), P[:, i])
    return np.concatenate([cs.ravel(), ct.ravel(), np.array(S), [D, Sp]])


@pytest.mark.parametrize("n,m", [(1, 5), (7, 7), (8, 20), (150, 150), (1000, 1300), (8192, 9000), (8193, 8193), (20000, 15000), (50001, 50000)])
def test_seventeen_numbers_equal_numpy_bit_for_bit(g, n, m):
    rng = np.random.default_rng(n)
    mv = np.ascontiguousarray(rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0)
    fx = np.ascontiguousarray(rng.normal(size=(3, m)) * np.array([[60.0], [40.0], [25.0]]) + 190.0)
    nn = rng.integers(0, m, size=n).astype(np.int32)
    mov, fix, nn_d = g.nat.to_dev(mv, dev=g.dev), g.nat.to_dev(fx, dev=g.dev), g.nat.to_dev(nn, dtype=g.t.int32, dev=g.dev)
    matched = fx[:, nn]                                                  # Fortran-ordered, as in perform_icp.py:20
    assert n == 1 or (matched.flags["F_CONTIGUOUS"] and not matched.flags["C_CONTIGUOUS"])
    got = g.K.similar_moments(mov, fix, nn_d, mov_sequential=False, fix_sequential=True).cpu().numpy()
    want = reference_numbers(mv, matched)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (got - want)
    # the other two memory orders: a Fortran-ordered moving cloud, C-ordered matches
    got2 = g.K.similar_moments(mov, fix, nn_d, mov_sequential=True, fix_sequential=False).cpu().numpy()
    want2 = reference_numbers(np.asfortranarray(mv), np.ascontiguousarray(matched))
    assert np.array_equal(got2.view(np.uint64), want2.view(np.uint64)), (got2 - want2)
    if n == m:                                                           # one to one, as get_similar_transform is called directly
        got3 = g.K.similar_moments(mov, fix, None, mov_sequential=False, fix_sequential=False).cpu().numpy()
        assert np.array_equal(got3.view(np.uint64), reference_numbers(mv, fx).view(np.uint64))


@pytest.mark.parametrize("n", [2, 9, 150, 8193, 50000])      # (a single point makes np.matmul a matrix-vector product: another BLAS kernel, and a meaningless fit)
def test_application_and_residual_equal_numpy_bit_for_bit(g, n):
    from platymatch_amd.estimate_transform.find_transform import apply_affine_host
    rng = np.random.default_rng(100 + n)
    mv = np.ascontiguousarray(rng.normal(size=(3, n)) * 50 + 200.0)
    fx = np.ascontiguousarray(rng.normal(size=(3, n + 3)) * 50 + 200.0)
    nn = rng.integers(0, n + 3, size=n).astype(np.int32)
    A = np.eye(4)
    A[:3, :3] = 1.01 * np.linalg.qr(rng.normal(size=(3, 3)))[0]
    A[:3, 3] = rng.normal(size=3) * 5
    moved = apply_affine_host(mv, A)                                     # np.vstack + np.matmul, apply_transform.py:14-17
    res = np.mean(np.linalg.norm(moved - fx[:, nn], axis=0))             # get_error, utils.py:77-88
    mov = g.nat.to_dev(mv, dev=g.dev)
    r = g.K.similar_apply(g.nat.to_dev(A, dev=g.dev).reshape(16), mov, g.nat.to_dev(fx, dev=g.dev), g.nat.to_dev(nn, dtype=g.t.int32, dev=g.dev))
    assert np.array_equal(mov.cpu().numpy().view(np.uint64), np.ascontiguousarray(moved).view(np.uint64))
    assert float(r.item()) == float(res)


def test_quaternion_matrix_and_fit_equal_the_oracle_on_the_reference_fixture(g, oracle):
    """similar_mode.npz (produced by the unmodified reference): the 4 x 4 matrix N of the first ICP iteration bit for bit, the fit
    of every iteration equal to the oracle's fit of the same clouds, and the chain as a whole."""
    from platymatch_amd.estimate_transform import perform_icp as pi
    from platymatch_amd.estimate_transform.find_transform import quaternion_matrix_from_moments, similar_from_moments
    d = np.load(os.path.join(GOLDEN, "similar_mode.npz"))
    mv, fx = d["moving"], d["fixed"]
    start = oracle.apply_affine_transform(mv, d["ransac_A_k4"])
    mov, fix = g.nat.to_dev(np.ascontiguousarray(start), dev=g.dev), g.nat.to_dev(fx, dev=g.dev)
    for it in range(6):
        nn = g.K.icp_nn(mov, fix, want_dist=False)[0]
        mh, i2 = mov.cpu().numpy(), nn.cpu().numpy()
        want = reference_numbers(mh, fx[:, i2])
        got = g.K.similar_moments(mov, fix, nn).cpu().numpy()
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), it
        N_want = quaternion_matrix_from_moments(want)
        assert np.array_equal(quaternion_matrix_from_moments(got).view(np.uint64), N_want.view(np.uint64))
        A_dev = similar_from_moments(got)
        A_or = oracle.get_similar_transform(mh, fx[:, i2])
        assert np.array_equal(A_dev, A_or), it                             # same NumPy calls on the same bits
        g.K.similar_apply(g.nat.to_dev(A_dev, dev=g.dev).reshape(16), mov, fix, nn)
    log, olog = {}, {}
    A = pi.perform_icp(start, fx, 12, "Similar", log=log)
    A_ref = oracle.perform_icp(start, fx, 12, "Similar", log=olog)
    assert np.array_equal(log["nn"], olog["nn"])
    assert np.array_equal(A, A_ref) and np.array_equal(log["residuals"], olog["residuals"])
    assert np.array_equal(np.asarray(log["moved"]), np.asarray(olog["moved"])) if "moved" in olog else True


def test_similar_icp_at_20k_points_follows_the_oracle_and_keeps_the_cloud_on_the_device(g, oracle):
    from platymatch_amd.estimate_transform import perform_icp as pi
    mv, fx, _ = synth_pair(20000, 8, sigma=0.5)
    th = 0.02
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    start = np.ascontiguousarray(1.01 * R @ (fx - fx.mean(1, keepdims=True)) + fx.mean(1, keepdims=True) + 0.5)
    log, olog = {}, {}
    A = pi.perform_icp(start, fx, 5, "Similar", log=log)
    A_ref = oracle.perform_icp(start, fx, 5, "Similar", log=olog)
    assert np.array_equal(log["nn"], olog["nn"])
    assert np.array_equal(A, A_ref) and np.array_equal(log["residuals"], olog["residuals"])
    # a Fortran-ordered moving array: the first iteration's centroid is summed column by column, as np.mean does it
    A_f = pi.perform_icp(np.asfortranarray(start), fx, 3, "Similar")
    assert np.array_equal(A_f, oracle.perform_icp(np.asfortranarray(start), fx, 3, "Similar"))


def test_get_similar_transform_by_the_device_route_equals_the_literal_host_sequence(g, oracle):
    """The mirror's get_similar_transform on large NumPy clouds and on GPU tensors (O(N) sums on the device) against the oracle's
    literal NumPy sequence: identical 4 x 4, for C-ordered and Fortran-ordered inputs (np.mean sums them differently)."""
    from platymatch_amd.estimate_transform import find_transform as ft
    rng = np.random.default_rng(5)
    n = 20000
    mv = np.ascontiguousarray(rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0)
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    fx = np.ascontiguousarray(1.03 * R @ mv + rng.normal(scale=0.5, size=(3, n)) + 7.0)
    assert n >= ft.DEVICE_MOMENTS_FROM
    for a, b in ((mv, fx), (np.asfortranarray(mv), fx), (mv, np.asfortranarray(fx))):
        assert np.array_equal(ft.get_similar_transform(a, b), oracle.get_similar_transform(a, b))
    got = ft.get_similar_transform(g.nat.to_dev(mv, dev=g.dev), g.nat.to_dev(fx, dev=g.dev))
    assert np.array_equal(got.cpu().numpy(), oracle.get_similar_transform(mv, fx))
    small = slice(0, 500)                                                 # below the threshold: the literal host sequence
    assert np.array_equal(ft.get_similar_transform(mv[:, small], fx[:, small]), oracle.get_similar_transform(mv[:, small], fx[:, small]))
