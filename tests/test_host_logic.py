"""Host-side logic of the product that needs no GPU."""
import os

import numpy as np

from conftest import load_golden


def test_shard_bounds():
    from platymatch_amd.pipeline import shard_bounds
    assert shard_bounds(10, 1) == [0, 10]
    assert shard_bounds(10, 3) == [0, 4, 7, 10]
    assert shard_bounds(2, 4) == [0, 1, 2, 2, 2]
    b = shard_bounds(50000, 8)
    assert b[-1] == 50000 and all(b[i + 1] - b[i] == 6250 for i in range(8))


def test_ransac_draws_follow_the_reference_rng_sequence(oracle):
    """do_ransac draws np.random.choice(n, 4, replace=False) once per trial from the global RNG
    (shape_context.py:122): the product's host-side draw must be that exact call sequence."""
    from platymatch_amd.estimate_transform.shape_context import draw_ransac_samples
    np.random.seed(0)
    ours = draw_ransac_samples(331, 4, 50)
    np.random.seed(0)
    ref = np.stack([np.random.choice(331, 4, replace=False) for _ in range(50)])
    assert np.array_equal(ours, ref) and ours.dtype == np.int32
    np.random.seed(0)
    assert np.array_equal(oracle.draw_ransac_samples(331, 4, 50), ref)
    # the C reproduction of the draws: same sets for many sizes, and the global stream continues where NumPy's would
    for n in (4, 5, 7, 128, 343, 4096, 5000, 65536, 70001):
        np.random.seed(n)
        want = np.stack([np.random.choice(n, 4, replace=False) for _ in range(40)])
        after_want = np.random.random(3)
        np.random.seed(n)
        got = draw_ransac_samples(n, 4, 40)
        assert np.array_equal(got, want) and np.array_equal(np.random.random(3), after_want), n
    np.random.seed(5)
    a = draw_ransac_samples(100, 4, 10)
    b = draw_ransac_samples(100, 4, 10)          # second call continues the stream: eight hypotheses draw one after another
    np.random.seed(5)
    assert np.array_equal(np.concatenate([a, b]), np.stack([np.random.choice(100, 4, replace=False) for _ in range(20)]))
    # a private RandomState(seed) gives what np.random.seed(seed) + the global generator gives, and leaves the latter alone;
    # sizes chosen so that draws start at every kind of position inside a 624-word state block, 1, 2 and 3 also as k
    np.random.seed(77)
    before = np.random.get_state()
    for n, k in ((1, 1), (2, 2), (3, 3), (9, 4), (1023, 4), (1024, 4), (1025, 3), (33000, 4)):
        rng = np.random.RandomState(n)
        got = np.concatenate([draw_ransac_samples(n, k, 7, rng=rng) for _ in range(3)])
        np.random.seed(n)
        want = np.stack([np.random.choice(n, k, replace=False) for _ in range(21)])
        assert np.array_equal(got, want), (n, k)
        np.random.set_state(before)
    import pytest
    with pytest.raises(ValueError):
        draw_ransac_samples(3, 4, 2)            # fewer pairs than samples: NumPy's own error, as in the reference


def test_similar_transform_host_is_the_reference_to_the_bit(oracle):
    """get_similar_transform's host restatement (row-0 eigenvector quirk and all, find_transform.py:21-99) against the
    reference's own outputs: micro.npz (C-ordered inputs) and similar_mode.npz (do_ransac's fancy-indexed samples)."""
    from conftest import GOLDEN
    from platymatch_amd.estimate_transform.find_transform import apply_affine_host, get_similar_transform, similar_transform_host
    d = load_golden("micro")
    P, Q = d["fit_moving"], d["fit_fixed"]
    assert np.array_equal(similar_transform_host(P, Q), d["fit_similar"])
    assert np.array_equal(get_similar_transform(P[:, :4], Q[:, :4]), d["fit_similar4"])          # NumPy in, NumPy out, no GPU involved
    s = np.load(os.path.join(GOLDEN, "similar_mode.npz"))
    mv, fx = s["moving"], s["fixed"]
    for k in (4, 6, 9, 20):
        got = np.stack([similar_transform_host(mv[:, i], fx[:, i]) for i in s["samples_k%d" % k]])
        assert np.array_equal(got, s["fits_k%d" % k]), k
    assert np.array_equal(apply_affine_host(mv, s["ransac_A_k4"]), oracle.apply_affine_transform(mv, s["ransac_A_k4"]))


def test_legacy_get_bin_index_helper(micro):
    """get_bin_index on explicit (r, theta, phi) lists (shape_context.py:46-58)."""
    from platymatch_amd.estimate_transform.shape_context import get_bin_index, get_Y
    nb = micro["rand_neighbors"]
    r = np.linalg.norm(nb, axis=1) / 55.0
    th = np.arccos(nb[:, 2] / np.linalg.norm(nb, axis=1))
    at = np.arctan2(nb[:, 1], nb[:, 0])
    ph = np.where(at < 0, 2 * np.pi + at, at)
    got = np.array(get_bin_index(list(r), list(th), list(ph), micro["r_edges"], 5, 6, 12))
    assert np.array_equal(got, micro["rand_bin_index"])
    y = get_Y(np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    assert np.allclose(y, [[0.0, 1.0, 0.0]])


def test_an_empty_neighbour_list_gives_the_references_all_nan_histogram():
    """shape_context.py:25-41 with no neighbour: the loops run zero times and sc / sc.sum() is 0 / 0 in every bin (what the unmodified
    reference returns: a (360,) array of NaN); no device is involved."""
    from platymatch_amd.estimate_transform.shape_context import get_shape_context
    for empty in (np.zeros((0, 3)), []):
        sc = get_shape_context(empty, 1.0)
        assert isinstance(sc, np.ndarray) and sc.shape == (360,) and np.isnan(sc).all()
    assert get_shape_context([], 2.0, n_rbins=3, n_thetabins=4, n_phibins=5).shape == (60,)


def test_install_as_platymatch_aliases():
    import sys
    import platymatch_amd
    saved = {k: v for k, v in sys.modules.items() if k == "platymatch" or k.startswith("platymatch.")}
    for k in saved:
        del sys.modules[k]
    try:
        platymatch_amd.install_as_platymatch()
        from platymatch.estimate_transform.shape_context import get_unary, get_unary_distance, do_ransac  # noqa: F401
        from platymatch.estimate_transform.perform_icp import perform_icp  # noqa: F401
        from platymatch.estimate_transform.find_transform import get_affine_transform, get_similar_transform  # noqa: F401
        from platymatch.estimate_transform.apply_transform import apply_affine_transform  # noqa: F401
        from platymatch.utils.utils import get_centroid, get_mean_distance  # noqa: F401
        import platymatch_amd.estimate_transform.shape_context as sc
        assert get_unary is sc.get_unary
    finally:
        for k in [k for k in sys.modules if k == "platymatch" or k.startswith("platymatch.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_similar_ransac_fits_follow_the_reference_bit_order():
    """transform='Similar' RANSAC fits (similar_mode.npz: the reference's get_similar_transform on do_ransac's fancy-indexed
    samples).  The 4 x 4 eigen-problem must be the reference's to the bit or LAPACK's eigenvector signs flip the result
    (a few per cent of the trials with moments summed in any other order); after it, agreement is to rounding."""
    from conftest import GOLDEN
    from platymatch_amd.estimate_transform.find_transform import similar_fit_batch, _numpy_sum_rows
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 8, 9, 16, 31, 127, 128, 129, 136, 255, 257, 1000):
        x = rng.normal(size=(40, n)) * 1e3
        assert np.array_equal(_numpy_sum_rows(x), np.array([np.sum(r) for r in x])), n        # NumPy's own 1-D order
    d = np.load(os.path.join(GOLDEN, "similar_mode.npz"))
    mv, fx = d["moving"], d["fixed"]
    for k in (4, 6, 9, 20):
        S = d["samples_k%d" % k]
        got = similar_fit_batch(np.moveaxis(mv[:, S], 0, 1), np.moveaxis(fx[:, S], 0, 1))
        ref = d["fits_k%d" % k]
        err = np.abs(got - ref).reshape(len(S), -1).max(1) / np.abs(ref).reshape(len(S), -1).max(1)
        assert err.max() < 1e-12, (k, err.max())
        assert np.array_equal(similar_fit_batch(np.moveaxis(mv[:, S[:1]], 0, 1), np.moveaxis(fx[:, S[:1]], 0, 1))[0], got[0])


def test_unary_array_rows_remember_their_origin_and_nothing_else_does():
    """get_unary's NumPy return type (shape_context.UnaryArray): an ordinary float64 ndarray whose INTEGER row views carry
    (descriptor set, frame, row) for get_unary_distance's table lookup; every derived array is plain data again."""
    import pickle
    from platymatch_amd.estimate_transform.shape_context import UnaryArray, _DescriptorSet
    host = np.random.default_rng(0).random((2, 6, 360))
    dset = _DescriptorSet(None, host)
    a = host[1].view(UnaryArray)
    a._pm_set, a._pm_frame = dset, 1
    assert isinstance(a, np.ndarray) and a.shape == (6, 360) and a.dtype == np.float64
    r = a[4]
    assert (r._pm_set is dset, r._pm_frame, r._pm_row) == (True, 1, 4) and np.array_equal(r, host[1, 4])
    assert a[-1]._pm_row == 5 and a[np.int64(2)]._pm_row == 2
    assert [row._pm_row for row in a] == list(range(6))                       # iteration yields tagged rows
    for derived in (a[1:3], a[1:3][0], a + 0.0, a * 2, a.copy(), a[4].copy(), a[4][10:20], a.T, a[[1, 2]], np.array(a), a.astype(np.float32)):
        assert getattr(derived, "_pm_set", None) is None and getattr(derived, "_pm_row", -1) < 0
    assert isinstance(a[4][7], np.float64) and float(a.sum()) == float(host[1].sum())
    assert type(pickle.loads(pickle.dumps(a))) is np.ndarray and np.array_equal(pickle.loads(pickle.dumps(a)), host[1])


def test_batch_cost_model_orders_pairs_largest_first():
    from platymatch_amd import pipeline as P
    sizes = [(2000, 2000), (20000, 19000), (5000, 9000), (3000, 3000)]
    cost = P.batch_costs(sizes)
    assert int(np.argmax(cost)) == 1 and cost[2] > cost[3] > cost[0]
    assert P.batch_assignment(sizes, 1) == [0, 0, 0, 0]
    owner = P.batch_assignment(sizes, 2)
    assert owner[1] != owner[2]                                              # the two largest pairs go to different ranks


def test_matrix_memory_policy_on_the_host_side():
    """device_memory.big_empty: below BIG_BLOCK_BYTES (and for anything that is not a GPU tensor) torch's own allocation, whatever
    the policy switch says; the batch switch nests and is per thread; nothing idle without a GPU."""
    import threading
    import torch
    from platymatch_amd import device_memory as D
    assert D.BIG_BLOCK_BYTES == 4 << 30 and 0.0 < D.MAX_IDLE_FRACTION < 1.0
    t = D.big_empty((3, 5), torch.float32, "cpu")
    assert tuple(t.shape) == (3, 5) and t.dtype == torch.float32 and t.device.type == "cpu"
    assert D.big_empty(7, torch.float64, "cpu").shape == (7,)
    assert D.idle_bytes() == 0 and D.trim() == 0
    seen = []
    with D.through_torch():
        with D.through_torch():
            seen.append(D._THREAD.depth)
            th = threading.Thread(target=lambda: seen.append(getattr(D._THREAD, "depth", 0)))
            th.start()
            th.join()
        seen.append(D._THREAD.depth)
    seen.append(D._THREAD.depth)
    assert seen == [2, 0, 1, 0]
