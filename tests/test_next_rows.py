"""SURVEY.md §8f rows widened into so far: PCA-only alignment, evaluation metrics, label-image centroids.
CPU part: the oracle against reference-generated known answers (tests/golden/next_rows.npz).
GPU part (-m gpu): the HIP path against the oracle and the same known answers."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def nx():
    return load_golden("next_rows")


def metrics_args(d):
    return (d["ev_moving_kp"], d["ev_moving_kp_ids"], d["ev_moving"], d["ev_moving_ids"], d["ev_fixed_kp"], d["ev_fixed_kp_ids"],
            d["ev_fixed"], d["ev_fixed_ids"], d["ev_T1"], d["ev_T2"])


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_oracle_pca_components(oracle, nx):
    for t in ("02", "04"):
        X = nx["pca_cloud_" + t]
        centred = X - oracle.get_centroid(X, False)                       # the widget centres the clouds first (_dock_widget.py:723-724)
        assert np.array_equal(oracle.pca_components(centred.transpose()), nx["pca_components_" + t])      # sklearn's bits (round 4)
        assert np.abs(oracle.pca_components(X.T) - nx["pca_components_" + t]).max() < 1e-12


def test_pca_alignment_is_sklearns_bit_for_bit(oracle, nx):
    """pipeline.pca_alignment (the widget's PCA-only branch, _dock_widget.py:722-731): sklearn's own NumPy calls on the caller's
    arrays, host code by design (the axes hang on BLAS / LAPACK rounding) — the reference's 3 x 3 matrices to the bit."""
    from platymatch_amd.pipeline import pca_alignment
    c02 = nx["pca_cloud_02"] - oracle.get_centroid(nx["pca_cloud_02"], False)
    c04 = nx["pca_cloud_04"] - oracle.get_centroid(nx["pca_cloud_04"], False)
    mt, ft = pca_alignment(c02, c04)
    assert np.array_equal(mt, nx["pca_components_02"]) and np.array_equal(ft, nx["pca_components_04"])
    assert mt.shape == (3, 3) and np.allclose(mt @ mt.T, np.eye(3), atol=1e-13)


def test_oracle_metrics_and_cdist(oracle, nx):
    acc, err = oracle.calculate_metrics(*metrics_args(nx))
    assert acc == nx["ev_accuracy"] and err == nx["ev_registration_error"]
    moved = oracle.apply_affine_transform(oracle.apply_affine_transform(nx["ev_moving"], nx["ev_T1"]), nx["ev_T2"])
    assert np.array_equal(oracle.cdist(moved, nx["ev_fixed"])[::20], nx["ev_cdist_rows"])


def test_oracle_label_centroids(oracle, nx):
    c, s, ids = oracle.label_centroids(nx["lab_image"], 2.0)
    assert np.array_equal(c, nx["lab_centroids"]) and np.array_equal(s, nx["lab_sizes_aniso2"]) and np.array_equal(ids, nx["lab_ids"])
    assert oracle.ransac_error_from_sizes([], [1.0]) == 16
    assert oracle.ransac_error_from_sizes([8.0, 8.0], [27.0]) == 0.5 * (8.0 ** (1 / 3) + 27.0 ** (1 / 3))


# ------------------------------------------------------------------------------------------------ HIP path (GPU)
@pytest.mark.gpu
def test_gpu_pca_components_kernel(oracle, nx):
    """pm_pca_components (the C-ABI entry; the mirror itself takes the host route above): within 1e-11 of sklearn."""
    from platymatch_amd import _kernels as K, _native as nat
    for t in ("02", "04"):
        got = K.pca_components(nat.to_dev(np.ascontiguousarray(nx["pca_cloud_" + t][:3]))).cpu().numpy()
        assert np.abs(got - nx["pca_components_" + t]).max() < 1e-11
        assert np.allclose(got @ got.T, np.eye(3), atol=1e-13)


@pytest.mark.gpu
def test_gpu_cdist_bit_exact_and_ragged(oracle, nx):
    from platymatch_amd import _kernels as K, _native as nat
    from platymatch_amd.evaluate_metrics import cdist
    rng = np.random.default_rng(4)
    for n, m in ((1, 1), (14, 200), (33, 513), (200, 200), (1000, 1025), (31, 7)):
        a, b = rng.normal(size=(3, n)) * 80, rng.normal(size=(3, m)) * 80
        assert np.array_equal(cdist(a, b), oracle.cdist(a, b))                   # scipy's operation order, correctly rounded sqrt
    moved = oracle.apply_affine_transform(oracle.apply_affine_transform(nx["ev_moving"], nx["ev_T1"]), nx["ev_T2"])
    assert np.array_equal(cdist(moved, nx["ev_fixed"])[::20], nx["ev_cdist_rows"])
    # odd leading dimension / unaligned views take the 8-byte store path and give the same numbers
    import torch
    a, b = nat.to_dev(rng.normal(size=(3, 50))), nat.to_dev(rng.normal(size=(3, 77)))
    big = torch.full((50, 101), -1.0, dtype=torch.float64, device=a.device)
    K.cdist(a, b, out=big[:, 3:80])
    assert torch.equal(big[:, 3:80], K.cdist(a, b)) and (big[:, :3] == -1).all() and (big[:, 80:] == -1).all()


@pytest.mark.gpu
def test_gpu_metrics(oracle, nx):
    from platymatch_amd.evaluate_metrics import calculate_metrics
    acc, err = calculate_metrics(*metrics_args(nx))
    assert acc == nx["ev_accuracy"]
    assert abs(err - nx["ev_registration_error"]) < 1e-12 * nx["ev_registration_error"]
    acc1, _ = calculate_metrics(*metrics_args(nx)[:8], np.matmul(nx["ev_T2"], nx["ev_T1"]))    # T2 defaults to identity
    assert 0.0 <= acc1 <= 1.0


@pytest.mark.gpu
def test_gpu_label_centroids(oracle, nx):
    from platymatch_amd.label_image import label_centroids, ransac_error_from_sizes
    c, s, ids = label_centroids(nx["lab_image"], 2.0)
    assert np.array_equal(c, nx["lab_centroids"]) and np.array_equal(s, nx["lab_sizes_aniso2"]) and np.array_equal(ids, nx["lab_ids"])
    rng = np.random.default_rng(9)
    vol = np.zeros((33, 65, 130), dtype=np.int32)                              # ragged row length, labels filling whole rows
    vol[5:20, 10:40, :] = 7
    vol[0, 0, :] = 2
    vol[32, 64, 129] = 900
    vol[10:12, 50:60, 100:130] = rng.integers(0, 5, size=(2, 10, 30)) * 11      # salt-and-pepper labels 11..44
    oc, osz, oid = oracle.label_centroids(vol, 1.0)
    gc, gsz, gid = label_centroids(vol, 1.0)
    assert np.array_equal(gc, oc) and np.array_equal(gsz, osz) and np.array_equal(gid, oid)
    assert ransac_error_from_sizes([], []) == 16
    assert ransac_error_from_sizes(gsz, osz) == oracle.ransac_error_from_sizes(gsz, osz)
    with pytest.raises(ValueError):
        label_centroids(np.zeros((4, 4)), 1.0)
    with pytest.raises(ValueError):
        label_centroids(np.full((2, 2, 2), -1, dtype=np.int32), 1.0)


@pytest.mark.gpu
def test_label_moments_exact_on_odd_volumes_blobs_and_noise():
    """pm_label_moments against NumPy's integer arithmetic (np.bincount with weights is float: use explicit int64 sums):
    unaligned row lengths, blobs (the per-brick label table's fast path), one label per voxel (the table overflows into the
    global accumulators), labels at the top of the range."""
    import torch
    from platymatch_amd import _kernels as K, _native as nat
    from platymatch_amd.build import build_native
    build_native()
    nat.load()
    rng = np.random.default_rng(4)
    for shape, kind in (((9, 13, 37), "noise"), ((20, 33, 130), "blobs"), ((5, 8, 16), "blobs"), ((33, 17, 257), "noise"), ((4, 4, 4), "one")):
        if kind == "noise":
            lab = rng.integers(0, 3000, size=shape).astype(np.int32)
        elif kind == "one":
            lab = np.full(shape, 7, dtype=np.int32)
        else:
            zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
            lab = (((zz // 5) * 100 + (yy // 6)) * 100 + xx // 7 + 1).astype(np.int32)
            lab[(zz + yy + xx) % 11 == 0] = 0
        n_labels = int(lab.max()) + 1
        counts, sums = K.label_moments(torch.as_tensor(lab, device="cuda"), n_labels=n_labels)
        zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
        want_c = np.bincount(lab.ravel(), minlength=n_labels).astype(np.int64)
        want_c[0] = 0                                                    # background is not accumulated
        assert np.array_equal(counts.cpu().numpy(), want_c), (shape, kind)
        for k, coord in enumerate((zz, yy, xx)):
            want = np.zeros(n_labels, dtype=np.int64)
            np.add.at(want, lab.ravel(), coord.ravel().astype(np.int64))
            want[0] = 0
            assert np.array_equal(sums[k].cpu().numpy(), want), (shape, kind, k)
