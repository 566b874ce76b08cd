"""On-disk formats: the reference's own asset files and np.savetxt text are the known answers."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from platymatch_amd import io as pmio


def test_read_reference_assets():
    """02-insitu.csv / 04-insitu.csv are the reference's test assets (id x y z, no header): reading them must give the
    clouds the reference's tests build (and the fixtures were generated from)."""
    det, ids = pmio.read_detections(os.path.join(GOLDEN, "02-insitu.csv"))
    assert det.shape == (3, 331) and ids.shape == (331,)
    assert np.array_equal(det, load_golden("insitu02_identity")["moving"])
    det4, _ = pmio.read_detections(os.path.join(GOLDEN, "04-insitu.csv"))
    assert np.array_equal(det4, load_golden("insitu04_affine")["moving"])
    raw = np.loadtxt(os.path.join(GOLDEN, "02-insitu.csv"))
    assert np.array_equal(det[::-1].T, raw[:, 1:4]) and np.array_equal(ids, raw[:, 0])     # x y z flipped to z y x


def test_detections_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    zyx = rng.uniform(0, 300, size=(17, 3))
    radii = rng.uniform(3, 9, size=17)
    p = str(tmp_path / "det.csv")
    pmio.write_detections(p, zyx, radii, anisotropy=2.5)
    first = open(p).readline().strip()
    assert first == "id dimension_z dimension_y dimension_x radius"                      # _dock_widget.py:165
    det, ids = pmio.read_detections(p, header=True, izyxr=True)
    want = zyx.copy()
    want[:, 0] *= 2.5
    # pandas' default float parser (what the reference reads with) is within one ulp, not always exact
    assert (np.abs(det - want.T) <= np.spacing(want.T)).all() and np.array_equal(ids, np.arange(1, 18))
    det_flipped, _ = pmio.read_detections(p, header=True, izyxr=False)
    assert np.array_equal(det_flipped, det[::-1])
    with pytest.raises(ValueError):
        pmio.write_detections(p, zyx[:, :2], radii)


def test_transform_files(tmp_path):
    d = load_golden("insitu02_affine")
    p = str(tmp_path / "t.txt")
    pmio.save_transform(p, d["A_icp"], d["A_sc"])
    q = str(tmp_path / "ref.txt")
    np.savetxt(q, np.matmul(d["A_icp"], d["A_sc"]), delimiter=' ', fmt='%1.3f')             # _dock_widget.py:428-432
    assert open(p).read() == open(q).read()
    A = pmio.read_transform(p)
    assert A.shape == (4, 4) and np.abs(A - d["A_final"]).max() <= 5e-4                       # three decimals survive
    np.savetxt(p, np.eye(3), delimiter=' ', fmt='%1.3f')
    with pytest.raises(AssertionError):
        pmio.read_transform(p)
    pmio.save_pca_transforms(str(tmp_path), np.eye(3), 2 * np.eye(3))
    assert np.array_equal(np.loadtxt(str(tmp_path / "fixed_transform.txt")), 2 * np.eye(3))
