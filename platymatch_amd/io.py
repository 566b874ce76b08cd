"""On-disk formats of the reference, headless.

The reference reads and writes these files from Qt dialogs; the functions here take paths and otherwise follow the
same conventions, so files produced by either side load in the other:
  detections CSV   read: `_browse_detections` (platymatch/utils/utils.py:19-34); write: `_export_detections`
                   (platymatch/_dock_widget.py:151-172)
  transform text   write: `_save_transform` (_dock_widget.py:426-438); read: `_browse_transform` (utils.py:37-43)
Pure host I/O: nothing here touches the GPU.
"""
import csv

import numpy as np
import pandas as pd


def read_detections(path, header=False, izyxr=False):
    """utils.py:19-34.  Space-delimited rows `id a b c [radius]`; `header=True` skips the first row (the widget's
    "header" tick box); columns 1..3 are (x, y, z) and are flipped to (z, y, x) unless `izyxr=True` (the "IZYXR"
    tick box: the file is already id, z, y, x, radius — what `write_detections` produces).
    -> (detections 3 x N float64 rows z, y, x; ids N)."""
    df = pd.read_csv(path, skiprows=[0] if header else None, delimiter=' ', header=None)
    arr = df.to_numpy()
    ids = arr[:, 0]
    det = arr[:, 1:4].astype(np.float64)
    if not izyxr:
        det = np.flip(det, 1)
    return np.ascontiguousarray(det.transpose()), ids.transpose()


def write_detections(path, nuclei_zyx, radii, anisotropy=1.0):
    """_dock_widget.py:151-172.  nuclei_zyx: N x 3 (z, y, x) in voxel units, radii: N.  Writes the header
    `id dimension_z dimension_y dimension_x radius`, ids counted from 1, z multiplied by `anisotropy`; values are
    written by csv.writer exactly as the reference writes its float row (so ids appear as `1.0`, `2.0`, ...)."""
    nuclei = np.asarray(nuclei_zyx, dtype=np.float64)
    radii = np.asarray(radii)
    if nuclei.ndim != 2 or nuclei.shape[1] != 3 or radii.shape[0] != nuclei.shape[0]:
        raise ValueError("nuclei must be N x 3 (z, y, x) with one radius each")
    with open(path, mode='w') as fh:
        writer = csv.writer(fh, delimiter=' ', quotechar='"', quoting=csv.QUOTE_MINIMAL)
        writer.writerow(['id', 'dimension_z', 'dimension_y', 'dimension_x', 'radius'])
        for idx, row in enumerate(nuclei):
            row_ = row.copy()
            row_[0] = float(anisotropy) * row_[0]
            writer.writerow(np.concatenate([np.array([int(idx) + 1]), row_, np.array([radii[idx]])]))


def save_transform(path, transform_matrix_icp, transform_matrix_sc=None):
    """_dock_widget.py:426-432: the exported matrix is A_icp @ A_sc, written with np.savetxt(fmt='%1.3f')
    (three decimals, as the reference).  Pass a single 4 x 4 to write it as is."""
    A = np.asarray(transform_matrix_icp, dtype=np.float64)
    if transform_matrix_sc is not None:
        A = np.matmul(A, np.asarray(transform_matrix_sc, dtype=np.float64))
    if A.shape != (4, 4):
        raise ValueError("transform must be 4 x 4")
    np.savetxt(path, A, delimiter=' ', fmt='%1.3f')


def save_pca_transforms(directory, moving_transform, fixed_transform):
    """_dock_widget.py:433-438 (PCA-only mode): two 3 x 3 text files in `directory`."""
    np.savetxt(directory + '/moving_transform.txt', np.asarray(moving_transform), delimiter=' ', fmt='%1.3f')
    np.savetxt(directory + '/fixed_transform.txt', np.asarray(fixed_transform), delimiter=' ', fmt='%1.3f')


def read_transform(path):
    """utils.py:37-43 -> 4 x 4 float64; anything else raises AssertionError like the reference."""
    A = pd.read_csv(path, skiprows=None, delimiter=' ', header=None).to_numpy().astype(np.float64)
    assert A.shape == (4, 4), 'Loaded transform does not have shape 4 x 4'
    return A
