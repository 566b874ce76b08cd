"""Matching accuracy and average registration error: the arithmetic of
EvaluateMetrics._calculate_metrics (platymatch/_dock_widget.py:1030-1080) as a function.

The three scipy `cdist` matrices are built on the device (pm_cdist, an HBM-write-bound kernel), the three
`linear_sum_assignment` solves stay on the host (lsap.py: SciPy's solver restated, identical indices), the transforms are
applied on the device.  Arrays are 3 x N float64 (z, y, x); ids are 1-D arrays as `_browse_detections` returns them."""
import numpy as np

from . import _kernels as K
from . import _native as nat
from .lsap import linear_sum_assignment


def cdist(a, b):
    """scipy.spatial.distance.cdist(a.T, b.T) for 3 x n, 3 x m clouds -> (n, m); same kind of array as `a`."""
    ta, tb = nat.to_dev(a), nat.to_dev(b)
    if ta.dim() != 2 or tb.dim() != 2 or ta.shape[0] != 3 or tb.shape[0] != 3:
        raise ValueError("clouds must be 3 x N")
    return nat.like_input(K.cdist(ta.contiguous(), tb.contiguous()), a)


def calculate_metrics(moving_keypoints, moving_keypoint_ids, moving_detections, moving_ids, fixed_keypoints,
                      fixed_keypoint_ids, fixed_detections, fixed_ids, transform_matrix_1, transform_matrix_2=None):
    """_dock_widget.py:1030-1080 -> (matching_accuracy, average_registration_error) as floats (the widget shows them
    with three decimals).  transform_matrix_2 defaults to identity; the combined transform is T2 @ T1 (:1027)."""
    moving_keypoint_ids, moving_ids = np.asarray(moving_keypoint_ids), np.asarray(moving_ids)
    fixed_keypoint_ids, fixed_ids = np.asarray(fixed_keypoint_ids), np.asarray(fixed_ids)
    T1 = np.asarray(transform_matrix_1, dtype=np.float64)
    T2 = np.eye(4) if transform_matrix_2 is None else np.asarray(transform_matrix_2, dtype=np.float64)
    mk, md = nat.to_dev(moving_keypoints)[:3].contiguous(), nat.to_dev(moving_detections)[:3].contiguous()
    fk, fd = nat.to_dev(fixed_keypoints)[:3].contiguous(), nat.to_dev(fixed_detections)[:3].contiguous()
    dev = mk.device

    # keypoints -> detections, per image (:1032-1042)
    r, c = linear_sum_assignment(K.cdist(mk, md).cpu().numpy())
    moving_dictionary = {moving_keypoint_ids[i]: moving_ids[c[i]] for i in r}
    r, c = linear_sum_assignment(K.cdist(fk, fd).cpu().numpy())
    fixed_dictionary = {fixed_keypoint_ids[i]: fixed_ids[c[i]] for i in r}

    # transformed moving detections <-> fixed detections (:1044-1051)
    moved = K.apply_affine(nat.to_dev(T1, dev=dev).reshape(16), md)
    moved = K.apply_affine(nat.to_dev(T2, dev=dev).reshape(16), moved)
    row_indices, col_indices = linear_sum_assignment(K.cdist(moved, fd).cpu().numpy())
    row_ids, col_ids = moving_ids[row_indices], fixed_ids[col_indices]

    hits = 0                                                                  # :1057-1062
    for key, det_id in moving_dictionary.items():
        if key in fixed_dictionary:
            matched = col_ids[np.where(row_ids == det_id)]
            if matched.size == 1 and matched[0] == fixed_dictionary[key]:
                hits += 1
    accuracy = hits / len(fixed_dictionary)                                   # :1067 (normalised by the fixed keypoints)

    # average registration error over the keypoints (:1071-1079)
    combined = nat.to_dev(np.matmul(T2, T1), dev=dev).reshape(16)
    tmk = K.apply_affine(combined, mk).cpu().numpy()
    fk_h = fk.cpu().numpy()
    distance = 0.0
    for i in range(tmk.shape[1]):
        sel = np.where(fixed_keypoint_ids == moving_keypoint_ids[i])[0]
        distance += np.linalg.norm(fk_h[:, sel] - tmk[:, i:i + 1])
    return accuracy, distance / len(moving_dictionary)
