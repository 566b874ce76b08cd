"""Point-cloud helpers with the reference's names and conventions (platymatch/utils/utils.py).

Arrays may be NumPy or torch (any device); results come back as the same kind.  All arithmetic
runs in the HIP kernels behind libplatymatch_hip.so.  The reference's GUI helpers
(_visualize_nuclei, _browse_detections, _browse_transform: utils.py:5-43) are out of scope.
"""
from .. import _kernels as K
from .. import _native as nat


def _cloud_3xn(detections, transposed):
    """-> GPU float64 [3, N]; a 4th row/column (ids, radii) is dropped as the reference does
    (utils.py:52-56, 69-70)."""
    t = nat.to_dev(detections)
    if t.dim() != 2:
        raise ValueError("detections must be 2-D")
    if transposed:
        t = t.t()
    if t.shape[0] not in (3, 4):
        raise ValueError("expected 3 (or 4) coordinate rows, got %d" % t.shape[0])
    return t[:3, :].contiguous()


def _mean_runs_sequentially(detections, transposed):
    """How np.mean adds the points up depends on the memory layout: NumPy's reduction loops along the axis with the smaller
    stride; if that is the point axis the sum is pairwise (pieces of 8 192, csrc/pm_pairwise.h), otherwise the points are added
    one after the other.  NumPy input: its strides decide; torch tensors count as row-major."""
    import numpy as np
    if nat.is_torch(detections):
        s0, s1 = detections.stride(0), detections.stride(1)
    else:
        a = np.asarray(detections)
        s0, s1 = abs(a.strides[0]), abs(a.strides[1])
    point_stride, coord_stride = (s0, s1) if transposed else (s1, s0)
    return not (point_stride <= coord_stride)


def get_centroid(detections, transposed=True):
    """utils/utils.py:48-56 -> [1, 3] if transposed else [3, 1].  The reference's bits: np.mean's own summation order (pairwise
    for the 3 x N layout the widget passes, point after point for N x 3)."""
    c = K.centroid(_cloud_3xn(detections, transposed), sequential=_mean_runs_sequentially(detections, transposed))
    c = c.reshape(1, 3) if transposed else c.reshape(3, 1)
    return nat.like_input(c, detections)


def get_mean_distance(detections, transposed=True):
    """utils/utils.py:58-75 -> scalar mean of all N(N-1)/2 pairwise distances."""
    md = K.mean_distance(_cloud_3xn(detections, transposed))
    return md[0] if nat.is_torch(detections) else float(md.item())


def get_error(moving_landmarks, fixed_landmarks):
    """utils/utils.py:77-88 -> mean column norm of (moving - fixed); None if both inputs are None."""
    if moving_landmarks is None and fixed_landmarks is None:
        return None
    a = nat.to_dev(moving_landmarks)
    b = nat.to_dev(fixed_landmarks)
    if a.shape != b.shape or a.dim() != 2 or a.shape[0] != 3:
        raise ValueError("landmarks must both be 3 x N")
    e = K.get_error(a.contiguous(), b.contiguous())
    return e[0] if nat.is_torch(moving_landmarks) else float(e.item())
