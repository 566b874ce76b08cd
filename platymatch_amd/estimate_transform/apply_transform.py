"""apply_affine_transform / apply_similar_transform (platymatch/estimate_transform/apply_transform.py)."""
from .. import _kernels as K
from .. import _native as nat


def apply_affine_transform(moving, affine_transform_matrix):
    """apply_transform.py:3-17: (A . [moving; 1])[:3]; a 4 x N input loses its 4th row first."""
    m = nat.to_dev(moving)
    if m.dim() != 2 or m.shape[0] not in (3, 4):
        raise ValueError("moving must be 3 x N (or 4 x N)")
    A = nat.to_dev(affine_transform_matrix, dev=m.device)
    if tuple(A.shape) != (4, 4):
        raise ValueError("affine_transform_matrix must be 4 x 4")
    out = K.apply_affine(A.reshape(16), m[:3, :].contiguous())
    return nat.like_input(out, moving)


def apply_similar_transform(source, scale, rotation, translation, with_ones=False):
    """apply_transform.py:19-33: scale * R . source + t (unused by the reference's own pipeline)."""
    torch = nat.torch_mod()
    s = nat.to_dev(source)
    if with_ones:
        s = s[:3, :]
    R = nat.to_dev(rotation, dev=s.device)
    t = nat.to_dev(translation, dev=s.device).reshape(3)
    A = torch.zeros((4, 4), dtype=torch.float64, device=s.device)
    A[:3, :3] = float(scale) * R
    A[:3, 3] = t
    A[3, 3] = 1.0
    out = K.apply_affine(A.reshape(16), s.contiguous())
    return nat.like_input(out, source)
