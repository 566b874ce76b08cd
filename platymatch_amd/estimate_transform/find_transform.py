"""get_affine_transform / get_similar_transform (platymatch/estimate_transform/find_transform.py)."""
import numpy as np

from .. import _kernels as K
from .. import _native as nat


def _pair(moving, fixed, drop_ones):
    m, f = nat.to_dev(moving), nat.to_dev(fixed)
    if m.dim() != 2 or f.dim() != 2 or m.shape[1] != f.shape[1]:
        raise ValueError("moving and fixed must be 2-D with the same number of points")
    rows = 4 if drop_ones else 3
    if m.shape[0] != rows or f.shape[0] != rows:
        raise ValueError("expected %d x N clouds" % rows)
    return m[:3, :].contiguous(), f[:3, :].contiguous()


def get_affine_transform(moving, fixed, with_ones=False):
    """find_transform.py:4-17: least-squares 4 x 4 with [fixed;1] = A [moving;1].

    The reference forms fixed . pinv(moving); for a full-rank cloud (>= 4 points, not coplanar)
    that is the least-squares solution, computed here on the device from centred moments
    (last row exactly 0 0 0 1, the reference's is 0 0 0 1 to ~1e-17).  Rank-deficient input,
    where pinv returns a minimum-norm answer, is not supported and raises ValueError."""
    m, f = _pair(moving, fixed, with_ones)
    if m.shape[1] < 4:
        raise ValueError("get_affine_transform needs at least 4 point pairs on the device path")
    A = K.fit_affine(m, f)
    if not bool(nat.torch_mod().isfinite(A).all()):
        raise ValueError("degenerate (coplanar or repeated) points: affine fit is rank deficient")
    return nat.like_input(A, moving)


def similar_from_sums(sums, origin6):
    """Horn's closed form (find_transform.py:27-99) from the 24 moment sums of pm_icp_accumulate.
    Tiny host step: a 4 x 4 eigen-decomposition.  Faithful to the reference's quirk at :60-66 —
    after sorting by eigenvalue, q is ROW 0 of the eigenvector matrix, not column 0 — so, as in the
    reference, the result depends on LAPACK's eigenvector signs."""
    s = np.asarray(sums, dtype=np.float64)
    o = np.asarray(origin6, dtype=np.float64)
    n = s[0]
    mb, fb = s[1:4] / n, s[4:7] / n
    Sfm = s[13:22].reshape(3, 3) - n * np.outer(fb, mb)        # sum Y_r P_c (centred)
    S = Sfm.T                                                   # S[a][b] = sum P_a Y_b   (:43-53)
    (Sxx, Sxy, Sxz), (Syx, Syy, Syz), (Szx, Szy, Szz) = S
    N = [[Sxx + Syy + Szz, Syz - Szy, -Sxz + Szx, Sxy - Syx],
         [-Szy + Syz, Sxx - Szz - Syy, Sxy + Syx, Sxz + Szx],
         [Szx - Sxz, Syx + Sxy, Syy - Szz - Sxx, Syz + Szy],
         [-Syx + Sxy, Szx + Sxz, Szy + Syz, Szz - Syy - Sxx]]
    w, V = np.linalg.eig(N)
    V = V[:, w.argsort()[::-1]]
    q0, q1, q2, q3 = V[0]
    Qbar = [[q0, -q1, -q2, -q3], [q1, q0, q3, -q2], [q2, -q3, q0, q1], [q3, q2, -q1, q0]]
    Q = [[q0, -q1, -q2, -q3], [q1, q0, -q3, q2], [q2, q3, q0, -q1], [q3, -q2, q1, q0]]
    R = np.matmul(np.transpose(Qbar), Q)[1:, 1:]
    D = s[22] - n * fb.dot(fb)                                  # sum |Y'|^2   (:89-91)
    Sp = (s[7] + s[10] + s[12]) - n * mb.dot(mb)                # sum |P'|^2
    sc = np.sqrt(D / Sp)
    t = (fb + o[3:6]) - sc * R.dot(mb + o[0:3])
    A = np.zeros((4, 4))
    A[:3, :3] = sc * R
    A[:3, 3] = t
    A[3, 3] = 1
    return A


def get_similar_transform(moving, fixed):
    """find_transform.py:21-99.  Moments are accumulated on the device; the 4 x 4 quaternion
    eigen-problem is solved on the host (see similar_from_sums)."""
    torch = nat.torch_mod()
    m, f = nat.to_dev(moving), nat.to_dev(fixed)
    if m.dim() != 2 or f.dim() != 2 or m.shape[1] != f.shape[1] or m.shape[0] < 3 or f.shape[0] < 3:
        raise ValueError("moving and fixed must be 3 x N")
    m, f = m[:3, :].contiguous(), f[:3, :].contiguous()
    origin = torch.cat([m[:, 0], f[:, 0]]).contiguous()
    sums = K.icp_accumulate(m, f, None, origin)
    A = similar_from_sums(sums.cpu().numpy(), origin.cpu().numpy())
    return torch.as_tensor(A, device=m.device) if nat.is_torch(moving) else A
