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


def affine_pinv_host(moving, fixed):
    """find_transform.py:11-17 on host arrays, the reference's own expression: [fixed; 1] . pinv([moving; 1]).
    The device path solves centred normal equations, which equal this for full-rank input; for rank-deficient input
    (fewer than four points, coplanar or repeated points) pinv returns the minimum-norm least-squares answer — whose last
    row is not 0 0 0 1 — and only this expression reproduces it."""
    moving, fixed = np.asarray(moving, dtype=np.float64), np.asarray(fixed, dtype=np.float64)
    one = np.ones((1, moving.shape[1]))
    return np.matmul(np.vstack((fixed[:3], one)), np.linalg.pinv(np.vstack((moving[:3], one))))


def affine_pinv_host_batch(P, Y):
    """The same for T small samples at once: P, Y [T, 3, k] -> [T, 4, 4] (np.linalg.pinv works on stacks)."""
    P, Y = np.asarray(P, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    one = np.ones((P.shape[0], 1, P.shape[2]))
    return np.matmul(np.concatenate((Y, one), axis=1), np.linalg.pinv(np.concatenate((P, one), axis=1)))


def get_affine_transform(moving, fixed, with_ones=False):
    """find_transform.py:4-17: least-squares 4 x 4 with [fixed;1] = A [moving;1].

    The reference forms fixed . pinv(moving).  For a full-rank cloud (>= 4 points, not coplanar) that is the
    least-squares solution, computed here on the device from centred moments (last row exactly 0 0 0 1, the reference's
    is 0 0 0 1 to ~1e-17).  For rank-deficient input the kernel reports it (pm_fit_affine's status word) and the
    reference's own expression runs on the host (affine_pinv_host): same minimum-norm answer as the reference."""
    torch = nat.torch_mod()
    m, f = _pair(moving, fixed, with_ones)
    degenerate = m.shape[1] < 4
    if not degenerate:
        status = torch.zeros(1, dtype=torch.int32, device=m.device)
        A = K.fit_affine(m, f, status=status)
        degenerate = int(status.item()) != 0 or not bool(torch.isfinite(A).all())
    if degenerate:
        A = torch.as_tensor(affine_pinv_host(m.cpu().numpy(), f.cpu().numpy()), device=m.device)
    return nat.like_input(A, moving)


def _quaternion_matrix(Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz):
    """The symmetric 4 x 4 matrix N of Horn's method exactly as the reference writes it (find_transform.py:55-58), from scalars
    or from arrays of T fits at once (rows as tuples: the caller stacks them)."""
    return [[Sxx + Syy + Szz, Syz - Szy, -Sxz + Szx, Sxy - Syx],
            [-Szy + Syz, Sxx - Szz - Syy, Sxy + Syx, Sxz + Szx],
            [Szx - Sxz, Syx + Sxy, Syy - Szz - Sxx, Syz + Szy],
            [-Syx + Sxy, Szx + Sxz, Szy + Syz, Szz - Syy - Sxx]]


def _rotation_from_one_N(N):
    """One 4 x 4 quaternion matrix -> R the reference's way (find_transform.py:60-84): np.linalg.eig, eigenvectors sorted by
    decreasing eigenvalue, q = ROW 0 of that matrix (the reference's quirk), R = (Qbar^T Q)[1:, 1:] by np.matmul."""
    w, V = np.linalg.eig(N)
    V = V[:, w.argsort()[::-1]]
    q0, q1, q2, q3 = V[0]                                      # row 0, as the reference has it
    Qbar = [[q0, -q1, -q2, -q3], [q1, q0, q3, -q2], [q2, -q3, q0, q1], [q3, q2, -q1, q0]]
    Q = [[q0, -q1, -q2, -q3], [q1, q0, -q3, q2], [q2, q3, q0, -q1], [q3, -q2, q1, q0]]
    return np.matmul(np.transpose(Qbar), Q)[1:, 1:]


def _similar_4x4(R, sc, cs, ct):
    """[s R | ct - s R cs; 0 0 0 1] (find_transform.py:94-99); cs, ct 3 x 1."""
    A = np.zeros((4, 4))
    A[:3, :3] = sc * R
    A[:3, 3:4] = ct - sc * np.matmul(R, cs)
    A[3, 3] = 1
    return A


def _rotation_from_N(N):
    """[T, 4, 4] quaternion matrices -> [T, 3, 3] rotations the reference's way (:60-84): eigenvectors sorted by
    decreasing eigenvalue, q = ROW 0 of that matrix, R = (Qbar^T Q)[1:, 1:]."""
    T = N.shape[0]
    w, V = np.linalg.eig(N)
    if np.iscomplexobj(w):      # a stack turns complex as a whole if one member does: redo each on its own, as the reference would
        pairs = [np.linalg.eig(N[t]) for t in range(T)]
        q = np.stack([Vt[:, wt.argsort()[::-1]][0] for wt, Vt in pairs])
    else:
        order = np.argsort(w, axis=1)[:, ::-1]
        q = np.take_along_axis(V, order[:, None, :], axis=2)[:, 0, :]           # row 0 of the sorted eigenvector matrix
    q0, q1, q2, q3 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Qbar = np.stack([np.stack([q0, -q1, -q2, -q3], 1), np.stack([q1, q0, q3, -q2], 1), np.stack([q2, -q3, q0, q1], 1),
                     np.stack([q3, q2, -q1, q0], 1)], 1)
    Q = np.stack([np.stack([q0, -q1, -q2, -q3], 1), np.stack([q1, q0, -q3, q2], 1), np.stack([q2, q3, q0, -q1], 1),
                  np.stack([q3, -q2, q1, q0], 1)], 1)
    return np.einsum('tki,tkj->tij', Qbar, Q)[:, 1:, 1:]       # (Qbar^T Q)[1:, 1:], summed over k in order


def _numpy_sum_rows(x):
    """Row sums of x [T, n] in the order np.sum uses for a contiguous 1-D float64 array of n elements (NumPy's pairwise
    summation: plain loop below 8 elements, eight running sums per block of up to 128, halves above that) —
    restated so that T sums are taken at once without NumPy choosing another loop order for the 2-D array."""
    n = x.shape[1]
    if n < 8:
        res = np.zeros(x.shape[0])
        for i in range(n):
            res = res + x[:, i]
        return res
    if n <= 128:
        r = [x[:, j].copy() for j in range(8)]
        top = n - (n % 8)
        for i in range(8, top, 8):
            for j in range(8):
                r[j] = r[j] + x[:, i + j]
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(top, n):
            res = res + x[:, i]
        return res
    half = n // 2
    half -= half % 8
    return _numpy_sum_rows(x[:, :half]) + _numpy_sum_rows(x[:, half:])


def similar_fit_batch(P, Y):
    """get_similar_transform (find_transform.py:21-99) for T small sets of pairs at once: P, Y are [T, 3, k] -> [T, 4, 4].
    Host NumPy, for the RANSAC fits of transform='Similar' (k = min_samples).

    The reference takes ROW 0 of np.linalg.eig's eigenvector matrix (:60-66), so its result hangs on LAPACK's eigenvector
    signs, and those flip when the 4 x 4 matrix N changes in the last bit.  N is therefore built with the reference's
    operations in the reference's order, as they run on what do_ransac passes in (shape_context.py:123-124: 3 x k
    arrays made by fancy indexing, i.e. Fortran-ordered): np.mean over such an array adds the k columns one after the
    other; np.sum over each contiguous product vector is NumPy's pairwise sum; the sixteen entries are the reference's
    expressions.  N is then bit-identical to the reference's and np.linalg.eig on the stack runs the same LAPACK call
    per matrix.  What follows the eigenvectors (R, s, t) is smooth in its inputs and agrees to rounding."""
    P = np.asarray(P, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    T, k = P.shape[0], P.shape[2]

    def mean_columns(x):                                       # [T, 3, k] -> [T, 3, 1]
        acc = x[:, :, 0].copy()
        for j in range(1, k):
            acc = acc + x[:, :, j]
        return (acc / k)[:, :, None]

    ct, cs = mean_columns(Y), mean_columns(P)                  # :27-28
    Yp, Pp = Y - ct, P - cs                                    # :31-32
    Px, Py, Pz = Pp[:, 0, :], Pp[:, 1, :], Pp[:, 2, :]
    Yx, Yy, Yz = Yp[:, 0, :], Yp[:, 1, :], Yp[:, 2, :]
    S = _numpy_sum_rows
    Sxx, Sxy, Sxz = S(Yx * Px), S(Px * Yy), S(Px * Yz)         # :43-53
    Syx, Syy, Syz = S(Py * Yx), S(Py * Yy), S(Py * Yz)
    Szx, Szy, Szz = S(Pz * Yx), S(Pz * Yy), S(Pz * Yz)
    N = np.empty((T, 4, 4))
    for row, entries in enumerate(_quaternion_matrix(Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz)):    # :55-58
        N[:, row] = np.stack(entries, axis=1)
    R = _rotation_from_N(N)
    D = np.zeros(T)
    Sp = np.zeros(T)
    for j in range(k):                                         # :86-91, one point at a time
        D = D + ((Yx[:, j] * Yx[:, j] + Yy[:, j] * Yy[:, j]) + Yz[:, j] * Yz[:, j])
        Sp = Sp + ((Px[:, j] * Px[:, j] + Py[:, j] * Py[:, j]) + Pz[:, j] * Pz[:, j])
    sc = np.sqrt(D / Sp)
    c = cs[:, :, 0]
    Rc = (R[:, :, 0] * c[:, None, 0] + R[:, :, 1] * c[:, None, 1]) + R[:, :, 2] * c[:, None, 2]
    A = np.zeros((T, 4, 4))
    A[:, :3, :3] = sc[:, None, None] * R
    A[:, :3, 3] = ct[:, :, 0] - sc[:, None] * Rc               # :94
    A[:, 3, 3] = 1
    return A


def similar_transform_host(moving, fixed):
    """get_similar_transform (find_transform.py:21-99) on host arrays, operation for operation in NumPy.

    Why on the host, and why literally: the reference takes ROW 0 of np.linalg.eig's eigenvector matrix (:60-66) as
    its quaternion, so its answer hangs on LAPACK's eigenvector signs; a one-ulp change of the input flips them in
    about one fit out of ten (measured: DESIGN.md §2), and in the ICP loop every iteration feeds the next.  Only the
    reference's own sequence of NumPy/BLAS/LAPACK calls on arrays of the same memory layout reproduces its result, so
    that is what runs here (SURVEY.md §8a row 14 prescribes exactly this); the O(N M) work of the mode — descriptors,
    costs, nearest neighbours, RANSAC scoring — stays on the device."""
    moving, fixed = np.asarray(moving, dtype=np.float64), np.asarray(fixed, dtype=np.float64)
    ct = np.mean(fixed, 1, keepdims=True)
    cs = np.mean(moving, 1, keepdims=True)
    Y = fixed[:3, :] - ct[:3, :]
    P = moving[:3, :] - cs[:3, :]
    Px, Py, Pz = P[0, :], P[1, :], P[2, :]
    Yx, Yy, Yz = Y[0, :], Y[1, :], Y[2, :]
    Sxx, Sxy, Sxz = np.sum(Yx * Px), np.sum(Px * Yy), np.sum(Px * Yz)
    Syx, Syy, Syz = np.sum(Py * Yx), np.sum(Py * Yy), np.sum(Py * Yz)
    Szx, Szy, Szz = np.sum(Pz * Yx), np.sum(Pz * Yy), np.sum(Pz * Yz)
    R = _rotation_from_one_N(_quaternion_matrix(Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz))
    D = Sp = 0
    for i in range(Y.shape[1]):                                # :86-91
        D += np.matmul(np.transpose(Y[:, i]), Y[:, i])
        Sp += np.matmul(np.transpose(P[:, i]), P[:, i])
    return _similar_4x4(R, np.sqrt(D / Sp), cs[:3, :], ct[:3, :])


def similar_from_moments(v):
    """find_transform.py:55-99 from the seventeen numbers the device computed in NumPy's arithmetic (K.similar_moments:
    com_source, com_target, the nine sums, D, Sp): the 4 x 4 quaternion matrix, np.linalg.eig, row 0 of the sorted
    eigenvector matrix, R, s, t — the reference's own NumPy calls on bit-identical input, hence its own result.
    Only these lines of get_similar_transform run on the host (the eigenvector-row quirk forces LAPACK: DESIGN.md §2)."""
    v = np.asarray(v, dtype=np.float64)
    cs, ct = v[0:3].reshape(3, 1), v[3:6].reshape(3, 1)
    R = _rotation_from_one_N(_quaternion_matrix(*(v[k] for k in range(6, 15))))
    return _similar_4x4(R, np.sqrt(v[15] / v[16]), cs, ct)


def quaternion_matrix_from_moments(v):
    """The 4 x 4 matrix N of find_transform.py:55-58 alone (tests compare it with the oracle's bit for bit)."""
    v = np.asarray(v, dtype=np.float64)
    return np.array(_quaternion_matrix(*(v[k] for k in range(6, 15))))


def apply_affine_host(moving, A):
    """apply_transform.py:3-17 on host arrays with the reference's own calls (vstack + np.matmul): the Similar-mode
    chain needs the moved cloud to the bit (see similar_transform_host)."""
    moving = np.asarray(moving)
    if moving.shape[0] == 4:
        moving = moving[:3, :]
    hom = np.vstack((moving, np.ones((1, moving.shape[1]))))
    return np.matmul(A, hom)[:3, :]


DEVICE_MOMENTS_FROM = 4096        # clouds from this size on have their O(N) sums taken on the device (NumPy inputs; GPU tensors always)


def _sequential(x):
    """Does np.mean(x, 1) add this 2-D array's columns one after the other (Fortran order) rather than pairwise along its rows?"""
    return bool(x.flags["F_CONTIGUOUS"] and not x.flags["C_CONTIGUOUS"])


def get_similar_transform(moving, fixed):
    """find_transform.py:21-99 -> 4 x 4.  The O(N) arithmetic — centroids, the nine product sums, D, Sp — runs on the device in
    NumPy's own order for GPU tensors and for clouds of DEVICE_MOMENTS_FROM points or more (K.similar_moments; the memory order
    of a NumPy input decides how its centroid is summed, as it does in np.mean); the 4 x 4 eigen-decomposition and what follows
    are the reference's own NumPy calls on the host (similar_from_moments).  Smaller NumPy inputs take the literal host sequence
    (similar_transform_host).  Same bits either way (tests/test_gpu_similar.py)."""
    torch = nat.torch_mod()
    if nat.is_torch(moving) or nat.is_torch(fixed):
        m, f = nat.to_dev(moving), nat.to_dev(fixed)
        if m.dim() != 2 or f.dim() != 2 or m.shape[1] != f.shape[1] or m.shape[0] < 3 or f.shape[0] < 3:
            raise ValueError("moving and fixed must be 3 x N")
        # a torch tensor's rows are what get summed: row-major (C order) unless it is a transposed view
        seq_m = bool(m.stride(0) == 1 and m.shape[1] > 1 and m.stride(1) != 1)
        seq_f = bool(f.stride(0) == 1 and f.shape[1] > 1 and f.stride(1) != 1)
        v = K.similar_moments(m[:3].contiguous(), f[:3].contiguous(), None, mov_sequential=seq_m, fix_sequential=seq_f)
        return torch.as_tensor(similar_from_moments(v.cpu().numpy()), device=m.device)
    m, f = np.asarray(moving), np.asarray(fixed)
    if m.ndim != 2 or f.ndim != 2 or m.shape[1] != f.shape[1] or m.shape[0] < 3 or f.shape[0] < 3:
        raise ValueError("moving and fixed must be 3 x N")
    if m.shape[1] >= DEVICE_MOMENTS_FROM and m.dtype == np.float64 and f.dtype == np.float64 and m.shape[0] == 3 and f.shape[0] == 3:
        md, fd = nat.to_dev(np.ascontiguousarray(m)), nat.to_dev(np.ascontiguousarray(f))
        v = K.similar_moments(md, fd, None, mov_sequential=_sequential(m), fix_sequential=_sequential(f))
        return similar_from_moments(v.cpu().numpy())
    return similar_transform_host(m, f)
