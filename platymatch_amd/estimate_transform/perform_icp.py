"""perform_icp (platymatch/estimate_transform/perform_icp.py)."""
import numpy as np

from .. import _kernels as K
from .. import _native as nat
from .find_transform import affine_pinv_host, similar_from_moments

VERBOSE = True   # the reference prints one residual line per iteration (perform_icp.py:24)
# Affine loop: one launch per iteration (pm_icp) by default.  ONE_LAUNCH = True runs iterations 1 .. n-1 in ONE launch of
# persistent workgroups instead (pm_icp_one_launch: points, matches and the 4 x 4 stay in registers / LDS, search tables warm
# in L2) — identical results, built and measured in round 3 and NOT faster on MI355X: a kernel boundary costs ~1.5 us, the
# grid-wide hand-off that replaces it (782 workgroups at 50 000 points: arrival counters + a generation word every workgroup
# polls) costs more, and every memory round trip of the reduction's tail takes about twice as long beside the resident,
# waiting workgroups (profiles/r03_icp_stamps_*.txt: 24.8 us per iteration against 21.6 at 50 000 points, 13.3 against 11.4 at
# 5 000).  Kept as an option for devices / drivers where the balance differs.
ONE_LAUNCH = False


def _icp_with_host_fits(m, f, iters, want_nn):
    """perform_icp.py:14-25 with get_affine_transform's pinv on the host: m (3 x N, GPU) is updated in place."""
    torch = nat.torch_mod()
    A_icp = torch.eye(4, dtype=torch.float64, device=m.device).reshape(16).contiguous()
    fh = f.cpu().numpy()
    grid = K.icp_grid(f) if iters else None
    res_l, nn_l = [], []
    for _ in range(iters):
        nn = K.icp_nn(m, f, want_dist=False, grid=grid)[0]                       # :15-16
        A_est = affine_pinv_host(m.cpu().numpy(), fh[:, nn.cpu().numpy()])      # :18
        parts = K.icp_apply(nat.to_dev(A_est, dev=m.device).reshape(16), m, f, nn, A_icp, nn_trusted=True)   # :23-25
        res_l.append(parts[0] / parts[1])
        nn_l.append(nn)
    res = torch.stack(res_l) if res_l else torch.empty(0, dtype=torch.float64, device=m.device)
    return A_icp.reshape(4, 4), res, (torch.stack(nn_l) if (want_nn and nn_l) else None)


def perform_icp(moving, fixed, icp_iterations=50, transform='Affine', log=None, one_launch=None):
    """perform_icp.py:7-26 -> A_icp (4 x 4).

    'Affine': the whole loop (nearest neighbours, refit, apply, compose) is enqueued on the
    device in one call with no host synchronisation until the result is read; if the moving cloud turns out
    (nearly) planar — where the reference's pinv gives a minimum-norm fit the normal equations cannot — the loop is
    rerun with pinv fits on the host (_icp_with_host_fits).
    'Similar': everything O(N) of an iteration runs on the device in NumPy's own arithmetic — nearest neighbours, the two
    centroids, the nine product sums, D and Sp (K.similar_moments), the application of the fit and the residual
    (K.similar_apply); the host receives seventeen numbers per iteration and runs the reference's own NumPy lines on them for
    the 4 x 4 eigen-decomposition and what follows (find_transform.similar_from_moments; similar_transform_host explains why
    that part can only be reproduced by NumPy/LAPACK itself).
    `log`, if a dict, receives 'nn' [iters, N] int32, 'residuals' [iters] and 'moved' [3, N]."""
    torch = nat.torch_mod()
    m, f = nat.to_dev(moving), nat.to_dev(fixed)
    if m.dim() != 2 or f.dim() != 2 or m.shape[0] not in (3, 4) or f.shape[0] not in (3, 4):
        raise ValueError("moving and fixed must be 3 x N (or 4 x N)")
    m = m[:3, :].contiguous().clone()          # updated in place below; the caller's array is never mutated
    f = f[:3, :].contiguous()
    iters = int(icp_iterations)
    want_nn = log is not None
    if transform == 'Affine':
        start = m.clone()
        status = torch.zeros(1, dtype=torch.int32, device=m.device)
        A, res, nn_all = K.icp(m, f, iters, want_nn=want_nn, status=status, one_launch=ONE_LAUNCH if one_launch is None else bool(one_launch))
        if int(status.item()) == 2:
            # the persistent launch gave up waiting for its own workgroups: other work held part of the device for seconds
            # (include/platymatch_hip.h: pm_icp_one_launch's contract).  Same loop, one launch per iteration.
            m.copy_(start)
            status.zero_()
            A, res, nn_all = K.icp(m, f, iters, want_nn=want_nn, status=status, one_launch=False)
        if int(status.item()) != 0 or not bool(torch.isfinite(A).all()):
            # a (nearly) planar moving cloud: the device's normal equations are singular where the reference's pinv
            # (find_transform.py:17) returns the minimum-norm fit.  Rerun with the search and the application on the
            # device and the reference's own expression for the 4 x 4 on the host, one small round trip per iteration.
            m = start
            A, res, nn_all = _icp_with_host_fits(m, f, iters, want_nn)
    elif transform == 'Similar':
        # Per iteration: nearest neighbours (device); the seventeen O(N) numbers of get_similar_transform in NumPy's own
        # arithmetic (device: K.similar_moments -> 136 bytes to the host); the 4 x 4 eigen-decomposition and what follows it
        # with the reference's own NumPy calls (host: similar_from_moments); application and residual as NumPy rounds them
        # (device: K.similar_apply).  No cloud crosses PCIe inside the loop.
        # np.mean adds a cloud up in the order its memory layout dictates: the caller's moving array decides for the first
        # iteration (afterwards it is np.matmul's C-ordered result), the matches fixed[:, nn] are always Fortran-ordered.
        first_sequential = False
        if not nat.is_torch(moving):
            mh0 = np.asarray(moving)
            first_sequential = bool(mh0.flags["F_CONTIGUOUS"] and not mh0.flags["C_CONTIGUOUS"])
        A_h = np.identity(4)
        grid = K.icp_grid(f) if iters else None
        ws = None
        if iters:
            lib = nat.load()
            ws = nat.workspace(lib.pm_similar_workspace(m.shape[1]), m.device)
        res_l, nn_l = [], []
        for it in range(iters):
            nn = K.icp_nn(m, f, want_dist=False, grid=grid)[0]               # perform_icp.py:15-16
            v = K.similar_moments(m, f, nn, mov_sequential=(it == 0 and first_sequential), fix_sequential=True, ws=ws)
            A_est = similar_from_moments(v.cpu().numpy())                     # :20 (the fit)
            res_l.append(K.similar_apply(nat.to_dev(A_est, dev=m.device).reshape(16), m, f, nn, ws=ws))   # :23, :24 (get_error)
            A_h = np.matmul(A_est, A_h)                                       # :25
            nn_l.append(nn)
        A = torch.as_tensor(A_h, device=f.device)
        res = torch.cat(res_l) if res_l else torch.empty(0, dtype=torch.float64, device=f.device)
        nn_all = torch.stack(nn_l) if (want_nn and nn_l) else None
    else:
        raise ValueError("transform must be 'Affine' or 'Similar'")
    if VERBOSE and iters:
        for i, r in enumerate(res.cpu().tolist()):
            print("Residual at iteration {} is {}".format(str(i), r))
    if log is not None:
        log['nn'] = nn_all.cpu().numpy() if nn_all is not None else np.zeros((0, m.shape[1]), np.int32)
        log['residuals'] = res.cpu().numpy()
        log['moved'] = nat.like_input(m, moving)
    return nat.like_input(A, moving)
