"""perform_icp (platymatch/estimate_transform/perform_icp.py)."""
import numpy as np

from .. import _kernels as K
from .. import _native as nat
from .find_transform import similar_from_sums

VERBOSE = True   # the reference prints one residual line per iteration (perform_icp.py:24)


def perform_icp(moving, fixed, icp_iterations=50, transform='Affine', log=None):
    """perform_icp.py:7-26 -> A_icp (4 x 4).

    'Affine': the whole loop (nearest neighbours, refit, apply, compose) is enqueued on the
    device in one call with no host synchronisation until the result is read.
    'Similar': nearest neighbours, moment sums, application and composition run on the device;
    the 4 x 4 quaternion eigen-problem per iteration is solved on the host.
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
        A, res, nn_all = K.icp(m, f, iters, want_nn=want_nn)
    elif transform == 'Similar':
        A = torch.eye(4, dtype=torch.float64, device=m.device)
        origin = torch.cat([f[:, 0], f[:, 0]]).contiguous()
        res_l, nn_l = [], []
        for _ in range(iters):
            nn, _ = K.icp_nn(m, f, want_dist=False)
            sums = K.icp_accumulate(m, f, nn, origin)
            A_est = torch.as_tensor(similar_from_sums(sums.cpu().numpy(), origin.cpu().numpy()), device=m.device)
            parts = K.icp_apply(A_est.reshape(16), m, f, nn, A.reshape(16))
            res_l.append(parts[0] / parts[1])
            nn_l.append(nn)
        res = torch.stack(res_l) if res_l else torch.empty(0, dtype=torch.float64, device=m.device)
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
