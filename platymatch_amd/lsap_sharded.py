"""The device-resident assignment solve (lsap.py) for a cost matrix whose ROWS are sharded over the ranks of a
torch.distributed group — the layout the sharded cost build leaves behind (pipeline.build_costs: rank g holds rows
[b_g, b_g+1) of every hypothesis, all M columns).

Nothing of the matrix travels.  One rank (the root of a hypothesis) runs the sparse-core solver; every query it makes of the
dense matrix — column minima, per-row candidates for the core / for pricing, the certificate — is answered by all ranks for
their own rows with the same HIP kernels as on one GPU, and only the answers cross the wire: k candidates per row, the
column duals (8 M bytes per pricing round), a few counters.  At 50 000 x 50 000 that is ~30 MB per hypothesis instead of
the 20 GB a gather of the matrix moves, and it is the only way to assign a matrix that does not fit one GPU (BASELINE
config 4: 200 000 x 200 000 is 320 GB; a rank's block of one hypothesis is 40 GB).

Protocol: the root broadcasts (operation, which matrix, arguments); every rank, the root included, runs it on its block;
the answers are gathered on the root.  Workers loop in `serve` until the root says stop.  Rows must be the short side
(N <= M): with N > M the solver works on the transpose, whose rows are this layout's columns."""
import numpy as np

from . import lsap


def _dist():
    import torch.distributed as dist
    return dist


def _answer(local, row0, rows, op, args):
    """One rank's share of a query on its block `local` (rows [row0, row0 + rows) of the matrix)."""
    if op == "row_select":
        v, k = args
        return local.row_select(v, k)
    if op == "diagonal":
        # entry (i, i) of the global matrix for the block's rows
        return local.block_diagonal(row0) if hasattr(local, "block_diagonal") else None
    if op == "col_min":
        return local.col_min()
    if op == "bid":                                      # the listed matrix rows that live in this block
        v, all_rows = args
        sel = np.flatnonzero((all_rows >= row0) & (all_rows < row0 + rows))
        j1, u1, u2 = local.bid(v, all_rows[sel] - row0)
        return sel, j1, u1, u2
    if op == "entries":
        all_rows, all_cols = args
        sel = np.flatnonzero((all_rows >= row0) & (all_rows < row0 + rows))
        return sel, local.entries(all_rows[sel] - row0, all_cols[sel])
    if op == "certificate":
        u, v, c4r, delta, eps, cap = args
        viol, loose, tight, red, bound = local.certificate(u[row0:row0 + rows], v, c4r[row0:row0 + rows], delta, eps, cap)
        if tight is not None and len(tight):
            tight = tight.copy()
            tight[:, 0] += row0                           # block rows -> matrix rows
        return viol, loose, tight, red, bound
    raise ValueError("unknown operation %r" % (op,))


class _Failed:
    """What a rank gathers instead of an answer when its share of a query raised (e.g. out of device memory while allocating
    the certificate's buffers): the two collectives of a query stay matched on every rank, and the root turns the message
    into the error solve_pair_sharded raises on ALL ranks."""

    def __init__(self, rank, exc):
        self.message = "rank %d: %s: %s" % (rank, type(exc).__name__, exc)


def _safe_answer(rank, local, row0, rows, op, args):
    try:
        return _answer(local, row0, rows, op, args)
    except Exception as e:                               # never leave the query between its broadcast and its gather
        return _Failed(rank, e)


class ShardedMatrix:
    """lsap.DeviceMatrix's interface over row blocks on several ranks — the object the ROOT hands to lsap.solve_core /
    lsap.certify.  `locals_` are this rank's blocks of the matrices that may be queried (e.g. a hypothesis and its twin)."""

    def __init__(self, locals_, which, bounds, group, root, n_cols):
        self.locals, self.which, self.bounds, self.group, self.root = locals_, which, bounds, group, root
        self.shape = (bounds[-1], n_cols)

    def _ask(self, op, args):
        dist = _dist()
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        dist.broadcast_object_list([(op, self.which, args)], src=_global(self.group, self.root), group=self.group)
        mine = _safe_answer(rank, self.locals[self.which], self.bounds[rank], self.bounds[rank + 1] - self.bounds[rank], op, args)
        parts = [None] * world
        dist.gather_object(mine, parts, dst=_global(self.group, self.root), group=self.group)
        failed = [p.message for p in parts if isinstance(p, _Failed)]
        if failed:                                        # raised with both collectives of the query completed on every rank
            raise RuntimeError("query %r failed on " % (op,) + "; ".join(failed))
        return parts

    def row_select(self, v, k):
        parts = self._ask("row_select", (v, k))
        return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), int(max(p[2] for p in parts)))

    def diagonal(self, n):
        return np.concatenate(self._ask("diagonal", None))[:n]

    def col_min(self):
        return np.minimum.reduce(self._ask("col_min", None))

    def bid(self, v, rows):
        rows = np.asarray(rows, dtype=np.int64)
        j1, u1, u2 = np.full(rows.size, -1, np.int32), np.full(rows.size, np.inf), np.full(rows.size, np.inf)
        for sel, a, b, c in self._ask("bid", (v, rows)):
            j1[sel], u1[sel], u2[sel] = a, b, c
        return j1, u1, u2

    def entries(self, rows, cols):
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        out = np.empty(rows.size)
        for sel, vals in self._ask("entries", (rows, cols)):
            out[sel] = vals
        return out

    def certificate(self, u, v, col4row, delta, eps, cap):
        parts = self._ask("certificate", (u, v, col4row, delta, eps, cap))
        viol, loose = sum(p[0] for p in parts), sum(p[1] for p in parts)
        bound = float(sum(p[4] for p in parts))
        if any(p[2] is None for p in parts) or sum(len(p[2]) for p in parts) > cap:
            return viol, loose, None, None, bound
        return viol, loose, np.concatenate([p[2] for p in parts]), np.concatenate([p[3] for p in parts]), bound

    def stop(self):
        _dist().broadcast_object_list([("stop", 0, None)], src=_global(self.group, self.root), group=self.group)


def _global(group, group_rank):
    dist = _dist()
    if group is None or group is dist.group.WORLD:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def serve(locals_, bounds, group, root):
    """A worker's side: answer the root's queries about this rank's blocks until it says stop."""
    dist = _dist()
    rank = dist.get_rank(group)
    row0, rows = bounds[rank], bounds[rank + 1] - bounds[rank]
    while True:
        box = [None]
        dist.broadcast_object_list(box, src=_global(group, root), group=group)
        op, which, args = box[0]
        if op == "stop":
            return
        dist.gather_object(_safe_answer(rank, locals_[which], row0, rows, op, args), None, dst=_global(group, root), group=group)


def solve_pair_sharded(local_h, local_twin, bounds, n_cols, group, root, info=None, accept_near_ties=False):
    """One hypothesis and (optionally) its twin, rows sharded: the root solves the first on its sparse core and certifies the
    result on both matrices; a twin that does not accept its sibling's duals is solved on its own core.  -> (col4row of the
    hypothesis or None, col4row of the twin or None) on EVERY rank; None = not certified (the caller takes another route
    for that matrix).  N <= M required.  accept_near_ties: an assignment certified optimal but not proven unique
    (lsap.certify: info["optimal"]) is returned instead of None; info["near_tie"] lists which (0 = hypothesis, 1 = twin)."""
    dist = _dist()
    rank = dist.get_rank(group)
    locals_ = [local_h] + ([local_twin] if local_twin is not None else [])
    out = [None, None, None]                              # col4row, twin's col4row, error message
    if rank == root:
        try:
            info = {} if info is None else info
            near = info["near_tie"] = []
            M = ShardedMatrix(locals_, 0, bounds, group, root, n_cols)
            sol = lsap.solve_core(M, info)
            certified = sol is not None and lsap.certify(M, *sol, info=info)
            if certified:
                out[0] = sol[2]
            elif accept_near_ties and sol is not None and info.get("optimal"):
                out[0] = sol[2]
                near.append(0)
            if local_twin is not None:
                Mt = ShardedMatrix(locals_, 1, bounds, group, root, n_cols)
                tinfo = info["twin"] = {}
                if certified and lsap.certify(Mt, *sol, info=tinfo):
                    out[1] = sol[2]
                    tinfo["route"] = "sibling's duals certified"
                else:                                     # the twin on its own
                    tinfo.clear()
                    sol_t = lsap.solve_core(Mt, tinfo)
                    if sol_t is not None and lsap.certify(Mt, *sol_t, info=tinfo):
                        out[1] = sol_t[2]
                    elif accept_near_ties and sol_t is not None and tinfo.get("optimal"):
                        out[1] = sol_t[2]
                        near.append(1)
                    tinfo["route"] = "own core"
        except Exception as e:                            # the workers are waiting for queries: release them, then raise everywhere
            out[2] = "%s: %s" % (type(e).__name__, e)
        finally:
            ShardedMatrix(locals_, 0, bounds, group, root, n_cols).stop()
    else:
        serve(locals_, bounds, group, root)
    box = [out]
    dist.broadcast_object_list(box, src=_global(group, root), group=group)
    if box[0][2] is not None:
        raise RuntimeError("sharded assignment failed on rank %d: %s" % (root, box[0][2]))
    return box[0][0], box[0][1]
