"""The device-resident assignment solve (lsap.py) for a cost matrix whose ROWS are sharded over the ranks of a
torch.distributed group — the layout the sharded cost build leaves behind (pipeline.build_costs: rank g holds rows
[b_g, b_g+1) of every hypothesis, all M columns).

Nothing of the matrix travels.  One rank (the root of a hypothesis) runs the sparse-core solver; every query it makes of the
dense matrix — column minima, per-row candidates for the core / for pricing, the certificate — is answered by all ranks for
their own rows with the same HIP kernels as on one GPU, and only the answers cross the wire: k candidates per row, the
column duals (8 M bytes per pricing round), a few counters.  At 50 000 x 50 000 that is ~30 MB per hypothesis instead of
the 20 GB a gather of the matrix moves, and it is the only way to assign a matrix that does not fit one GPU (BASELINE
config 4: 200 000 x 200 000 is 320 GB; a rank's block of one hypothesis is 40 GB).

Protocol (round 4: TENSOR collectives only — no pickled objects; every buffer has a shape all ranks can derive):
  1. the root broadcasts a header, int64 [8] = (operation, which matrix, k or cap, listed rows, has v, 0, 0, 0);
  2. the operation's arguments follow as fixed-shape broadcasts (float64 v [M]; int64 row / column lists; for the certificate
     one float64 vector u | v | delta | eps and int32 col4row);
  3. every rank, the root included, answers for its own block; the answers are gathered on the root (torch.distributed.gather
     into ONE allocation, blocks padded to the largest): first a status record, int32 [2 + 64] = (failed, flag, message bytes),
     then the operation's payload (candidates int32 / float64 [rows, k]; for the certificate four float64 counters, then —
     the root having broadcast the largest list length — the near-tight entries).
A rank whose share raises still takes part in every collective of the query (zeros as payload) and the root raises its message
on all ranks afterwards.  Workers loop in `serve` until the root sends the stop header.  With the "nccl" backend (RCCL) the
buffers live on the GPU and a rank's candidates go from the kernel's output straight into the gather (DeviceMatrix.row_select_t);
with "gloo" they are host tensors.  Rows must be the short side (N <= M): with N > M the solver works on the transpose, whose
rows are this layout's columns.
Round 5: the same protocol serves the DEFAULT cost mode's float32 filter blocks (solve_pair_sharded_filtered: a FilteredMatrix on
the root, exact entries evaluated there), and several pairings at once with their roots' host solvers side by side
(solve_pairs_sharded_filtered: one all-reduce of pending headers per round, then the queries in pairing order)."""
import numpy as np

from . import lsap

OP_STOP, OP_ROW_SELECT, OP_DIAGONAL, OP_COL_MIN, OP_BID, OP_ENTRIES, OP_CERTIFICATE = range(7)
OP_NAMES = ("stop", "row_select", "diagonal", "col_min", "bid", "entries", "certificate")
_MSG_WORDS = 64          # 256 bytes of error message per rank


def _dist():
    import torch.distributed as dist
    return dist


def _torch():
    import torch
    return torch


def _global(group, group_rank):
    dist = _dist()
    if group is None or group is dist.group.WORLD:
        return group_rank
    return dist.get_global_rank(group, group_rank)


class _Wire:
    """The collectives of one solve: where the buffers live (host for gloo, the current GPU for RCCL) and the three moves."""

    def __init__(self, bounds, n_cols, group, root):
        dist, torch = _dist(), _torch()
        self.group, self.root = group, root
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.src = _global(group, root)
        self.bounds, self.nc, self.nr = bounds, int(n_cols), int(bounds[-1])
        self.row0, self.rows = bounds[self.rank], bounds[self.rank + 1] - bounds[self.rank]
        self.biggest = max(bounds[g + 1] - bounds[g] for g in range(self.world))
        self.on_host = dist.get_backend(group) == "gloo"
        self.dev = torch.device("cpu") if self.on_host else torch.device("cuda", torch.cuda.current_device())
        self.is_root = self.rank == root

    def tensor(self, a, dtype):
        torch = _torch()
        if lsap.nat.is_torch(a):
            return a.to(device=self.dev, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.dev, dtype=dtype)

    def empty(self, shape, dtype):
        return _torch().empty(shape, dtype=dtype, device=self.dev)

    def zeros(self, shape, dtype):
        return _torch().zeros(shape, dtype=dtype, device=self.dev)

    def bcast(self, t):
        _dist().broadcast(t, src=self.src, group=self.group)
        return t

    def gather(self, t):
        """-> [world, *t.shape] on the root (one allocation), None elsewhere."""
        whole = self.empty((self.world,) + tuple(t.shape), t.dtype) if self.is_root else None
        _dist().gather(t.contiguous(), [whole[g] for g in range(self.world)] if self.is_root else None, dst=self.src, group=self.group)
        return whole


def _host(t):
    return t.cpu().numpy()


def _status(wire, failed, flag, message):
    st = np.zeros(2 + _MSG_WORDS, dtype=np.int32)
    st[0], st[1] = int(failed), int(flag)
    if message:
        raw = message.encode("utf-8", "replace")[:4 * _MSG_WORDS]
        st[2:].view(np.uint8)[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    return wire.tensor(st, _torch().int32)


def _message(words):
    raw = np.ascontiguousarray(words, dtype=np.int32).view(np.uint8).tobytes()
    return raw.split(b"\0", 1)[0].decode("utf-8", "replace")


def _header_of(request):
    """(op, which, arguments) -> the five integers every rank needs to shape the query's buffers."""
    op, which, args = request
    return (int(op), int(which), int(args.get("k", args.get("cap", 0))), int(len(args["rows"])) if "rows" in args else 0,
            1 if args.get("v") is not None else 0)


def _query(wire, locals_, request=None, header=None):
    """One query, executed by EVERY rank: the root passes request = (op, which, dict of arguments), the workers None.
    header: the query's five integers when every rank knows them already (the multiplexed driver below exchanges the headers of
    all pairings in one all-reduce); None: the root broadcasts them first.
    -> ("stop", None) / (op, gathered answer on the root | None elsewhere).  All collectives of the query complete on every rank
    before the root raises a rank's failure."""
    torch = _torch()
    i64, i32, f64 = torch.int64, torch.int32, torch.float64
    args = {}
    if wire.is_root:
        args = request[2]
    if header is None:
        if wire.is_root:
            h = np.zeros(8, dtype=np.int64)
            h[:5] = _header_of(request)
            hdr = wire.tensor(h, i64)
        else:
            hdr = wire.empty(8, i64)
        wire.bcast(hdr)
        header = tuple(int(x) for x in _host(hdr)[:5])
    op, which, a, n_listed, has_v = header
    if op == OP_STOP:
        return "stop", None
    nr, nc, rows, row0, big = wire.nr, wire.nc, wire.rows, wire.row0, wire.biggest
    # ---- arguments
    v_t = rows_t = cols_t = uv_t = c4r_t = None
    if op in (OP_ROW_SELECT, OP_BID) and has_v:
        v_t = wire.bcast(wire.tensor(args["v"], f64) if wire.is_root else wire.empty(nc, f64))
    if op in (OP_BID, OP_ENTRIES):
        rows_t = wire.bcast(wire.tensor(args["rows"], i64) if wire.is_root else wire.empty(n_listed, i64))
    if op == OP_ENTRIES:
        cols_t = wire.bcast(wire.tensor(args["cols"], i64) if wire.is_root else wire.empty(n_listed, i64))
    if op == OP_CERTIFICATE:
        if wire.is_root:
            uv = np.concatenate([np.asarray(args["u"], dtype=np.float64), np.asarray(args["v"], dtype=np.float64),
                                 [float(args["delta"]), float(args["eps"])]])
        uv_t = wire.bcast(wire.tensor(uv, f64) if wire.is_root else wire.empty(nr + nc + 2, f64))
        c4r_t = wire.bcast(wire.tensor(args["col4row"], i32) if wire.is_root else wire.empty(nr, i32))
    # ---- this rank's share (never leaves the query between its broadcasts and its gathers)
    local = locals_[which]
    failed, flag, message, ans = 0, 0, "", None
    try:
        if op == OP_ROW_SELECT:
            if hasattr(local, "row_select_t") and not wire.on_host:
                ans = local.row_select_t(v_t, a)                                   # device tensors, straight into the gather
                flag = int(ans[2].item())
            else:
                ans = local.row_select(None if v_t is None else _host(v_t), a)
                flag = int(ans[2])
        elif op == OP_DIAGONAL:
            ans = local.block_diagonal(row0)
        elif op == OP_COL_MIN:
            ans = local.col_min()
        elif op == OP_BID:
            all_rows = _host(rows_t)
            sel = np.flatnonzero((all_rows >= row0) & (all_rows < row0 + rows))
            ans = (sel,) + tuple(local.bid(_host(v_t), all_rows[sel] - row0))
        elif op == OP_ENTRIES:
            all_rows, all_cols = _host(rows_t), _host(cols_t)
            sel = np.flatnonzero((all_rows >= row0) & (all_rows < row0 + rows))
            ans = (sel, local.entries(all_rows[sel] - row0, all_cols[sel]))
        elif op == OP_CERTIFICATE:
            uv = _host(uv_t)
            c4r = _host(c4r_t)
            ans = local.certificate(uv[row0:row0 + rows], uv[nr:nr + nc], c4r[row0:row0 + rows], float(uv[nr + nc]), float(uv[nr + nc + 1]), a)
        else:
            raise ValueError("unknown operation %d" % op)
    except Exception as e:
        failed, message, ans = 1, "rank %d: %s: %s" % (wire.rank, type(e).__name__, e), None
    status = wire.gather(_status(wire, failed, flag, message))
    # ---- answers (fixed shapes; a failed rank sends zeros)
    out = None
    if op == OP_ROW_SELECT:
        c_t, x_t = wire.zeros((big, a), i32), wire.zeros((big, a), f64)
        if ans is not None:
            c_t[:rows] = wire.tensor(ans[0], i32)
            x_t[:rows] = wire.tensor(ans[1], f64)
        C, X = wire.gather(c_t), wire.gather(x_t)
        if wire.is_root:
            b = wire.bounds
            out = (np.concatenate([_host(C[g, :b[g + 1] - b[g]]) for g in range(wire.world)]),
                   np.concatenate([_host(X[g, :b[g + 1] - b[g]]) for g in range(wire.world)]),
                   int(_host(status[:, 1]).max()))
    elif op == OP_DIAGONAL:
        d_t = wire.zeros(big, f64)
        if ans is not None:
            d_t[:len(ans)] = wire.tensor(ans, f64)
        D = wire.gather(d_t)
        if wire.is_root:
            b = wire.bounds
            out = np.concatenate([_host(D[g, :max(0, min(b[g + 1] - b[g], nc - b[g]))]) for g in range(wire.world)])
    elif op == OP_COL_MIN:
        V = wire.gather(wire.tensor(ans, f64) if ans is not None else wire.zeros(nc, f64))
        if wire.is_root:
            out = np.minimum.reduce(_host(V))
    elif op == OP_BID:
        j_t, u_t = wire.zeros(n_listed, i32), wire.zeros((2, n_listed), f64)
        if ans is not None and len(ans[0]):
            sel = wire.tensor(ans[0], i64)
            j_t[sel] = wire.tensor(ans[1], i32)
            u_t[0, sel], u_t[1, sel] = wire.tensor(ans[2], f64), wire.tensor(ans[3], f64)
        J, Uu = wire.gather(j_t), wire.gather(u_t)
        if wire.is_root:
            owner = np.searchsorted(np.asarray(wire.bounds), _host(rows_t), side="right") - 1
            pick = np.arange(n_listed)
            out = (_host(J)[owner, pick], _host(Uu)[owner, 0, pick], _host(Uu)[owner, 1, pick])
    elif op == OP_ENTRIES:
        e_t = wire.zeros(n_listed, f64)
        if ans is not None and len(ans[0]):
            e_t[wire.tensor(ans[0], i64)] = wire.tensor(ans[1], f64)
        E = wire.gather(e_t)
        if wire.is_root:
            owner = np.searchsorted(np.asarray(wire.bounds), _host(rows_t), side="right") - 1
            out = _host(E)[owner, np.arange(n_listed)]
    elif op == OP_CERTIFICATE:
        cnt = np.zeros(4)
        if ans is not None:
            viol, loose, tight, red, bound = ans
            cnt[:] = (viol, loose, -1.0 if tight is None else len(tight), bound)
        CNT = wire.gather(wire.tensor(cnt, f64))
        most = wire.zeros(1, i64)
        if wire.is_root:
            counts = _host(CNT)[:, 2]
            overflow = bool((counts < 0).any()) or counts.sum() > a
            most[0] = 0 if overflow else int(counts.max())
        wire.bcast(most)
        most = int(most.item())
        T = R = None
        if most > 0:
            t_t, r_t = wire.zeros((most, 2), i32), wire.zeros(most, f64)
            if ans is not None and ans[2] is not None and len(ans[2]):
                tt = np.array(ans[2], dtype=np.int32, copy=True)
                tt[:, 0] += row0                                                    # block rows -> matrix rows
                t_t[:len(tt)] = wire.tensor(tt, i32)
                r_t[:len(tt)] = wire.tensor(ans[3], f64)
            T, R = wire.gather(t_t), wire.gather(r_t)
        if wire.is_root:
            c = _host(CNT)
            viol, loose, bound = int(c[:, 0].sum()), int(c[:, 1].sum()), float(c[:, 3].sum())
            if overflow:
                out = (viol, loose, None, None, bound)
            elif most == 0:
                out = (viol, loose, np.zeros((0, 2), dtype=np.int32), np.zeros(0), bound)
            else:
                lens = c[:, 2].astype(np.int64)
                out = (viol, loose, np.concatenate([_host(T[g, :lens[g]]) for g in range(wire.world)]),
                       np.concatenate([_host(R[g, :lens[g]]) for g in range(wire.world)]), bound)
    if wire.is_root:
        st = _host(status)
        bad = [_message(st[g, 2:]) for g in range(wire.world) if st[g, 0]]
        if bad:                                           # raised with every collective of the query completed on every rank
            raise RuntimeError("query %r failed on " % (OP_NAMES[op],) + "; ".join(bad))
    return op, out


class ShardedMatrix:
    """lsap.DeviceMatrix's interface over row blocks on several ranks — the object the ROOT hands to lsap.solve_core /
    lsap.certify.  `locals_` are this rank's blocks of the matrices that may be queried (e.g. a hypothesis and its twin)."""

    def __init__(self, locals_, which, wire):
        self.locals, self.which, self.wire = locals_, which, wire
        self.shape = (wire.nr, wire.nc)

    def _ask(self, op, **args):
        return _query(self.wire, self.locals, (op, self.which, args))[1]

    def row_select(self, v, k):
        return self._ask(OP_ROW_SELECT, v=v, k=int(k))

    def diagonal(self, n):
        return self._ask(OP_DIAGONAL)[:n]

    def col_min(self):
        return self._ask(OP_COL_MIN)

    def bid(self, v, rows):
        rows = np.asarray(rows, dtype=np.int64)
        if rows.size == 0:
            return np.zeros(0, np.int32), np.zeros(0), np.zeros(0)
        return self._ask(OP_BID, v=v, rows=rows)

    def entries(self, rows, cols):
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        if rows.size == 0:
            return np.zeros(0)
        return self._ask(OP_ENTRIES, rows=rows, cols=cols)

    def certificate(self, u, v, col4row, delta, eps, cap):
        return self._ask(OP_CERTIFICATE, u=u, v=v, col4row=col4row, delta=delta, eps=eps, cap=int(cap))

    def stop(self):
        _query(self.wire, self.locals, (OP_STOP, 0, {}))


def serve(locals_, wire):
    """A worker's side: answer the root's queries about this rank's blocks until it sends the stop header."""
    while _query(wire, locals_)[0] != "stop":
        pass


def _settle(M, sol, info, out, slot):
    c4r = lsap.resolve_near_ties(M, sol, info)
    if c4r is None:
        return False
    out[slot] = c4r
    info["route"] = "sharded device (optimal; near-tie settled by the dense algorithm on %d row(s))" % sum(info["resolved_groups"])
    return True


def solve_pair_sharded(local_h, local_twin, bounds, n_cols, group, root, info=None, accept_near_ties=False, settle_near_ties=True):
    """One hypothesis and (optionally) its twin, rows sharded: the root solves the first on its sparse core and certifies the
    result on both matrices; a twin that does not accept its sibling's duals is solved on its own core.  -> (col4row of the
    hypothesis or None, col4row of the twin or None) on EVERY rank; None = not certified (the caller takes another route
    for that matrix).  N <= M required.  accept_near_ties: an assignment certified optimal but not proven unique
    (lsap.certify: info["optimal"]) is returned instead of None; info["near_tie"] lists which (0 = hypothesis, 1 = twin).
    settle_near_ties: a certified optimum with alternatives inside the margin is settled on the blocks its near-tight entries connect
    (lsap.resolve_near_ties).  Callers that CAN hand the whole matrix to SciPy's algorithm instead (it fits the dense solver) pass
    False and take that route — for exact ties SciPy's pick on a block need not be its pick on the whole matrix."""
    wire = _Wire(bounds, n_cols, group, root)
    locals_ = [local_h] + ([local_twin] if local_twin is not None else [])
    out = [None, None, None]                              # col4row, twin's col4row, error message
    if wire.is_root:
        try:
            info = {} if info is None else info
            near = info["near_tie"] = []
            M = ShardedMatrix(locals_, 0, wire)
            sol = lsap.solve_core(M, info)
            certified = sol is not None and lsap.certify(M, *sol, info=info)
            if certified:
                out[0] = sol[2]
            elif settle_near_ties and sol is not None and info.get("optimal") and _settle(M, sol, info, out, 0):
                pass                                      # near-ties settled on their blocks (lsap.resolve_near_ties: a few entries travel)
            elif accept_near_ties and sol is not None and info.get("optimal"):
                out[0] = sol[2]
                near.append(0)
            if local_twin is not None:
                Mt = ShardedMatrix(locals_, 1, wire)
                tinfo = info["twin"] = {}
                if certified and lsap.certify(Mt, *sol, info=tinfo):
                    out[1] = sol[2]
                    tinfo["route"] = "sibling's duals certified"
                else:                                     # the twin on its own
                    tinfo.clear()
                    sol_t = lsap.solve_core(Mt, tinfo)
                    if sol_t is not None and lsap.certify(Mt, *sol_t, info=tinfo):
                        out[1] = sol_t[2]
                    elif settle_near_ties and sol_t is not None and tinfo.get("optimal") and _settle(Mt, sol_t, tinfo, out, 1):
                        pass
                    elif accept_near_ties and sol_t is not None and tinfo.get("optimal"):
                        out[1] = sol_t[2]
                        near.append(1)
                    tinfo["route"] = "own core"
        except Exception as e:                            # the workers are waiting for queries: release them, then raise everywhere
            out[2] = "%s: %s" % (type(e).__name__, e)
        finally:
            ShardedMatrix(locals_, 0, wire).stop()
    else:
        serve(locals_, wire)
    return _share_result(wire, out, root)


def _share_result(wire, out, root):
    """out = [col4row of the hypothesis | None, of the twin | None, error message | None] on the root -> the two vectors on EVERY
    rank: int64 (has hypothesis, has twin, failed) | int32 [2, N] | the message if it failed (raised on all ranks)."""
    torch = _torch()
    if wire.is_root:
        res = wire.tensor(np.array([out[0] is not None, out[1] is not None, out[2] is not None], dtype=np.int64), torch.int64)
        both = np.zeros((2, wire.nr), dtype=np.int32)
        for k in range(2):
            if out[k] is not None:
                both[k] = np.asarray(out[k])
        c4r = wire.tensor(both, torch.int32)
    else:
        res, c4r = wire.empty(3, torch.int64), wire.empty((2, wire.nr), torch.int32)
    wire.bcast(res)
    wire.bcast(c4r)
    has_h, has_t, err = (int(x) for x in _host(res))
    if err:
        msg = wire.bcast(_status(wire, 1, 0, out[2] if wire.is_root else ""))
        raise RuntimeError("sharded assignment failed on rank %d: %s" % (root, _message(_host(msg)[2:])))
    got = _host(c4r)
    return (got[0].copy() if has_h else None), (got[1].copy() if has_t else None)


def solve_pair_sharded_filtered(local_filter, exact_entries, cost_delta, bounds, n_cols, group, root, info=None, exact_entries_t=None,
                                entries_device=None):
    """One pairing (hypothesis + twin) from ROW BLOCKS OF ITS FILTER MATRIX (pipeline.assign_sharded_filtered): local_filter is this
    rank's block of an approximate matrix within cost_delta of both exact matrices (lsap.DeviceMatrix over the float32 block; the CPU
    tests pass a NumPy double).  The root runs lsap.solve_core on a lsap.FilteredMatrix whose selector is the ShardedMatrix over
    those blocks — column minima, row selection and the listing pass are answered by every rank for its rows — and whose costs are
    exact_entries(rows, cols) -> (hypothesis's exact values, twin's), evaluated on the root alone (only the root calls it); the
    result is certified against both exact matrices on their listed entries (lsap.certify_listed: the exact mode's own margins).
    -> (col4row, col4row) on every rank, or (None, None) if the pairing could not be proven (the caller builds it exactly)."""
    wire = _Wire(bounds, n_cols, group, root)
    locals_ = [local_filter]
    out = [None, None, None]
    if wire.is_root:
        try:
            info = {} if info is None else info
            tinfo = info["twin"] = {}
            M = lsap.FilteredMatrix(ShardedMatrix(locals_, 0, wire), exact_entries, cost_delta, exact_entries_t=exact_entries_t,
                                    entries_device=entries_device)
            sol = lsap.solve_core(M, info)
            if sol is not None:
                ok = lsap.certify_listed(M, *sol, exact_entries=exact_entries, cost_delta=cost_delta, infos=[info, tinfo])
                if len(ok) == 2 and all(ok):
                    out[0] = out[1] = sol[2]
                    info["exact_evaluated"] = M.exact_evaluated
        except Exception as e:                            # the workers are waiting for queries: release them, then raise everywhere
            out[2] = "%s: %s" % (type(e).__name__, e)
        finally:
            ShardedMatrix(locals_, 0, wire).stop()
    else:
        serve(locals_, wire)
    return _share_result(wire, out, root)


# ---- several pairings at once over ONE ordered sequence of collectives (round 5) ----------------------------------------------------
# solve_pair_sharded(_filtered) settles one pairing at a time: its root's host solver (0.3-0.4 s at 50 000 nuclei) runs while every
# other rank waits for the next query, and the four pairings of a registration follow one another — a sharded 50k registration then
# costs 4 x the one-GPU critical path.  Here the roots (pairing t -> rank t mod G) run their host solvers CONCURRENTLY, each on a
# thread of its own rank, and one loop on every rank serves all of them: per round ONE all-reduce carries the pending query headers
# of all pairings (a root with nothing to ask contributes zeros), then the pending queries are answered in pairing order with
# exactly the collectives of _query.  Every rank issues the same collectives in the same order on one communicator — no second
# communicator, no tags — and a root waits at most one round (~0.1 ms) for its turn.
class _Channel:
    """Between a root's solver thread and the serving loop of its rank."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.asked = threading.Event()
        self.answered = threading.Event()
        self.request = None              # (op, which, args) | "done"
        self.answer = None               # the gathered answer | an exception to raise in the solver

    def ask(self, request):              # solver thread
        with self.lock:
            self.request = request
            self.answered.clear()
            self.asked.set()
        self.answered.wait()
        ans, self.answer = self.answer, None
        if isinstance(ans, BaseException):
            raise ans
        return ans

    def pending(self):                   # serving loop: the request waiting for service, or None
        return self.request if self.asked.is_set() else None

    def deliver(self, answer):           # serving loop
        with self.lock:
            self.request = None
            self.asked.clear()
            self.answer = answer
            self.answered.set()

    def finish(self):                    # solver thread: nothing more to ask
        with self.lock:
            self.request = "done"
            self.asked.set()


class MultiplexedMatrix(ShardedMatrix):
    """ShardedMatrix whose queries go through the serving loop of solve_pairs_sharded_filtered instead of straight onto the wire."""

    def __init__(self, locals_, which, wire, channel):
        super().__init__(locals_, which, wire)
        self.channel = channel

    def _ask(self, op, **args):
        return self.channel.ask((op, self.which, args))


def solve_pairs_sharded_filtered(jobs, group, cost_delta, poll_s=1e-4):
    """Several pairings from row blocks of their filter matrices, the roots' host solvers running concurrently.
    jobs: list of dicts {local: this rank's block of the pairing's filter matrix, exact_entries: (rows, cols) -> (exact hypothesis
    values, exact twin values) — called on the pairing's root only —, bounds, n_cols, root, info (dict or None), device / stream
    (optional: the GPU and stream the root's exact evaluations belong to — a new thread would start on device 0's default stream),
    exact_entries_t (optional: the same with GPU tensors in and out; the root then finishes its selections on `device`)}.
    -> list of (col4row, col4row) | (None, None) per job, on every rank (as solve_pair_sharded_filtered)."""
    import threading
    import time
    torch, dist = _torch(), _dist()
    wires = [_Wire(j["bounds"], j["n_cols"], group, j["root"]) for j in jobs]
    n_jobs = len(jobs)
    outs = [[None, None, None] for _ in jobs]
    channels = [(_Channel() if w.is_root else None) for w in wires]
    threads = []
    my_roots = [k for k, w in enumerate(wires) if w.is_root]
    pin_base = lsap._pin_base()

    def solver(k, slot):
        job = jobs[k]
        dev, stream = job.get("device"), job.get("stream")
        if dev is not None:                                   # the root's exact evaluations run on the caller's device and stream
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                return solver_body(k, slot)
        return solver_body(k, slot)

    def solver_body(k, slot):
        lsap.pin_solver_thread(None if pin_base is None else pin_base + slot)
        job, out, ch = jobs[k], outs[k], channels[k]
        try:
            info = job["info"] if job.get("info") is not None else {}
            tinfo = info["twin"] = {}
            M = lsap.FilteredMatrix(MultiplexedMatrix([job["local"]], 0, wires[k], ch), job["exact_entries"], cost_delta,
                                    exact_entries_t=job.get("exact_entries_t"), entries_device=job.get("device"))
            sol = lsap.solve_core(M, info)
            if sol is not None:
                ok = lsap.certify_listed(M, *sol, exact_entries=job["exact_entries"], cost_delta=cost_delta, infos=[info, tinfo])
                if len(ok) == 2 and all(ok):
                    out[0] = out[1] = sol[2]
                    info["exact_evaluated"] = M.exact_evaluated
        except Exception as e:           # noqa: BLE001 — travels to every rank through _share_result
            out[2] = "%s: %s" % (type(e).__name__, e)
        finally:
            ch.finish()

    for slot, k in enumerate(my_roots):
        th = threading.Thread(target=solver, args=(k, slot), name="pm-sharded-root-%d" % k, daemon=True)
        th.start()
        threads.append(th)
    w0 = wires[0]
    active = set(range(n_jobs))
    results = [None] * n_jobs
    failure = None
    while active:
        h = np.zeros((n_jobs, 8), dtype=np.int64)
        asked = {}
        for k in my_roots:
            if k not in active:
                continue
            req = channels[k].pending()
            if req is None:
                continue
            asked[k] = req
            h[k, 0] = 1
            h[k, 1:6] = (OP_STOP, 0, 0, 0, 0) if req == "done" else _header_of(req)
        hdr = w0.tensor(h, torch.int64)
        dist.all_reduce(hdr, op=dist.ReduceOp.SUM, group=group)
        hh = _host(hdr)
        if not hh[:, 0].any():
            time.sleep(poll_s)               # every root is in its host phase: do not spin on the wire
            continue
        for k in sorted(active):
            if not hh[k, 0]:
                continue
            header = tuple(int(x) for x in hh[k, 1:6])
            if header[0] == OP_STOP:
                try:
                    results[k] = _share_result(wires[k], outs[k], jobs[k]["root"])
                except RuntimeError as e:     # the root's solver failed: raised on every rank, after the other pairings are through
                    failure = failure or e
                    results[k] = (None, None)
                active.discard(k)
                continue
            try:
                _, ans = _query(wires[k], [jobs[k]["local"]], asked.get(k), header=header)
                if wires[k].is_root:
                    channels[k].deliver(ans)
            except RuntimeError as e:         # a rank's share failed: every collective of the query has completed everywhere
                if wires[k].is_root:
                    channels[k].deliver(e)
    for th in threads:
        th.join()
    if failure is not None:
        raise failure
    return results

