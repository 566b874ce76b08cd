"""The product's only compute backend: the HIP kernels behind libplatymatch_hip.so, as the registration driver (pipeline.py) calls them.
(The CPU tests substitute a double with the same methods built on the oracle; there is no CPU fallback in the product.)"""
import numpy as np

from . import _native as nat
from .cost_buffers import kept_cost_bytes
from .device_memory import idle_bytes

RELAXED_VARIANT = 2            # pm_chi2_cost8_relaxed: 0 all computed, 1 94 x 94 term table, 2 64 x 64 table at three waves per SIMD (fastest at 50k)


class GpuBackend:
    """The product's only compute backend: the HIP kernels behind libplatymatch_hip.so."""

    device_sampler = True      # do_ransac can draw its index sets on the device (unseeded runs)

    def __init__(self, dev=None):
        from . import _kernels
        self.K = _kernels
        self.device = nat.device(dev)

    def cloud(self, x):
        t = nat.to_dev(x, dev=self.device)
        if t.dim() != 2 or t.shape[0] not in (3, 4):
            raise ValueError("clouds must be 3 x N (or 4 x N)")
        return t[:3, :].contiguous()

    def axis(self, xyz, view=None):
        """First PCA axis of the cloud as the reference gets it from sklearn (shape_context.py:162-165): sklearn's own NumPy calls on
        the host (shape_context.pca_axis_host: the reference's bits), on `view` — the N x 3 array the reference would pass, made
        from the caller's own array (shape_context.pca_view) — or, without one, on a host copy of the device cloud."""
        from .estimate_transform.shape_context import pca_axis_host
        if view is None:
            view = xyz.cpu().numpy().transpose()
        torch = nat.torch_mod()
        host = torch.from_numpy(np.ascontiguousarray(pca_axis_host(view), dtype=np.float64).reshape(3)).pin_memory()
        return host.to(xyz.device, non_blocking=True)     # queued behind this stream's launches: a pageable copy would make the host wait for them

    def stats(self, xyz, view=None):
        if view is None:
            view = xyz.cpu().numpy().transpose()      # read back BEFORE this call's launches are queued in front of the copy
        c, md = self.K.centroid(xyz), self.K.mean_distance(xyz)
        return c, md, self.axis(xyz, view)            # the host's ~0.3 ms of NumPy run while the device sums the pair distances

    def mean_distance_partials(self, xyz, row_offset, row_stride):
        return self.K.mean_distance_partials(xyz, row_offset, row_stride)

    def mean_distance_finish(self, partials, n):
        return self.K.mean_distance_finish(partials, n)

    def centroid_and_axis(self, xyz, view=None):
        if view is None:
            view = xyz.cpu().numpy().transpose()
        return self.K.centroid(xyz), self.axis(xyz, view)

    def shape_context(self, xyz, c, md, x0, nf, row0, nrows, guards=None):
        r = self.K.shape_context(xyz, c, x0, md, nf, row0=row0, nrows=nrows)
        if guards is not None:
            guards.append(r["guard"])         # int32 GPU [2]: neighbours too close to a ring radius / sector edge (edge guard)
        return r["hist"]

    def symmetry_flag(self, sc_m, sc_f):
        return self.K.chi2_symmetry_flag(sc_m, sc_f)

    def chi2_cost8(self, sc_m, sc_f, out=None):
        if sc_f.shape[0] == 1:            # frame 1 only: gather_fixed_descriptors verified the permutation relation
            return self.K.chi2_cost8_frame1(sc_m[0], sc_f[0], out=out)
        return self.K.chi2_cost8(sc_m, sc_f, out=out)

    def chi2_cost8_relaxed(self, sc_m, sc_f, out=None):
        """The eight matrices in relaxed float64 arithmetic (K.chi2_cost8_relaxed) -> (U, delta).  Only where the frame-permutation
        relation holds: the caller has checked it (chi2_symmetric / the sharded gather's verdict)."""
        return self.K.chi2_cost8_relaxed(sc_m[0], sc_f[0], out=out, variant=RELAXED_VARIANT), self.K.chi2_relaxed_delta()

    def chi2_cost_pair_into(self, sc_m1, sc_f1, pairing, out8):
        """One pairing's two EXACT matrices written over their slots of an eight-matrix buffer (the relaxed route's rebuild)."""
        return self.K.chi2_cost_pair_into(sc_m1, sc_f1, pairing, out8)

    def chi2_cost_pair(self, sc_m, sc_f, pairing, out=None):
        """The two matrices of one pairing (hypothesis + twin) only -> [2, rows, M]; symmetric iff sc_f holds frame 1 only or
        the permutation relation checks out (symmetric_hint caches the check of the whole-cloud call)."""
        sym = sc_f.shape[0] == 1 or self.K.chi2_symmetric(sc_m, sc_f)
        return self.K.chi2_cost_pair(sc_m, sc_f, pairing, sym, out=out)

    def chi2_symmetric(self, sc_m, sc_f):
        """Do frames 2..4 permute frame 1's phi sectors bit for bit on these rows (the premise of the half-cost, relaxed and filter
        builds)?  One pass over the descriptors."""
        return sc_f.shape[0] == 1 or self.K.chi2_symmetric(sc_m, sc_f)

    def chi2_filter4(self, a1, b1, out=None, dtype=None):
        """The four pairings' FILTER matrices [4, rows of a1, rows of b1] (packed float32 arithmetic; within chi2_filter_delta() of
        the exact costs) — a1's rows may be a rank's block."""
        return self.K.chi2_filter4(a1, b1, out=out, dtype=dtype)

    def chi2_filter_pair(self, a1, b1, pairing, out=None, dtype=None):
        return self.K.chi2_filter_pair(a1, b1, pairing, out=out, dtype=dtype)

    def chi2_filter_delta(self):
        return self.K.chi2_filter_delta()

    def chi2_entries(self, sc_m1, sc_f1, pairing, rows, cols, trusted=False):
        """Listed entries of pairing t's two EXACT matrices (hypothesis, twin) -> two float64 GPU tensors."""
        return self.K.chi2_entries(sc_m1, sc_f1, pairing, rows, cols, trusted=trusted)

    def chi2_cost_single(self, scA, scB):
        """One matrix chi2(scA[i], scB[j]) for any two descriptor sets [*, 360] (pm_chi2_cost)."""
        return self.K.chi2_cost(scA.contiguous(), scB.contiguous())

    def free_bytes(self):
        """Device memory a new allocation can draw on: what the driver reports free plus what torch's caching allocator holds
        without using (the eight matrices of a previous registration sit there: counting them as taken would send the next
        registration of the same size into the streamed mode)."""
        import torch
        cached = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        # (+ this stream's kept buffer: it IS the room; + the idle raw blocks of device_memory.py: released when an allocation needs them)
        return torch.cuda.mem_get_info(self.device)[0] + max(int(cached), 0) + kept_cost_bytes(self.device) + idle_bytes(self.device)

    def row_argmin(self, U):
        return self.K.row_argmin(U)

    def draw_samples(self, n, min_samples, trials, rng=None):
        from .estimate_transform.shape_context import draw_ransac_samples
        return draw_ransac_samples(n, min_samples, trials, rng=rng)

    def do_ransac(self, mov, fix, rows, cols, trials, error, transform, min_samples, samples=None, device_seed=None, run=0, defer=None,
                  prelaunched=None):
        from .estimate_transform.shape_context import do_ransac
        return do_ransac(mov, fix, min_samples=min_samples, trials=trials, error=error, transform=transform,
                         rows=rows, cols=cols, samples=samples, device_seed=device_seed, run=run, defer=defer, prelaunched=prelaunched)

    def ransac_prelaunch(self, mov, fix, rows, cols, trials, error, min_samples, device_seed, run):
        from .estimate_transform.shape_context import ransac_prelaunch
        return ransac_prelaunch(mov, fix, rows, cols, min_samples, trials, error, device_seed, run)

    def refit_winner(self, deferred):
        """The chosen hypothesis's RANSAC model by the reference's own host expression (shape_context.refit_affine_winner)."""
        from .estimate_transform.shape_context import refit_affine_winner
        return refit_affine_winner(deferred)

    def fit(self, kp_m, kp_f, transform):
        from .estimate_transform.find_transform import get_affine_transform, get_similar_transform
        fn = get_affine_transform if transform == 'Affine' else get_similar_transform
        return nat.to_dev(fn(kp_m, kp_f), dev=self.device)

    def apply_affine(self, A, xyz):
        return self.K.apply_affine(A.reshape(16).contiguous(), xyz)

    def icp(self, mov, fix, iters, transform, log, one_launch=None):
        from .estimate_transform.perform_icp import perform_icp
        return perform_icp(mov, fix, iters, transform, log=log, one_launch=one_launch)

    def icp_grid(self, fix):
        """Bin the fixed cloud once per ICP run; the run owns the grid and hands it to icp_nn (no hidden backend state:
        a backend may be shared between threads and the allocator reuses addresses)."""
        return self.K.icp_grid(fix)

    def icp_nn(self, mov, fix, grid=None):
        return self.K.icp_nn(mov, fix, want_dist=False, grid=grid)[0]

    def icp_accumulate(self, mov, fix, nn, origin, out=None):
        return self.K.icp_accumulate(mov, fix, nn, origin, out=out, nn_trusted=True)     # nn is icp_nn's own output

    def icp_update(self, sums, origin, mov, fix, nn, A_icp, parts_out=None, status=None):
        return self.K.icp_update(sums, origin, mov, fix, nn, A_icp, parts_out=parts_out, nn_trusted=True, status=status)
