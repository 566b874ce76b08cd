"""Achieved HBM rates of the kernels behind SURVEY §8f's rows (evaluation metrics, label image) and of the per-row arg-min:
all three are streaming kernels whose roofline is HBM bandwidth (algorithmic bytes / time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from platymatch_amd import _kernels as K


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, nbytes, ms):
    gbs = nbytes / ms / 1e6
    print("%-58s %8.3f ms  %7.1f GB/s  (%.0f %% of the 8 TB/s HBM peak)" % (name, ms, gbs, gbs / 80), flush=True)


n = 20000
a = torch.rand((3, n), dtype=torch.float64, device="cuda") * 300
b = torch.rand((3, n), dtype=torch.float64, device="cuda") * 300
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
report("pm_cdist %d x %d (8 B written per pair)" % (n, n), 8.0 * n * n, timed(lambda: K.cdist(a, b, out=out)))
# a nucleus-like label image: 12-voxel cubes of one label each on a 16-voxel lattice (42 % foreground, 32 768 labels)
g = torch.arange(512, device="cuda")
cell = (g // 16)
inside = (g % 16) < 12
lab = ((cell[:, None, None] * 32 + cell[None, :, None]) * 32 + cell[None, None, :] + 1).to(torch.int32)
lab = torch.where(inside[:, None, None] & inside[None, :, None] & inside[None, None, :], lab, torch.zeros_like(lab)).contiguous()
nl = int(lab.max()) + 1
report("pm_label_moments 512^3 int32, blobs (4 B read per voxel)", 4.0 * lab.numel(), timed(lambda: K.label_moments(lab, n_labels=nl)))
noise = torch.randint(1, 5000, (256, 256, 256), dtype=torch.int32, device="cuda")
report("pm_label_moments 256^3, a different label per voxel (worst case)", 4.0 * noise.numel(), timed(lambda: K.label_moments(noise, n_labels=5000)))
U = torch.rand((8, 6250, 50000), dtype=torch.float64, device="cuda")
report("pm_row_argmin 8 x 6250 x 50000 (8 B read per entry)", 8.0 * U.numel(), timed(lambda: K.row_argmin(U)))
