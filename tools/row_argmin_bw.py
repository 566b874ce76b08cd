"""Effective HBM read rate of pm_row_argmin on a stack of cost matrices (8 bytes per entry, read once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from platymatch_amd import _kernels as K

rows, cols = int(sys.argv[1]) if len(sys.argv) > 1 else 6250, int(sys.argv[2]) if len(sys.argv) > 2 else 50000
U = torch.rand((8, rows, cols), dtype=torch.float64, device="cuda")
K.row_argmin(U)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    idx = K.row_argmin(U)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
assert torch.equal(idx.long(), U.argmin(-1))
print("row_argmin 8 x %d x %d: %.3f ms  %.1f GB/s (%.1f%% of 8 TB/s)" % (rows, cols, ms, U.numel() * 8 / ms / 1e6, U.numel() * 8 / ms / 1e6 / 80))
