"""Occupancy guard (build container, no GPU): the wide kernels of the path keep the register / scratch budgets their design
assumes.  A rare branch compiled into a hot kernel costs every wave its registers and nothing else shows it — round 4 found
mean_distance_chunks at 191 vector registers (two waves per SIMD) because of the route one piece in 150 000 takes.  The numbers
come from the code object metadata of `hipcc -S` (tools/kernel_resources.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not found")     # (build._hipcc's own search)

#          file                    kernel name starts with                              max vgpr  scratch allowed
BUDGETS = [("pm_stats.hip", "pm::mean_distance_chunks", 96, False),
           ("pm_shape_context.hip", "void pm::sc_tile_kernel<4>", 64, False),
           ("pm_shape_context.hip", "void pm::sc_tile_kernel<2>", 64, False),
           ("pm_chi2.hip", "void pm::chi2_sym_kernel<4, 2, -1, 94, false>", 256, False),      # two waves per SIMD by design (220 registers)
           ("pm_chi2.hip", "pm::counts_extract_kernel", 64, False),
           ("pm_icp_grid.hip", "void pm::icp_iter_kernel<4, false>", 128, False),
           ("pm_icp_grid.hip", "void pm::icp_iter_kernel<8, false>", 128, False),
           ("pm_lsap_dev.hip", "void pm::row_select_kernel<true, double>", 32, False),
           ("pm_lsap_dev.hip", "void pm::certificate_kernel<double>", 32, False)]


@pytest.fixture(scope="module")
def rows():
    import kernel_resources as KR
    return KR.census(sorted({b[0] for b in BUDGETS}))


@pytest.mark.parametrize("file,prefix,max_vgpr,scratch_ok", BUDGETS)
def test_hot_kernels_keep_their_register_budget(rows, file, prefix, max_vgpr, scratch_ok):
    match = [r for r in rows if r["file"] == file and (r["name"].startswith(prefix) or prefix in r["name"])]
    if not match:
        pytest.skip("kernel names not demangled on this host (no c++filt)")
    for r in match:
        assert r["vgpr"] <= max_vgpr, r            # (.vgpr_count is the unified total on gfx950: it contains the AGPRs)
        assert r["spills"] == 0 and (scratch_ok or r["scratch"] == 0), r
