"""How close the device's cloud statistics come to the reference fixtures and to the oracle (a probe, not a test; it lives under
tests/ because it calls the oracle).  Usage: python tests/probes/axis_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, oracle
from conftest import SCENARIOS, load_golden
from platymatch_amd import _kernels as K, _native as nat
oracle.build()
dev = torch.device("cuda:0")
worst_axis = worst_md = 0
for name in SCENARIOS:
    d = load_golden(name)
    for cloud, x0, md in ((d["moving"], d["x0_m"], float(d["mean_dist_m"])), (d["fixed"], d["x0_f"], float(d["mean_dist_f"]))):
        xyz = nat.to_dev(np.ascontiguousarray(cloud[:3]), dev=dev)
        a = K.pca_axis(xyz).cpu().numpy(); m = float(K.mean_distance(xyz).item())
        da = min(np.abs(a - x0).max(), np.abs(a + x0).max()); dm = abs(m - md) / md
        worst_axis = max(worst_axis, da); worst_md = max(worst_md, dm)
        print(name, "axis diff %.2e  mean-distance rel diff %.2e" % (da, dm))
for n in (5000, 50000):
    mv, fx, _ = bench.synth(n)
    for cloud in (mv, fx):
        xyz = nat.to_dev(cloud, dev=dev)
        a = K.pca_axis(xyz).cpu().numpy()
        o = oracle.pca_axis(cloud.T)
        print(n, "axis vs oracle (sklearn's algorithm restated) %.2e" % min(np.abs(a - o).max(), np.abs(a + o).max()))
print("worst", worst_axis, worst_md)
