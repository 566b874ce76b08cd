#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes -> profiles/pmc_traffic.json (read by bench.py for `roofline.traffic`).

Collect (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "HBM"):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH -- python3 tools/profile_build.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE -- python3 tools/profile_build.py
then:  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <N> <round tag>
Units and corrections as the guide prescribes: both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests
of wide streaming reads at 64 bytes, so it is doubled (an upper bound for kernels whose loads are narrower)."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                name = row["Kernel_Name"].split("(")[0]
                tot[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch_csv, write_csv, n, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    import hashlib
    csrc = os.path.join(ROOT, "platymatch_amd", "csrc")
    shas = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16] for f in ("pm_chi2.hip", "pm_shape_context.hip")}
    out = {"n": n, "round": tag, "source": [os.path.relpath(fetch_csv, ROOT), os.path.relpath(write_csv, ROOT)],
           # the kernel sources these counters were collected on: bench.py flags the figures as stale when a source has changed since
           "kernel_source_sha16": shas,
           "unit": "GB per launch (FETCH_SIZE KiB x 2 for gfx950 + WRITE_SIZE KiB)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f_kib, launches = fetch.get(k, (0.0, 0))
        w_kib, _ = write.get(k, (0.0, 0))
        out["kernels"][k] = {"launches": launches, "fetch_gb_raw": f_kib * 1024 / 1e9, "fetch_gb_corrected": 2 * f_kib * 1024 / 1e9,
                             "write_gb": w_kib * 1024 / 1e9, "traffic_gb": (2 * f_kib + w_kib) * 1024 / 1e9}
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)
    for k, v in out["kernels"].items():
        print("%-60s %8.3f GB/launch (fetch x2 %.3f + write %.3f), %d launches" % (k[:60], v["traffic_gb"], v["fetch_gb_corrected"], v["write_gb"], v["launches"]))


if __name__ == "__main__":
    main()
