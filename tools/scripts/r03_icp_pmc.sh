set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_icp -o icp -- python3 tools/icp_profile.py 50000 50 > gpurun_out/pmc_icp.log 2>&1 || (tail -20 gpurun_out/pmc_icp.log; exit 1)
f=$(find gpurun_out/pmc_icp -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in acc.items():
    if "icp_iter_kernel<4" in k or "icp_iter_kernelILi4" in k or "icp_iter" in k:
        n = max(cnt[k], 1)
        print(k, "launches", n, {c: round(x / n) for c, x in v.items()})
P
