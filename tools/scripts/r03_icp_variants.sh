set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for defs in "" "-DGR_INFLIGHT=4" "-DGR_INFLIGHT=4 -DPM_GR_FEW_FROM=1073741824" "-DPM_GR_FEW_FROM=1073741824" "-DGR_INFLIGHT=3"; do
  for n in 5000 50000; do PM_EXTRA_DEFINES="$defs" timeout -k 10 200 python tools/icp_stamps.py $n 20 2>&1 | grep -v amdgpu.ids; done
done | tee gpurun_out/r03_icp_stamps_variants.txt
