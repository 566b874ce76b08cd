set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for defs in "-DPM_ICP_POLL_SLEEP=16" "-DPM_ICP_POLL_SLEEP=48" "-DPM_ICP_POLL_SLEEP=100"; do
  for n in 5000 50000; do PM_EXTRA_DEFINES="$defs" timeout -k 10 200 python tools/icp_stamps.py $n 20 2>&1 | grep -v amdgpu.ids; done
done | tee gpurun_out/r03_icp_stamps_loop_variants.txt
