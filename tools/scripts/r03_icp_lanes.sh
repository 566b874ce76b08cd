set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export PM_STAMPS_LOOP=0
for defs in "" "-DPM_GR_TRY_LANES=2" "-DPM_GR_TRY_LANES=1" "-DPM_GR_TRY_LANES=2 -DPM_GR_FEW_FROM=1024" "-DPM_GR_TRY_LANES=4 -DPM_GR_FEW_FROM=1024"; do
  for n in 5000 50000; do PM_EXTRA_DEFINES="$defs" timeout -k 10 200 python tools/icp_stamps.py $n 20 2>&1 | grep -v amdgpu.ids; done
done | tee gpurun_out/r03_icp_lanes.txt
