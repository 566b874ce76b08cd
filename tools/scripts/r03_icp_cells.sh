set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for defs in "-DPM_GR_CELLS_PER_POINT=8" "-DPM_GR_CELLS_PER_POINT=16" "-DPM_GR_CELLS_PER_POINT=32" "-DPM_GR_CELLS_PER_POINT=64" "-DPM_GR_CELLS_PER_POINT=32 -DPM_GR_FEW_FROM=1073741824"; do
  for n in 5000 20000 50000; do PM_STAMPS_LOOP=0 PM_EXTRA_DEFINES="$defs" timeout -k 10 200 python tools/icp_stamps.py $n 20 2>&1 | grep -v amdgpu.ids | grep -E "TIMING|search done|total fetched"; done
done | tee gpurun_out/r03_icp_cells.txt
