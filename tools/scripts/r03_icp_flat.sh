set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_icp_loop.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_degenerate.py -x -q -m gpu -k "icp or nn or planar or grid or one_launch or concurrent or fixture_through" > gpurun_out/r03_icp_tests.log 2>&1 || (tail -40 gpurun_out/r03_icp_tests.log; exit 1)
tail -3 gpurun_out/r03_icp_tests.log
for n in 5000 20000 50000; do timeout -k 10 120 python tools/icp_profile.py $n 200 2>&1 | grep -v amdgpu.ids | head -1; done | tee gpurun_out/r03_icp_timing_flat.txt
for defs in ""; do
  for n in 5000 50000; do PM_STAMPS_LOOP=0 PM_EXTRA_DEFINES="$defs" timeout -k 10 200 python tools/icp_stamps.py $n 20 2>&1 | grep -v amdgpu.ids | grep -E "diagnostic|n = |loaded|search done|arrival add|total fetched"; done
done | tee gpurun_out/r03_icp_stamps_flat.txt
