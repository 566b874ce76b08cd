# round 3: ICP iteration kernel — parity tests, then time per iteration at 5k / 20k / 50k and a kernel trace at 50k
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_icp_loop.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_degenerate.py -x -q -m gpu -k "icp or nn or planar or grid or one_launch or concurrent or fixture_through" > gpurun_out/r03_icp_tests.log 2>&1 || (tail -40 gpurun_out/r03_icp_tests.log; exit 1)
tail -3 gpurun_out/r03_icp_tests.log
for n in 5000 20000 50000; do timeout -k 10 120 python tools/icp_profile.py $n 200 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03_icp_timing_${TAG:-x}.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_icp_${TAG:-x} -o icp -- python3 tools/icp_profile.py 50000 200 > gpurun_out/prof_icp.log 2>&1
f=$(find gpurun_out/prof_icp_${TAG:-x} -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-220
