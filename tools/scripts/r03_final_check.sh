cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/r03_gpu_suite.log
tail -4 gpurun_out/r03_gpu_suite.log
for n in 5000 20000 50000; do timeout -k 10 120 python tools/icp_profile.py $n 50 2>&1 | grep -v amdgpu.ids | head -1; done | tee gpurun_out/r03_icp_50iters.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_icp_final -o icp -- python3 tools/icp_profile.py 50000 50 > gpurun_out/prof_icp.log 2>&1
f=$(find gpurun_out/prof_icp_final -name "*kernel_stats.csv" | head -1); grep -E "grid_|icp_iter|fillBuffer" "$f" | cut -c1-150
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
