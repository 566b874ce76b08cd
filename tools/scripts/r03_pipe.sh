set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -x -q -m gpu -k "statistic or utils_mirror or mean_distance or rccl or collective or backend or driver or end_to_end or guard" 2>&1 | tail -6
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o st -- python3 tools/profile_build.py 50000 > gpurun_out/prof_stats.log 2>&1
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); grep -E "mean_distance|centroid|pca" "$f" | cut -c1-170
timeout -k 10 400 python tools/pipelined_assign_probe.py 50000 3 2>&1 | tee gpurun_out/r03_pipelined_probe.txt
