# round 3: parity of the tile kernel, then a kernel trace of the bench (descriptor + ICP kernels)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "hist or tile or degenerate or end_to_end or neighbor or config4 or sampled or symmetry or frame" > gpurun_out/r03_sc_tests.log 2>&1 || (tail -40 gpurun_out/r03_sc_tests.log; exit 1)
tail -5 gpurun_out/r03_sc_tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03c -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-assignment > gpurun_out/prof_r03c.log 2>&1
tail -c 1500 gpurun_out/prof_r03c.log | cut -c1-1500
find gpurun_out/prof_r03c -name "*kernel_stats.csv" | head -3
f=$(find gpurun_out/prof_r03c -name "*kernel_stats.csv" | head -1)
head -25 "$f"
