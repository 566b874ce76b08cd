set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout -k 10 500 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench exit $?") 
tail -c 600 gpurun_out/r02_bench.json | cut -c1-600
echo "[prof] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-assignment > gpurun_out/prof_bench.log 2>&1
echo "[prof] pmc fetch"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH -- python3 tools/profile_build.py > gpurun_out/pmc_fetch.log 2>&1
echo "[prof] pmc write"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE -- python3 tools/profile_build.py > gpurun_out/pmc_write.log 2>&1
find gpurun_out/prof_bench gpurun_out/pmc_FETCH gpurun_out/pmc_WRITE -name "*.csv" | head -20
