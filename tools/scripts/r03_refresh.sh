# round 3: the evidence files DESIGN.md quotes (bench line, kernel trace, PMC passes, complete registrations, the batch)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "[bench]"; (timeout -k 10 500 python bench.py > gpurun_out/r03_bench50k.json 2> gpurun_out/r03_bench50k.err; echo "bench rc=$?")
echo "[kernel trace]"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-assignment > gpurun_out/prof_r03.log 2>&1; echo "rc=$?"
echo "[pmc fetch]"; timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r03_FETCH -- python3 tools/profile_build.py > gpurun_out/pmc_fetch.log 2>&1; echo "rc=$?"
echo "[pmc write]"; timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r03_WRITE -- python3 tools/profile_build.py > gpurun_out/pmc_write.log 2>&1; echo "rc=$?"
find gpurun_out/prof_r03 gpurun_out/pmc_r03_FETCH gpurun_out/pmc_r03_WRITE -name "*.csv" | head
echo "[e2e]"; (for n in 5000 20000 50000; do timeout -k 10 300 python tools/e2e_timing.py $n; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_e2e.txt; tail -4 gpurun_out/r03_e2e.txt | cut -c1-300
echo "[batch seeded]"; timeout -k 10 300 python tools/batch_throughput.py --pairs 64 --workers 8 --json gpurun_out/r03_batch64_seeded.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "[batch unseeded]"; timeout -k 10 300 python tools/batch_throughput.py --pairs 64 --workers 8 --unseeded --json gpurun_out/r03_batch64_unseeded.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
