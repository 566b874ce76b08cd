cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/r03_gpu_suite.log
tail -4 gpurun_out/r03_gpu_suite.log
echo "[bench]"; (timeout -k 10 500 python bench.py > gpurun_out/r03_bench50k.json 2> gpurun_out/r03_bench50k.err; echo "bench rc=$?")
echo "[kernel trace]"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03f -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-assignment > gpurun_out/prof_r03f.log 2>&1; echo "rc=$?"
python - <<'P'
import json
d = json.load(open("gpurun_out/r03_bench50k.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms")})
P
head -12 gpurun_out/prof_r03f/bench_kernel_stats.csv | cut -c1-140
(for n in 5000 20000; do timeout -k 10 300 python tools/e2e_timing.py $n; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_e2e_final.txt; grep "unseeded" gpurun_out/r03_e2e_final.txt | cut -c1-60
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
