set -e
cd $GRAFT_REPO_ROOT
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "exit $?" >> gpurun_out/gpu_suite.log) & PID=$!
while kill -0 $PID 2>/dev/null; do sleep 30; echo "[suite running]"; done
tail -3 gpurun_out/gpu_suite.log
for n in 5000 20000 50000; do
  timeout -k 10 300 python tools/e2e_timing.py $n > gpurun_out/e2e_$n.log 2>&1
  grep -v amdgpu gpurun_out/e2e_$n.log | tail -2 | cut -c1-330
done
timeout -k 10 300 python tools/lsap_probe.py 50000 42 > gpurun_out/lsap_probe_50k.log 2>&1
bash tools/scripts/lsap_all.sh
(timeout -k 10 400 python tools/big_registration.py 100000 8000 50 > gpurun_out/big_100k.log 2>&1; echo "exit $?" >> gpurun_out/big_100k.log) & PID=$!
while kill -0 $PID 2>/dev/null; do sleep 20; echo "[100k running]"; done
grep -v amdgpu gpurun_out/big_100k.log | tail -4 | cut -c1-300
(timeout -k 10 500 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench exit $?")
