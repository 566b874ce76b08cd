set -e
cd $GRAFT_REPO_ROOT
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "exit $?" >> gpurun_out/gpu_suite.log) & PID=$!
while kill -0 $PID 2>/dev/null; do sleep 30; echo "[suite running]"; done
tail -3 gpurun_out/gpu_suite.log
for n in 5000 20000 50000; do
  timeout -k 10 300 python tools/e2e_timing.py $n > gpurun_out/e2e_$n.log 2>&1
  grep -v amdgpu gpurun_out/e2e_$n.log | tail -2 | cut -c1-330
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
