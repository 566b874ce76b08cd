set -e
cd $GRAFT_REPO_ROOT
for mode in 0 1 0 1; do
  (PM_CHI2_TERM_TABLE=$mode timeout -k 10 300 python tools/batch_throughput.py --pairs 64 --workers 5 --json gpurun_out/batch64_t$mode.json > gpurun_out/batch64_t$mode.log 2>&1; echo "exit $?" >> gpurun_out/batch64_t$mode.log) & PID=$!
  while kill -0 $PID 2>/dev/null; do sleep 20; echo "[batch table=$mode running]"; done
  grep -v amdgpu gpurun_out/batch64_t$mode.log | tail -2 | cut -c1-330
done
