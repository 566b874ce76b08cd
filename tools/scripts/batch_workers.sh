set -e
cd $GRAFT_REPO_ROOT
for w in 5 8 10 12 8 10; do
  (timeout -k 10 300 python tools/batch_throughput.py --pairs 64 --workers $w --json gpurun_out/batch64_w$w.json > gpurun_out/batch64_w$w.log 2>&1; echo "exit $?" >> gpurun_out/batch64_w$w.log) & PID=$!
  while kill -0 $PID 2>/dev/null; do sleep 20; echo "[batch workers=$w running]"; done
  echo "workers=$w: $(grep -v amdgpu gpurun_out/batch64_w$w.log | grep registrations | cut -c30-330)"
done
