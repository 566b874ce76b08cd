set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/lsap_probe.py 50000 42 > gpurun_out/lsap_probe_50k.log 2>&1; grep -v amdgpu gpurun_out/lsap_probe_50k.log | grep "solve_eight" | cut -c1-100
(timeout -k 10 400 python tools/big_registration.py 100000 8000 50 > gpurun_out/big_100k.log 2>&1; echo "exit $?" >> gpurun_out/big_100k.log) & PID=$!
while kill -0 $PID 2>/dev/null; do sleep 20; echo "[100k running]"; done
grep -v amdgpu gpurun_out/big_100k.log | tail -4 | cut -c1-400
for r in 1 2; do
(timeout -k 10 300 python tools/batch_throughput.py --pairs 64 --workers 5 --json gpurun_out/batch64_r$r.json > gpurun_out/batch64_r$r.log 2>&1; echo "exit $?" >> gpurun_out/batch64_r$r.log) & PID=$!
while kill -0 $PID 2>/dev/null; do sleep 20; echo "[batch running]"; done
grep -v amdgpu gpurun_out/batch64_r$r.log | tail -2 | cut -c1-330
done
