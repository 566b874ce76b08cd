set -e
cd $GRAFT_REPO_ROOT
for tail in 16 64 100000; do
for n in 100000; do
(PM_LSAP_WAR_TAIL=$tail timeout -k 10 300 python tools/lsap_probe_pair.py $n 1 > gpurun_out/lsap_100k.log 2>&1; echo "exit $?" >> gpurun_out/lsap_100k.log) & PID=$!
while kill -0 $PID 2>/dev/null; do sleep 20; echo "[running]"; done
echo "tail=$tail $(grep -v amdgpu gpurun_out/lsap_100k.log | grep pairing | cut -c1-330)"
done
PM_LSAP_WAR_TAIL=$tail PM_LSAP_HYPS=0,1,2,3 timeout -k 10 300 python tools/lsap_probe.py 50000 42 > gpurun_out/lsap_50k_tail.log 2>&1
grep -v amdgpu gpurun_out/lsap_50k_tail.log | grep "hyp\|solve_eight" | sed "s/certify.*auction_bids/ auction_bids/; s/'rounds.*core_seconds/ core_seconds/; s/routes.*//; s/, 'violations.*//" | cut -c1-200
done
