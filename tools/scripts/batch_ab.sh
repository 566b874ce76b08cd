set -e
cd $GRAFT_REPO_ROOT
for mode in "1 1" "0 0" "1 1" "0 0" "1 0" "0 1"; do
  set -- $mode
  (PM_CHI2_TERM_TABLE=$1 PM_LSAP_AUCTION=$2 timeout -k 10 300 python tools/batch_throughput.py --pairs 64 --workers 5 --json gpurun_out/batch64_t$1a$2.json > gpurun_out/batch64_t$1a$2.log 2>&1; echo "exit $?" >> gpurun_out/batch64_t$1a$2.log) & PID=$!
  while kill -0 $PID 2>/dev/null; do sleep 20; echo "[batch table=$1 auction=$2 running]"; done
  echo "table=$1 auction=$2: $(grep -v amdgpu gpurun_out/batch64_t$1a$2.log | grep registrations | cut -c1-330)"
done
