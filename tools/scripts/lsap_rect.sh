set -e
cd $GRAFT_REPO_ROOT

for cfg in "20000 18000" "20000 19800" "50000 47000" "50000 50000"; do
  set -- $cfg
  echo "== N=$1 M=$2"
  PM_LSAP_M=$2 PM_LSAP_HYPS=0,1 timeout -k 10 300 python tools/lsap_probe.py $1 42 > gpurun_out/lsap_rect_$1_$2.log 2>&1
  grep -v amdgpu gpurun_out/lsap_rect_$1_$2.log | grep "hyp\|solve_eight" | sed 's/certify.*rounds/ rounds/; s/violations.*//' | cut -c1-260
done
