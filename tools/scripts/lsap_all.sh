set -e
cd $GRAFT_REPO_ROOT
for cfg in "20000 20000" "20000 19800" "20000 18000" "18000 20000" "50000 50000" "50000 47000" "47000 50000"; do
  set -- $cfg
  PM_LSAP_N=$1 PM_LSAP_M=$2 PM_LSAP_HYPS=0 timeout -k 10 300 python tools/lsap_probe.py $(( $1 > $2 ? $1 : $2 )) 42 > gpurun_out/lsap_all_$1_$2.log 2>&1
  echo "N=$1 M=$2: $(grep -v amdgpu gpurun_out/lsap_all_$1_$2.log | grep solve_eight | cut -c1-32)"
done
