set -e
cd $GRAFT_REPO_ROOT
for n in 20000 50000; do
  for a in 0 1; do
    echo "== n=$n auction=$a"
    PM_LSAP_AUCTION=$a PM_LSAP_HYPS=0,1,2 timeout -k 10 300 python tools/lsap_probe.py $n 42 > gpurun_out/lsap_${n}_a$a.log 2>&1
    grep -v amdgpu gpurun_out/lsap_${n}_a$a.log | grep -v "violated per" | cut -c1-420
  done
done
