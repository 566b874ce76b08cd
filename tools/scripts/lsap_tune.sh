set -e
cd $GRAFT_REPO_ROOT
for cfg in "bids_per_row=40" "bids_per_row=120" "bids_per_row=400" "bids_per_row=120 PM_LSAP_eps_min=1e-5" "bids_per_row=120 PM_LSAP_factor=3"; do
  echo "== $cfg"
  env PM_LSAP_$cfg PM_LSAP_M=19800 PM_LSAP_HYPS=0,1,2 timeout -k 10 300 python tools/lsap_probe.py 20000 42 > gpurun_out/lsap_tune.log 2>&1
  grep -v amdgpu gpurun_out/lsap_tune.log | grep "hyp\|solve_eight" | sed "s/certify.*auction_violated/ auction_violated/; s/'dummy_scans'.*core_seconds/ core_seconds/; s/routes.*//; s/, 'violations.*//" | cut -c1-250
done
