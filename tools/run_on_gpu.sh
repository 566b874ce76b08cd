#!/bin/bash
# One runner for the GPU-box jobs behind the files in profiles/ (replaces the 36 one-off tools/scripts/*.sh of rounds 2-3; their
# exact command sequences are in the git history).  Usage, from the build container:
#     gpurun --timeout 1100 -- 'bash tools/run_on_gpu.sh <tag> <job> [<job> ...]'         e.g.  r04 suite bench profile e2e
# Every job writes under gpurun_out/<tag>_*; copy what is to be kept into profiles/ (tools/collect_profiles.py <tag>).
# One GPU process at a time, every step under its own `timeout -k 10`; a failing step ends the call (set -e): no retries.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${1:?tag (e.g. r04)}; shift
O=gpurun_out
mkdir -p $O
for JOB in "$@"; do
  echo "[$TAG] $JOB"
  case $JOB in
    suite)      # the GPU parity suite as the driver runs it
      timeout -k 10 1000 python -m pytest tests -m gpu ${PYTEST_STOP:---maxfail=6} -q -s > $O/${TAG}_gpu_suite.log 2>&1; tail -3 $O/${TAG}_gpu_suite.log ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
    bench)      # the headline line (N = M = 50 000) with the CPU baseline and both extras
      timeout -k 10 600 python bench.py --steps ${STEPS:-20} --warmup 2 > $O/${TAG}_bench50k.json 2> $O/${TAG}_bench50k.err; tail -c 400 $O/${TAG}_bench50k.json ;;
    profile)    # rocprofv3 kernel trace of the bench command + the two PMC passes (separate runs, kernel-trace only beside --pmc)
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-assignment > $O/${TAG}_prof.log 2>&1
      timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_FETCH -- python3 tools/profile_build.py > $O/${TAG}_pmc_fetch.log 2>&1
      timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_WRITE -- python3 tools/profile_build.py > $O/${TAG}_pmc_write.log 2>&1
      find $O/${TAG}_prof $O/${TAG}_pmc_FETCH $O/${TAG}_pmc_WRITE -name "*.csv" | head ;;
    e2e)        # complete registrations, wall clock (seeded and unseeded)
      : > $O/${TAG}_e2e.txt
      for n in ${SIZES:-5000 20000 50000}; do timeout -k 10 400 python tools/e2e_timing.py $n 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_e2e.txt; done
      grep -E "unseeded" $O/${TAG}_e2e.txt | cut -c1-300 ;;
    lsap)       # the assignment stage phase by phase
      timeout -k 10 600 python tools/lsap_phase_probe.py ${SIZES:-5000 20000 50000} 2>&1 | grep -v amdgpu.ids > $O/${TAG}_lsap_phases.txt; grep "all eight" $O/${TAG}_lsap_phases.txt ;;
    batch)      # BASELINE config 5: 64 pairs of 2k-20k nuclei
      timeout -k 10 400 python tools/batch_throughput.py --pairs 64 --workers 8 --json $O/${TAG}_batch64_seeded.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
      timeout -k 10 400 python tools/batch_throughput.py --pairs 64 --workers 8 --unseeded --json $O/${TAG}_batch64_unseeded.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 ;;
    rankshare)  # one rank's share of the multi-GPU runs (prediction for the driver's SCALE curve)
      timeout -k 10 500 python tools/rank_share.py $O/${TAG}_rank_share.json 2>&1 | grep -v amdgpu.ids | tail -8 ;;
    big)        # 100 000 x 100 000 on one GPU (hypotheses streamed two matrices at a time)
      timeout -k 10 600 python tools/big_registration.py 100000 8000 50 2>&1 | grep -v amdgpu.ids > $O/${TAG}_registration_100k.txt; tail -4 $O/${TAG}_registration_100k.txt | cut -c1-400 ;;
    soak)       # random registrations against the oracle for SOAK_SECONDS (tests/probes/soak_parity.py; PM_SOAK_LOPSIDED=1 for the lopsided family)
      timeout -k 10 $(( ${SOAK_SECONDS:-240} + 120 )) python tests/probes/soak_parity.py ${SOAK_SECONDS:-240} ${SOAK_POINTS:-500} ${SOAK_SEED:-0} 2>&1 | grep -v amdgpu.ids > $O/${TAG}_soak_parity.txt || true; tail -6 $O/${TAG}_soak_parity.txt ;;
    asoak)      # the gate of cost_mode='auto' as the default: assignment vectors against cost_mode='exact' on adversarial registrations (tools/auto_soak.py)
      timeout -k 10 $(( ${SOAK_SECONDS:-600} + 240 )) python tools/auto_soak.py --seconds ${SOAK_SECONDS:-600} --seed0 ${SOAK_SEED:-0} --workers ${SOAK_WORKERS:-4} --filter-from ${SOAK_FILTER_FROM:-8192} --max-points ${SOAK_POINTS:-20000} 2>&1 | grep --line-buffered -v amdgpu.ids > $O/${TAG}_auto_soak_${SOAK_FILTER_FROM:-8192}_seed${SOAK_SEED:-0}.txt || true; tail -12 $O/${TAG}_auto_soak_${SOAK_FILTER_FROM:-8192}_seed${SOAK_SEED:-0}.txt ;;
    cold)       # the first registration of a fresh process against the following ones, default mode and exact, with and without reserve()
      : > $O/${TAG}_cold_start.txt
      for mode in auto exact; do timeout -k 10 300 python tools/cold_start.py ${COLD_N:-50000} $mode 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_cold_start.txt; echo >> $O/${TAG}_cold_start.txt; done
      timeout -k 10 300 python tools/cold_start.py ${COLD_N:-50000} auto --reserve 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_cold_start.txt
      grep -E "registration 1|registration 3|reserve" $O/${TAG}_cold_start.txt | cut -c1-260 ;;
    fphase)     # the default mode's filtered solves, query by query, side by side and alone
      echo "== solver threads pinned to one L3 domain each (the default)" > $O/${TAG}_filter_phases.txt
      timeout -k 10 300 python tools/filter_phase_probe.py ${FP_N:-50000} 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_filter_phases.txt
      echo "== PM_LSAP_PIN=0: placement left to the scheduler" >> $O/${TAG}_filter_phases.txt
      PM_LSAP_PIN=0 timeout -k 10 300 python tools/filter_phase_probe.py ${FP_N:-50000} 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_filter_phases.txt
      echo "== one pairing after the other" >> $O/${TAG}_filter_phases.txt
      timeout -k 10 300 python tools/filter_phase_probe.py ${FP_N:-50000} --alone 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_filter_phases.txt; grep -E "^==|registration|pairing [0-9]" $O/${TAG}_filter_phases.txt | cut -c1-330 ;;
    alloc)      # where a large fresh allocation's time goes
      timeout -k 10 300 python tools/alloc_probe.py ${ALLOC_GB:-40} 2>&1 | grep -v amdgpu.ids > $O/${TAG}_alloc_probe.txt; cat $O/${TAG}_alloc_probe.txt ;;
    icp)        # ICP per-iteration timing and phase stamps (diagnostic build)
      for n in 5000 20000 50000; do timeout -k 10 200 python tools/icp_profile.py $n 2>&1 | grep -v amdgpu.ids; done > $O/${TAG}_icp_timing.txt; tail -6 $O/${TAG}_icp_timing.txt ;;
    *) echo "unknown job $JOB"; exit 2 ;;
  esac
done
