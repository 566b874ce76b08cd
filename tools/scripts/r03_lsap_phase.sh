set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in 5000 20000 50000; do timeout -k 10 300 python tools/lsap_phase_probe.py $n 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03_lsap_phases.txt
