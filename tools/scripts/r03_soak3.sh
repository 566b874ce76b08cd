cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/r03_gpu_suite.log
tail -4 gpurun_out/r03_gpu_suite.log
timeout -k 10 300 python tests/probes/soak_parity_c.py 150 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_parity_family_c.txt | tail -14
timeout -k 10 300 python tests/probes/soak_parity.py 120 500 2000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_parity_seeds2000.txt | tail -6
