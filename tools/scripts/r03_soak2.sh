set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_soak.py tests/test_gpu_parity.py tests/test_gpu_degenerate.py tests/test_gpu_widget_replay.py tests/test_gpu_sampler.py tests/test_gpu_batch.py tests/test_gpu_rccl.py -x -q -m gpu 2>&1 | tail -5
PM_E2E_REPEAT=3 timeout -k 10 200 python tools/e2e_timing.py 5000 2>&1 | grep -E "unseeded|whole" | cut -c1-70
timeout -k 10 500 python tests/probes/soak_parity.py ${SOAK_SECONDS:-200} 800 1000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_parity_b.txt
