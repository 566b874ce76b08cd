set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_degenerate.py tests/test_gpu_widget_replay.py tests/test_gpu_sampler.py tests/test_gpu_similar.py tests/test_next_rows.py -x -q -m gpu -k "ransac or end_to_end or estimate or fit or apply or transform or replay or sampler or similar or utils or icp or planar or coplanar or get_Y" 2>&1 | tail -5
timeout -k 10 500 python tests/probes/soak_parity.py ${SOAK_SECONDS:-240} 500 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_parity.txt
