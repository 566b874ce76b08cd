set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_fullsize.py -x -q -m gpu -k "guard or histogram or tile or soak or slice or lattice or statistic or config" 2>&1 | tail -5
timeout -k 10 200 python tests/probes/soak_b_guard.py 153 3 0 1 2 4 5 6 12 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout -k 10 300 python tests/probes/soak_parity.py 60 500 0 2>&1 | grep -v amdgpu.ids | tail -3
