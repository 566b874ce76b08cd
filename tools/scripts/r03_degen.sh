set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_degenerate.py tests/test_gpu_parity.py tests/test_gpu_icp_loop.py tests/test_gpu_soak.py tests/test_gpu_fullsize.py -x -q -m gpu -k "icp or planar or degenerate or fit or soak or slice or lattice or end_to_end or estimate" 2>&1 | tail -4
timeout -k 10 60 python tests/probes/soak_parity_c.py 4 25712 2>&1 | grep -E "^soak C|^mismatches|^  seed"
timeout -k 10 60 python tests/probes/soak_parity_c.py 4 25791 2>&1 | grep -E "^soak C|^mismatches|^  seed"
timeout -k 10 60 python tests/probes/soak_parity_c.py 4 2727 2>&1 | grep -E "^soak C|^mismatches|^  seed"
timeout -k 10 200 python tests/probes/soak_parity_c.py 120 40000 2>&1 | grep -E "^soak C|^mismatches|^  seed"
