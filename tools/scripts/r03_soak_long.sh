cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 700 python tests/probes/soak_parity.py 540 600 10000 2>&1 | grep --line-buffered -v amdgpu.ids | tee gpurun_out/r03_soak_parity_seeds10000.txt | tail -4
timeout -k 10 400 python tests/probes/soak_parity_c.py 240 20000 2>&1 | grep --line-buffered -v -E "amdgpu.ids|Warning|^  [a-z]" | tee gpurun_out/r03_soak_parity_family_c_seeds20000.txt | tail -3
timeout -k 10 300 python tests/probes/soak_parity_b.py 150 400 5000 2>&1 | grep --line-buffered -v -E "amdgpu.ids|Warning|^  [a-z]" | tee gpurun_out/r03_soak_parity_family_b_seeds5000.txt | tail -14
