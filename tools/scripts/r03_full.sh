# round 3: the whole GPU suite, the default bench, the sharded N > M rehearsal (3 ranks sharing the one GPU over gloo)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/r03_gpu_suite.log
tail -6 gpurun_out/r03_gpu_suite.log
(timeout -k 10 500 python bench.py > gpurun_out/r03_bench50k.json 2> gpurun_out/r03_bench50k.err; echo "bench rc=$?")
tail -c 400 gpurun_out/r03_bench50k.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r03_bench50k.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms")})
print(d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"]["hbm_frac"], d["roofline"]["attainable_hbm_frac_upper_bound"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:200])
P
PM_STREAM_HYPOTHESES=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 tools/two_rank_registration.py 6000 5000 2>&1 | grep -v "amdgpu.ids\|Gloo" | tail -8 | tee gpurun_out/r03_sharded_n_gt_m_rehearsal.txt
