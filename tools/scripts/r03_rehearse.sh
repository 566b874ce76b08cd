set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export PM_BENCH_ONE_DEVICE=1 PM_BENCH_BACKEND=gloo
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r03_rehearse_2ranks_50k.json 2> gpurun_out/r03_rehearse_2ranks_50k.err || { tail -20 gpurun_out/r03_rehearse_2ranks_50k.err; exit 1; }
python - <<'P'
import json
for f in ("gpurun_out/r03_rehearse_2ranks_50k.json",):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["n_gpus"], d["ms_per_step"], d["value"], d["stage_ms"], d["config"]["sharding"])
P
timeout -k 10 300 python bench.py --gpus 4 --points 20000 --steps 2 --warmup 1 > gpurun_out/r03_rehearse_4ranks_20k.json 2> gpurun_out/r03_rehearse_4ranks_20k.err || { tail -20 gpurun_out/r03_rehearse_4ranks_20k.err; exit 1; }
python - <<'P'
import json
for f in ("gpurun_out/r03_rehearse_4ranks_20k.json",):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["n_gpus"], d["ms_per_step"], d["value"], d["stage_ms"], d["config"]["sharding"])
P
unset PM_BENCH_ONE_DEVICE PM_BENCH_BACKEND
timeout -k 10 300 python bench.py --points 20000 --steps 2 --warmup 1 > gpurun_out/r03_1rank_20k.json 2>/dev/null
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03_1rank_20k.json").read().strip().splitlines()[-1]); print("1 rank 20k", d["ms_per_step"], d["value"], d["stage_ms"])
P
