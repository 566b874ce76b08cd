cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {
  port=$((29600 + RANDOM % 300))
  echo "=== world $1 case $3 streamed=$4"
  PM_SOAK_CASE=1 PM_STREAM_HYPOTHESES=$4 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/two_rank_registration.py $2 $2 $3 2>&1 | grep -v "Warning\|amdgpu.ids\|^  s = \|^  sc = \|^\*\*\*\*\|OMP_NUM" | grep -E "Error|error:|raise |File \"/root/repo|ICP |MISMATCH|routes|N=|lattice" | head -30
}
run 2 2600 3 0
run 3 2600 29 1
run 3 2600 34 1
