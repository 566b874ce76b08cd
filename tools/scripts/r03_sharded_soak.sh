cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
fail=0
run() { # world n m seed streamed
  port=$((29600 + RANDOM % 300))
  out=$(PM_STREAM_HYPOTHESES=$5 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/two_rank_registration.py $2 $3 $4 2>&1)
  rc=$?
  echo "world $1 N=$2 M=$3 seed $4 streamed=$5: rc=$rc $(echo "$out" | grep -c ' OK$') of 2 OK"
  if [ $rc -ne 0 ]; then fail=1; echo "$out" | tail -15; fi
}
run 2 1100 2900 1 0
run 2 2900 1100 2 0
run 3 2047 2049 3 0
run 3 2500 2500 4 1
run 4 1800 2600 5 1
run 4 2600 1800 6 1
run 2 1025 1025 7 0
run 5 3000 3100 8 0
run 3 1200 3000 9 1
run 2 3100 3000 10 1
echo "fail=$fail"
exit $fail
