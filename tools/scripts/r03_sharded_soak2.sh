cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
fail=0
run() { # world maxpoints seed streamed
  port=$((29600 + RANDOM % 300))
  out=$(PM_SOAK_CASE=1 PM_STREAM_HYPOTHESES=$4 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/two_rank_registration.py $2 $2 $3 2>&1)
  rc=$?
  echo "world $1 soak case $3 (up to $2 points) streamed=$4: rc=$rc $(echo "$out" | grep -c ' OK$') of 2 OK; $(echo "$out" | grep -m1 'assignment routes' | cut -c1-150)"
  if [ $rc -ne 0 ]; then fail=1; echo "$out" | grep -v Warning | tail -12; fi
}
for s in 4 9 14 19 24 3 7 12; do run 2 2600 $s 0; done
for s in 29 34 8 13; do run 3 2600 $s 1; done
echo "fail=$fail"
exit $fail
