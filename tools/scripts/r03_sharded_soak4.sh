cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
fail=0
run() { # world maxpoints seed streamed
  port=$((29600 + RANDOM % 300))
  out=$(PM_SOAK_CASE=1 PM_STREAM_HYPOTHESES=$4 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/two_rank_registration.py $2 $2 $3 2>&1)
  rc=$?
  echo "world $1 soak case $3 (up to $2 points) streamed=$4: rc=$rc $(echo "$out" | grep -c ' OK$') of 2 OK; $(echo "$out" | grep -m1 'assignment routes' | cut -c1-110)"
  if [ $rc -ne 0 ]; then fail=1; echo "$out" | grep -E "Error|MISMATCH|ICP " | grep -v "ChildFailed\|elastic" | head -6; fi
}
w=2
for s in $(seq 40 75); do
  st=$(( s % 2 ))
  run $w 2400 $s $st
  w=$(( w % 4 + 2 )); if [ $w -gt 4 ]; then w=2; fi
done
echo "fail=$fail"
exit $fail
