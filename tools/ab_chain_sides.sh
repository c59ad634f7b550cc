# same-box A/B: the moving-object chains on one side stream vs alternating over two (a chain may then take two pair launches)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sub"
J='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], "value %.1f pairs/s, pair launch %.1f us" % (d["value"], d["roofline"]["avg_launch_ms"] * 1e3))'
for i in 1 2 3; do
  $B 2>/dev/null | python -c "$J" "1 side stream "
  $B --chain-sides 2 2>/dev/null | python -c "$J" "2 side streams"
  $B --no-moving-object 2>/dev/null | python -c "$J" "no chain      "
done
