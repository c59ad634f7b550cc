# same-box A/B of the moving-object chain underneath the pair launches: bench.py's value with the gather path, round 2's sort path and no chain
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/bench_chain.py
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sub"
J='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], "value %.1f pairs/s, pair launch %.1f us" % (d["value"], d["roofline"]["avg_launch_ms"] * 1e3))'
for i in 1 2; do
  $B 2>/dev/null | python -c "$J" "gather"
  $B --witness --tune fwarp_path=2 2>/dev/null | python -c "$J" "sort  "
  $B --no-moving-object 2>/dev/null | python -c "$J" "none  "
done
mkdir -p gpurun_out/chain_r5
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/chain_r5/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --pairs-per-step 16 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/chain_r5/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize.py gpurun_out/chain_r5 2>&1 | head -8
