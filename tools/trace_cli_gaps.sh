set -e
cd $GRAFT_REPO_ROOT
REPO=$(pwd); OUT=$REPO/gpurun_out/cli_gaps; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
sed -n '/^python - <<PY/,/^PY/p' profiles/run_profile_generator.sh | sed 's/range(40)/range(80)/' > /tmp/mk.sh
bash /tmp/mk.sh
python gen_3dphoto_dynamic.py --base /tmp/clidata --out /tmp/cliout0 --ckpt_path random:0 --inpaint builtin > /dev/null 2>&1
cd /tmp
for fill in builtin none; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/$fill -o c -- python $REPO/gen_3dphoto_dynamic.py --base /tmp/clidata --out /tmp/cliout_$fill --ckpt_path random:0 --inpaint $fill > $OUT/run_$fill.log 2>&1
tail -2 $OUT/run_$fill.log
python $REPO/tools/trace_gaps.py $OUT/$fill 0.5
rm -rf $OUT/$fill
done
