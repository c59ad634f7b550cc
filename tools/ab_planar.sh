cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "planar" 2>&1 | tail -3
for m in 1 0; do for lds in 1 2 0; do echo "== layout 0 (planar [S,4,H,W]) mask=$m planar_lds=$lds"; python tools/bench_stage_b.py --layout 0 --mask $m --aux 0 --planar-lds $lds --variants 1 2>&1 | tail -4; done; done
echo "== layout 3 (split rgb + sigma) mask=1"; for lds in 1 2; do python tools/bench_stage_b.py --layout 3 --mask 1 --aux 0 --planar-lds $lds --variants 1 2>&1 | tail -3; done
python tools/soak_planar.py 2>&1 | tail -3
python tools/bench_shared_views.py
