cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-sub 2>&1 | tail -c 900
echo
MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python - <<'PY'
import torch, torch.distributed as dist
from mpiflow_amd import pipeline
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
t = torch.arange(7, dtype=torch.float64, device="cuda:0")
dist.all_reduce(t); dist.barrier()
s = pipeline.empty_stats(); s.update(pairs=5, sum_flow_mag=1.5, hole_px=10.0, max_flow_mag=3.0, wall_seconds=2.0, neg_min_flow=1.0)
print("rccl world=1 all_reduce ok", t.tolist(), pipeline.reduce_stats(s))
print(pipeline.mask_max_table(["a"], "/nonexistent", 0, 1))
dist.destroy_process_group()
PY
