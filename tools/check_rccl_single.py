"""GPU-box check of the RCCL code path with the one GPU a gpurun box has: world_size 1, backend nccl.
Run as: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/check_rccl_single.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from stnerf_amd import ops, synthetic as syn
from stnerf_amd.parallel import gather_tiles, make_row_renderer, render_view_sharded

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group(backend="nccl", device_id=dev)
t = torch.arange(35, dtype=torch.float32, device=dev).reshape(7, 5)
g = gather_tiles(t, 7 * dist.get_world_size())
assert torch.equal(g, t)
dist.barrier()
tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
assert float(tt.item()) == 1.5
print("rccl world", dist.get_world_size(), "ok: all_gather_into_tensor, barrier, all_reduce(MAX)")
dist.destroy_process_group()
