"""The collectives bench.py issues at N > 1 (barrier, film sum-reduce to rank 0, MAX of a float64, SUM of int64s),
run over RCCL at world size 1 on a film the renderer has just written: the 1-GPU boxes cannot host two ranks, so
this is what of the N > 1 path can be exercised on real hardware besides the gloo tests. Launch:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_world1_check.py"""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
torch.cuda.init()
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
b = ge._load_binding()
scene = b.HostScene(path=os.path.join(REPO, "scenes", "killeroo-simple.pbrt"), xres=480, yres=270, spp=8)
gpu = b.GpuScene(scene)
h, w = scene.film_shape
film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
gpu.render(tile_rank=0, tile_nranks=1, film_device_ptr=film.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
before = film.clone()
dist.barrier()
dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
cnt = torch.tensor([1 << 40, 7, 9], dtype=torch.int64, device="cuda")
dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert torch.equal(film, before) and float(film.sum()) > 0
assert float(t.item()) == 1.25 and cnt.tolist() == [1 << 40, 7, 9]
dist.barrier()
dist.destroy_process_group()
print("rccl world-1 collectives ok; film sum", float(before.sum()))
