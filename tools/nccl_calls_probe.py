"""The torch.distributed calls bench.py makes on its N>1 path (RCCL group with device_id, a gloo side group for the
boundary columns, barrier, all_gather of the best triple, all_reduce MAX of the step time, send/recv on the gloo
group), run with however many ranks the box has GPUs for -- on the one-GPU test box that is a 1-rank group, which
still creates the RCCL communicator and runs every collective through it:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 tools/nccl_calls_probe.py"""
import os
import torch
import torch.distributed as dist

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
side = dist.new_group(backend="gloo")
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([rank + 5, 7, 22], dtype=torch.int64, device=dev)
out = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(out, t)
assert out[rank].tolist() == [rank + 5, 7, 22]
x = torch.tensor([1.5 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
assert float(x.item()) == 1.5 + world - 1
if world > 1:                       # one boundary segment around the ring of bands on the gloo side group
    buf = torch.full((32768, 2), rank, dtype=torch.int32)
    if rank + 1 < world:
        dist.send(buf, dst=rank + 1, group=side)
    if rank > 0:
        got = torch.empty_like(buf)
        dist.recv(got, src=rank - 1, group=side)
        assert int(got[0, 0]) == rank - 1
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("nccl calls ok, world", world)
