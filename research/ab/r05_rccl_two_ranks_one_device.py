"""EXPERIMENT: does RCCL accept two ranks on ONE device (the only kind of peer a one-GPU box can offer)?  torchrun --nproc-per-node 2."""
import os
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
t = torch.tensor([rank + 1], dtype=torch.int32, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("rank", rank, "all_reduce MAX ->", int(t.item()), flush=True)
dist.destroy_process_group()
