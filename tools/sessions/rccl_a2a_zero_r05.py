"""RCCL accepts an all_to_all_single whose split sizes are all zero (the contribution of a rank that shares nothing at a level)?  One rank."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29542")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
e = torch.empty(0, dtype=torch.float64, device="cuda")
w = dist.all_to_all_single(e, e.clone(), output_split_sizes=[0], input_split_sizes=[0], async_op=True); w.wait()
torch.cuda.synchronize()
print("all-zero splits over RCCL: ok")
dist.destroy_process_group()
