import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
g = torch.ones(1000, device="cuda")
w = dist.all_reduce(g[100:600], async_op=True)
w.wait(); torch.cuda.synchronize()
print("nccl ok", g.sum().item(), dist.get_backend())
dist.barrier(); dist.destroy_process_group()
