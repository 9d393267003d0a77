import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); out = [torch.empty_like(t)]
dist.all_gather(out, t); dist.barrier(); torch.cuda.synchronize()
x = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(x, op=dist.ReduceOp.MAX)
print("nccl single-rank ok", out[0].sum().item(), x.item())
dist.destroy_process_group()
