import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
r = dist.get_rank()
a = torch.full((4,), float(r), device="cuda")
b = torch.zeros(4, device="cuda")
ops = [dist.P2POp(dist.isend, a, 1 - r), dist.P2POp(dist.irecv, b, 1 - r)]
for w in dist.batch_isend_irecv(ops): w.wait()
torch.cuda.synchronize()
print("rank", r, "got", b.tolist())
dist.destroy_process_group()
