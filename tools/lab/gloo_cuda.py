import os, torch, torch.distributed as dist
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=r, world_size=w)
dev = "cuda:0"
send = torch.full((6,), float(r + 1), device=dev)
recv = torch.zeros(6 * w, device=dev)
try:
    dist.all_gather_into_tensor(recv, send)
    torch.cuda.synchronize()
    print(r, "all_gather_into_tensor ok", recv.tolist())
except Exception as e:
    print(r, "all_gather_into_tensor FAILED", type(e).__name__, str(e)[:200])
t = torch.tensor([float(r)], device=dev, dtype=torch.float64)
try:
    dist.all_reduce(t, op=dist.ReduceOp.MAX); print(r, "all_reduce ok", t.item())
except Exception as e:
    print(r, "all_reduce FAILED", type(e).__name__, str(e)[:200])
try:
    dist.barrier(); print(r, "barrier ok")
except Exception as e:
    print(r, "barrier FAILED", str(e)[:200])
dist.destroy_process_group()
