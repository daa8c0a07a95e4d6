import os, torch, torch.distributed as dist
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=r, world_size=w, device_id=torch.device("cuda:0"))
    send = torch.full((6,), float(r + 1), device="cuda:0"); recv = torch.zeros(6 * w, device="cuda:0")
    dist.all_gather_into_tensor(recv, send); torch.cuda.synchronize(); print(r, "nccl same-device ok", recv.tolist())
    dist.destroy_process_group()
except Exception as e:
    print(r, "nccl same-device FAILED", type(e).__name__, str(e)[:300])
