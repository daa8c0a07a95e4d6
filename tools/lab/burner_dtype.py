"""lab: keeps the GPU busy from a second process with LIBRARY matmuls of a given dtype (bf16 / fp16 / fp32) for N seconds"""
import sys, time, torch
secs = float(sys.argv[1]); dt = dict(bf16=torch.bfloat16, fp16=torch.float16, fp32=torch.float32)[sys.argv[2]]
a = torch.randn(8192, 8192, device="cuda").to(dt); b = torch.randn(8192, 8192, device="cuda").to(dt)
t_end = time.time() + secs; n = 0
while time.time() < t_end:
    for _ in range(20):
        c = a @ b
    torch.cuda.synchronize(); n += 20
print("burner", sys.argv[2], "matmuls", n, flush=True)
