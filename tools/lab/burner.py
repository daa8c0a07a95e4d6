"""lab: keeps the GPU busy from a second process (fp32 matmuls through the BLAS library) for N seconds"""
import sys, time, torch
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else 60
a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
n = 0
while time.time() < t_end:
    for _ in range(20):
        c = a @ b
    torch.cuda.synchronize(); n += 20
print("burner matmuls", n)
