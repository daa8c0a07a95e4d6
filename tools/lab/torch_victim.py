"""lab: is a plain torch matmul / elementwise chain (library kernels, nothing of ours) reproducible call to call?"""
import sys, torch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(8192, 4096, device="cuda", generator=g); b = torch.randn(4096, 4096, device="cuda", generator=g)
x = torch.randn(64_000_000, device="cuda", generator=g)
def el(x):
    return torch.sigmoid(x) * torch.exp(-x * x) + torch.reciprocal(1.0 + x * x) + torch.sin(x)
ref_mm = a @ b; ref_el = el(x)
torch.cuda.synchronize()
bad_mm = bad_el = 0
for r in range(reps):
    c = a @ b; y = el(x)
    torch.cuda.synchronize()
    bad_mm += int(not torch.equal(c, ref_mm)); bad_el += int(not torch.equal(y, ref_el))
print(f"torch victim: matmul {bad_mm} of {reps} differ, elementwise {bad_el} of {reps} differ", flush=True)
