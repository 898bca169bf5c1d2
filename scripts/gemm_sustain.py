import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from olmoasr_amd import ops
M, N, K = 192000, 4096, 4096
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5): ops.gemm(x, w, M, N, K, out=out)
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < 5:
    for _ in range(20): ops.gemm(x, w, M, N, K, out=out)
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
print(os.environ.get("OASR_LIB", "default lib"), f"{2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s-equivalent sustained", flush=True)
