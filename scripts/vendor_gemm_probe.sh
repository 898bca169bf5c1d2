export TMPDIR=/tmp
python - <<'PY' &
import time, torch
M, N, K = 192000, 4096, 4096
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
t0 = time.time(); n = 0
while time.time() - t0 < 12:
    for _ in range(20):
        torch.matmul(x, w.t(), out=out)
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
print(f"torch.matmul loop: {n} launches in {dt:.1f} s = {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s sustained", flush=True)
PY
PID=$!
sleep 5
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -i -E "Power \(W\)|sclk" | head -4; sleep 1; done
wait $PID
rm -rf /tmp/vk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vk -o vk -- python -c "
import torch
x = torch.randn(192000, 4096, device='cuda').bfloat16(); w = (torch.randn(4096, 4096, device='cuda') * 0.02).bfloat16(); out = torch.empty(192000, 4096, device='cuda', dtype=torch.bfloat16)
for _ in range(3): torch.matmul(x, w.t(), out=out)
torch.cuda.synchronize()
" > /dev/null 2>&1
f=$(find /tmp/vk -name "*kernel_trace.csv" | head -1); python scripts/rocprof_summary.py "$f" | head -5 | cut -c1-200
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "Cijk" in r["Kernel_Name"] or "gemm" in r["Kernel_Name"].lower():
        print({k: r[k] for k in ("Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in r}); break
PY
