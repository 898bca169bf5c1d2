#!/bin/bash
# Board power / shader clock while the layer GEMMs run back to back (evidence for the power-budget ceiling in DESIGN.md 3b).
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r02d}
mkdir -p $OUT
{
  echo "### idle"
  rocm-smi --showpower --showmaxpower --showclocks --showperflevel 2>&1 | grep -v "^$" | head -40
  python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from olmoasr_amd import ops
M, N, K = 192000, 4096, 4096
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
t0 = time.time(); n = 0
while time.time() - t0 < 14:
    for _ in range(20):
        ops.gemm(x, w, M, N, K, out=out)
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
print(f"GEMM loop: {n} launches in {dt:.1f} s = {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s sustained", flush=True)
PY
  PID=$!
  sleep 5
  for i in 1 2 3 4; do
    echo "### under load, sample $i"
    rocm-smi --showpower --showclocks 2>&1 | grep -i -E "power|sclk|mclk|fclk" | head -12
    sleep 1
  done
  wait $PID
} > $OUT/gpu_power_probe.txt 2>&1
cat $OUT/gpu_power_probe.txt
