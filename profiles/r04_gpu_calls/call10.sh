#!/bin/bash
# round 4, GPU call 10: attention forward with the row sums of P through the matrix pipe (ones-MFMA) instead of 32 VALU adds per tile
mkdir -p gpurun_out/r04
for i in 1 2; do
  python scripts/attn_bench.py 20 2>&1 | grep -v amdgpu.ids | sed 's/^/base /'
  OASR_LIB=$PWD/olmoasr_amd/liboasr_ones.so python scripts/attn_bench.py 20 2>&1 | grep -v amdgpu.ids | sed 's/^/ones /'
done | tee gpurun_out/r04/call10_attn_ones.txt
OASR_LIB=$PWD/olmoasr_amd/liboasr_ones.so python -m pytest tests/test_gpu_ops.py -q -k "attention" 2>&1 | tail -3 | tee -a gpurun_out/r04/call10_attn_ones.txt
for lib in liboasr.so liboasr_ones.so liboasr.so liboasr_ones.so; do
  OASR_LIB=$PWD/olmoasr_amd/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', j['ms_per_step'], j['final_loss'])"
done | tee -a gpurun_out/r04/call10_attn_ones.txt
