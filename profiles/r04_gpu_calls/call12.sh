#!/bin/bash
# round 4, GPU call 12: attention forward with Q pre-multiplied by log2(e)/8 and the score accumulators seeded with -m_ref (one VALU less per score)
mkdir -p gpurun_out/r04
for i in 1 2; do
  python scripts/attn_bench.py 20 2>&1 | grep -v amdgpu.ids | sed 's/^/base   /'
  OASR_LIB=$PWD/olmoasr_amd/liboasr_qscale.so python scripts/attn_bench.py 20 2>&1 | grep -v amdgpu.ids | sed 's/^/qscale /'
done | tee gpurun_out/r04/call12_attn_qscale.txt
OASR_LIB=$PWD/olmoasr_amd/liboasr_qscale.so python -m pytest tests/test_gpu_ops.py -q -k "attention" 2>&1 | tail -4 | tee -a gpurun_out/r04/call12_attn_qscale.txt
for lib in liboasr.so liboasr_qscale.so liboasr.so liboasr_qscale.so; do
  OASR_LIB=$PWD/olmoasr_amd/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', j['ms_per_step'], j['final_loss'])"
done | tee -a gpurun_out/r04/call12_attn_qscale.txt
