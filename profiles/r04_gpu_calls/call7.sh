#!/bin/bash
# round 4, GPU call 7: grid quantisation -- ping-pong (code4) vs 256x128 (code2) at the row counts the span step's decoder backward runs on
mkdir -p gpurun_out/r04
MS=18688,24064,16000 python scripts/gemm_ab.py 5 2 4 > gpurun_out/r04/call7_gemm_quant.txt 2>&1
cat gpurun_out/r04/call7_gemm_quant.txt
for v in 1 0 1 0; do
  OASR_GEMM_QUANT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QUANT=$v', j['ms_per_step'], j['final_loss'])"
done | tee gpurun_out/r04/call7_quant_ab.txt
