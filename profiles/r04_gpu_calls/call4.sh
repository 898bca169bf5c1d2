#!/bin/bash
# round 4, GPU call 4: weight-gradient kernel choice at the shapes the span step made new (decoder ~18.7k token rows) and the
# cross-attention key|value gradient [2048 x 1024] over 192k tokens, never swept before
mkdir -p gpurun_out/r04
TOKENS=192000 SHAPES=2048x1024,3072x1024 SPLITS=4,8,16,24 python scripts/wgrad_sweep.py 5 > gpurun_out/r04/call4_wgrad_sweep.txt 2>&1
TOKENS=18688 SHAPES=1024x1024,3072x1024,4096x1024,1024x4096 SPLITS=1,2,4,8,16 python scripts/wgrad_sweep.py 5 >> gpurun_out/r04/call4_wgrad_sweep.txt 2>&1
cat gpurun_out/r04/call4_wgrad_sweep.txt
for v in 0 8 16; do
  OASR_WGRAD_PP32=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PP32=$v', j['ms_per_step'])"
done | tee gpurun_out/r04/call4_pp32_ab.txt
