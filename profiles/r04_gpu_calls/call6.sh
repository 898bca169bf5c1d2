#!/bin/bash
# round 4, GPU call 6: cross-attention backward kernel choice under the span step (few active query tiles per sample)
mkdir -p gpurun_out/r04
for v in 3 2 1 0 3; do
  OASR_ATTN_ROWS_PP=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ROWS_PP=$v', j['ms_per_step'], j['final_loss'])"
done | tee gpurun_out/r04/call6_rows_pp_ab.txt
