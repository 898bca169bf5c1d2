#!/bin/bash
# round 6, call 13: log-mel quad kernel, cross-lane radix-4 as v_fmac_f32_dpp (signs folded into the twiddle rows): parity + A/B against the previous build
export OASR_TESTING_HOOKS=1
python -m pytest tests/test_gpu_ops.py tests/test_gpu_data.py -m gpu -q --timeout 900 -k "log_mel or data" 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
for rep in 1 2 3; do
  OASR_LIB=$PWD/scratch/abl/liboasr_before_fmac.so python scripts/mel_bench.py 2>&1 | tail -1 | sed 's/^/before /'
  python scripts/mel_bench.py 2>&1 | tail -1 | sed 's/^/fmac   /'
done
