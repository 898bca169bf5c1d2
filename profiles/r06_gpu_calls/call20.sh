#!/bin/bash
# round 6, call 20: chip-wide decoder step: DPP wave reductions, contiguous unit runs (no division loops); 576 vs 512 threads (flag 64)
O=gpurun_out/r06w4
mkdir -p $O
export OASR_TESTING_HOOKS=1
for f in 256 320; do
  echo "=== OASR_XCD_FLAGS=$f" >> $O/probe.txt
  OASR_XCD_FLAGS=$f timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 >> $O/probe.txt 2>&1
done
timeout 300 python scripts/decode_xcd_probe.py medium 1 300 1,5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py small 1 32 2,5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py large 1 32 2,5 >> $O/probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_decode_step.py -x -q --timeout 600 2>&1 | tail -5 > $O/tests.txt
grep -v amdgpu.ids $O/probe.txt; cat $O/tests.txt
