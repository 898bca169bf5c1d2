#!/bin/bash
# round 6, call 19: chip-wide decoder step with a helper wave (wave 0 requests no weights): 576 threads (8 compute waves) vs 512 (7; flag 64), stamps
O=gpurun_out/r06w3
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
