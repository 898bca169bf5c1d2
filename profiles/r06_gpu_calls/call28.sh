#!/bin/bash
# round 6, call 28: chip-wide decoder step: two-head merge passes, leaner LayerNorm; probe + the WHOLE GPU suite on the new default engine
O=gpurun_out/r06w12
mkdir -p $O
export OASR_TESTING_HOOKS=1
timeout 300 python scripts/decode_xcd_probe.py medium 1 32 2,5 > $O/probe.txt 2>&1
OASR_XCD_FLAGS=256 timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py medium 1 300 5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py small 1 32 5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py large 1 32 5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py base 1 32 2,5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py tiny 1 32 1,5 >> $O/probe.txt 2>&1
grep -v amdgpu.ids $O/probe.txt
timeout 1700 python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -12 > $O/suite.txt
cat $O/suite.txt
