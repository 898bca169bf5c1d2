#!/bin/bash
# round 6, call 30: chip-wide decoder step: the helper wave polls with no load in flight (LayerNorm parameters and bias staged by the compute waves), hand-over buffers by projection parity
O=gpurun_out/r06w14
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
timeout 900 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode_parity.py -m gpu -q --timeout 800 2>&1 | tail -3
