#!/bin/bash
# round 6, call 27: chip-wide decoder step: old self-attention rows before the poll, one wave reduction per (wave, row)
O=gpurun_out/r06w11
mkdir -p $O
export OASR_TESTING_HOOKS=1
timeout 300 python scripts/decode_xcd_probe.py medium 1 32 2,5 > $O/probe.txt 2>&1
OASR_XCD_FLAGS=256 timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 >> $O/probe.txt 2>&1
OASR_XCD_FLAGS=$((256 + 100 * 512)) timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py medium 1 300 5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py small 1 32 5 >> $O/probe.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py large 1 32 5 >> $O/probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_decode_step.py -x -q --timeout 600 2>&1 | tail -5 > $O/tests.txt
grep -v amdgpu.ids $O/probe.txt; cat $O/tests.txt
