#!/bin/bash
# round 6, call 24: which of the two changes of call 23 broke parity: fp32 FMA products (default build) vs v_dot2c (flag 64 build, 512 threads)
O=gpurun_out/r06w8
mkdir -p $O
export OASR_TESTING_HOOKS=1
timeout 300 python scripts/decode_xcd_probe.py medium 1 32 2,5 > $O/probe.txt 2>&1
OASR_XCD_FLAGS=64 timeout 300 python scripts/decode_xcd_probe.py medium 1 32 2,5 >> $O/probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_decode_step.py -x -q --timeout 600 2>&1 | tail -5 > $O/tests.txt
grep -v amdgpu.ids $O/probe.txt; cat $O/tests.txt
