#!/bin/bash
# round 6, call 17: first light of the chip-wide one-launch decoder step (csrc/decode_wide.hip, mode 5): probe against the multi-launch step and the one-XCD team,
# in-kernel stamps, the step-engine tests
O=gpurun_out/r06w
mkdir -p $O
export OASR_TESTING_HOOKS=1
timeout 300 python scripts/decode_xcd_probe.py medium 1 32 1,2,5 > $O/probe_medium.txt 2>&1
OASR_XCD_FLAGS=256 timeout 300 python scripts/decode_xcd_probe.py medium 1 32 5 > $O/probe_medium_stamps.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py small 1 32 1,2,5 > $O/probe_small.txt 2>&1
timeout 300 python scripts/decode_xcd_probe.py medium 1 300 1,5 > $O/probe_medium_300.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_decode_step.py -x -q --timeout 600 2>&1 | tail -15 > $O/tests.txt
tail -4 $O/probe_medium.txt; tail -12 $O/probe_medium_stamps.txt; tail -3 $O/probe_small.txt; tail -2 $O/probe_medium_300.txt; cat $O/tests.txt
