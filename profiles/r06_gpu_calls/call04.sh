#!/bin/bash
# round 6, call 4: new GPU tests (attention scores / qk / word timestamps; load_model in_memory; decode fallback) + the whole GPU suite once
O=gpurun_out/r06d
mkdir -p $O
python -m pytest tests/test_gpu_timing.py -m gpu -x -q --timeout 900 2>&1 | tail -25 > $O/timing_tests.log
cat $O/timing_tests.log
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -12 > $O/suite.log
cat $O/suite.log
