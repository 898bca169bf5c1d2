#!/bin/bash
# round 6, call 12: after the fix of the one-launch decode guard (negative layer stride): decode tests, the engine probe, the two C5 lines
O=gpurun_out/r06k
mkdir -p $O
python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode_parity.py -m gpu -q --timeout 1500 2>&1 | tail -4
python scripts/decode_xcd_probe.py small 1 32 -1,1,2 2>&1 | tail -3 | cut -c1-200 | tee $O/decode_probe.txt
python scripts/decode_xcd_probe.py medium 1 32 -1,1,2 2>&1 | tail -3 | cut -c1-200 | tee -a $O/decode_probe.txt
python scripts/transcribe_bench.py 2>&1 | tail -1 | tee $O/c5_random.log
python scripts/transcribe_trained_bench.py 20 small 2>&1 | tail -1 | tee $O/c5_trained.log
python scripts/transcribe_trained_bench.py 20 small 2>&1 | tail -1 | tee -a $O/c5_trained.log
