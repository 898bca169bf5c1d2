#!/bin/bash
# round 6, call 29: the chip-wide decoder step as the B = 1 default: decode tests (tiny / large added), engine probe at the default, the two C5 lines
# (small, twice; medium once), bench hbm_kernels line
O=gpurun_out/r06w13
mkdir -p $O
python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode_parity.py -m gpu -q --timeout 1500 2>&1 | tail -4 | tee $O/tests.txt
for v in tiny base small medium large; do OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py $v 1 32 -1,1,2 2>&1 | grep -v amdgpu | cut -c1-230 | tee -a $O/decode_probe.txt; done
OASR_TESTING_HOOKS=1 python scripts/decode_xcd_probe.py medium 1 300 -1,2 2>&1 | grep -v amdgpu | cut -c1-230 | tee -a $O/decode_probe.txt
python scripts/transcribe_bench.py 2>&1 | tail -1 | tee $O/c5_random.log
python scripts/transcribe_trained_bench.py 20 small 2>&1 | tail -1 | tee $O/c5_trained.log
python scripts/transcribe_trained_bench.py 20 small 2>&1 | tail -1 | tee -a $O/c5_trained.log
python scripts/transcribe_trained_bench.py 20 medium 2>&1 | tail -1 | tee -a $O/c5_trained.log
