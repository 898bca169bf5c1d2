#!/bin/bash
# round 6, call 33: chip-wide decoder step with the final LayerNorm + logits projection as its last phase (one launch per token behind the embedding)
O=gpurun_out/r06w15
mkdir -p $O
export OASR_TESTING_HOOKS=1
for v in tiny base small medium large; do timeout 300 python scripts/decode_xcd_probe.py $v 1 32 1,-1 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt; done
timeout 300 python scripts/decode_xcd_probe.py medium 1 300 1,-1 2>&1 | grep -v amdgpu | cut -c1-250 >> $O/probe.txt
cat $O/probe.txt
timeout 900 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_decode_parity.py -m gpu -q --timeout 800 2>&1 | tail -3
