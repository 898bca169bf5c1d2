#!/bin/bash
# round 5, call 22: log-mel FFT kernel with 16 instead of 32 frames per workgroup (43 KiB of LDS: three workgroups per CU, 16 item groups): parity tests + timing A/B
mkdir -p gpurun_out/r05k
python -m pytest tests/test_gpu_ops.py tests/test_gpu_span.py tests/test_gpu_properties.py -x -q -m gpu -k "mel or log_mel or unfinalized" 2>&1 | grep -E "passed|failed|rror" | tail -3
export OASR_TESTING_HOOKS=1
for i in 1 2; do
  OASR_LOGMEL=fft32 python scripts/mel_bench.py 2>&1 | grep log_mel | sed 's/^/fft32 /'
  python scripts/mel_bench.py 2>&1 | grep log_mel | sed 's/^/fft16 /'
done
