#!/bin/bash
# round 6, call 7: quad log-mel kernel with staging loads issued up front + LDS-transposed row stores: default build and the int16-staging build; parity
O=gpurun_out/r06g
mkdir -p $O
export OASR_TESTING_HOOKS=1
for rep in 1 2; do
  OASR_LOGMEL=fft python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/mel_bench.txt
  python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/mel_bench.txt
  OASR_LIB=$PWD/scratch/abl/liboasr_logmel_s16w4.so python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/mel_bench.txt
done
OASR_LIB=$PWD/scratch/abl/liboasr_logmel_s16w4.so python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 900 -k "log_mel" 2>&1 | tail -4
python -m pytest tests/test_gpu_ops.py tests/test_gpu_data.py -m gpu -q --timeout 900 -k "log_mel or data" 2>&1 | tail -4
