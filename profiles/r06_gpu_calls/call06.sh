#!/bin/bash
# round 6, call 6: quad-lane log-mel kernel variants (kernel alone and with finalize, 128 clips per launch): old LDS FFT, quad (fp32 staging, 3 waves/SIMD),
# int16 staging with launch bounds for 4 / 3 waves per SIMD; parity tests under the int16-staging build
O=gpurun_out/r06f
mkdir -p $O
export OASR_TESTING_HOOKS=1
for rep in 1 2; do
  OASR_LOGMEL=fft python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/mel_bench.txt
  python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/mel_bench.txt
  for v in s16w4 s16w3 w4; do
    OASR_LIB=$PWD/scratch/abl/liboasr_logmel_$v.so python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/mel_bench.txt
  done
done
OASR_LIB=$PWD/scratch/abl/liboasr_logmel_s16w4.so python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 900 -k "log_mel" 2>&1 | tail -4
python -m pytest tests/test_gpu_ops.py tests/test_gpu_data.py -m gpu -q --timeout 900 -k "log_mel or data" 2>&1 | tail -4
