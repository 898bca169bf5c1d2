#!/bin/bash
# round 6, call 5: the quad-lane register log-mel kernel (csrc/logmel_quad.h): parity tests, then scripts/mel_bench.py old (OASR_LOGMEL=fft) vs new, slp on / off
O=gpurun_out/r06e
mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 900 -k "log_mel" 2>&1 | tail -15 > $O/logmel_tests.log
cat $O/logmel_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
export OASR_TESTING_HOOKS=1
for rep in 1 2; do
  OASR_LOGMEL=fft python scripts/mel_bench.py 2>&1 | tail -1 | sed 's/^/old-lds-fft  /' | tee -a $O/mel_bench.txt
  python scripts/mel_bench.py 2>&1 | tail -1 | sed 's/^/quad         /' | tee -a $O/mel_bench.txt
  OASR_LIB=$PWD/scratch/abl/liboasr_logmel_noslp.so python scripts/mel_bench.py 2>&1 | tail -1 | sed 's/^/quad-noslp   /' | tee -a $O/mel_bench.txt
done
python -m pytest tests/test_gpu_timing.py tests/test_gpu_modules.py tests/test_gpu_model.py -m gpu -q --timeout 900 2>&1 | tail -6
