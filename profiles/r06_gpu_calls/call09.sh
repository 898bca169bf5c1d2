#!/bin/bash
# round 6, call 9: (a) quad log-mel kernel with bit-selects instead of VCC cndmasks: parity + timing; (b) the GELU epilogues' select as v_bfi:
# layer GEMMs old library vs new (scripts/gemm_ab.py, M = 192000) and the whole step old vs new, interleaved on one box
O=gpurun_out/r06i
mkdir -p $O
export OASR_TESTING_HOOKS=1
python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 900 -k "log_mel or gemm_gelu or gelu or epilogue" 2>&1 | tail -4
for rep in 1 2; do python scripts/mel_bench.py 2>&1 | tail -1 | tee -a $O/mel_bench.txt; done
OLD=$PWD/scratch/abl/liboasr_oldgelu.so
for rep in 1 2; do
  MS=192000 OASR_LIB=$OLD python scripts/gemm_ab.py 5 0 2>&1 | grep -E "mlp1|dgelu|qkv" | sed 's/^/old  /' | tee -a $O/gemm_ab.txt
  MS=192000 python scripts/gemm_ab.py 5 0 2>&1 | grep -E "mlp1|dgelu|qkv" | sed 's/^/new  /' | tee -a $O/gemm_ab.txt
done
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export OASR_LIB=$OLD; else unset OASR_LIB; fi
    python bench.py --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/bench_$lib.json
    python - <<PY | tee -a $O/step_ab.txt
import json
j=json.loads(open("$O/bench_$lib.json").read())
r=j["roofline"]
print("$lib", "ms/step", j["ms_per_step"], j["per_step_ms"], "dominant frac", r["frac"], "gemm_ms", r["gemm_ms_per_step"], "logmel", [ (k[:30], v.get("frac_of_8TBps")) for k,v in r.get("hbm_kernels",{}).items() if "logmel" in k][:1])
PY
  done
done
unset OASR_LIB
python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_sizes.py -m gpu -q --timeout 1500 2>&1 | tail -4
