#!/bin/bash
# round 4, GPU call 16: GELU + GELU' in the GEMM epilogue on packed fp32 (v_pk_fma/mul/add_f32) vs scalar: bit identity (crc) and time, interleaved
mkdir -p gpurun_out/r04
for i in 1 2 3; do
  for lib in liboasr_nopk.so liboasr.so; do
    OASR_LIB=$PWD/olmoasr_amd/$lib python scripts/gelu_epilogue_ab.py 192000 2>&1 | sed "s/^/$lib /"
  done
done | tee gpurun_out/r04/call16_gelu_pk.txt
