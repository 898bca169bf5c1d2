#!/bin/bash
# round 6, call 42: same-box A/B of the LayerNorm wave sums: the final binary (DPP path) against the binary one commit before (ds_bpermute butterfly),
# interleaved twice; whole step + the two LayerNorm lines of hbm_kernels
O=gpurun_out/r06lnab
mkdir -p $O
cp olmoasr_amd/liboasr.so /tmp/new.so
run() {
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --ab-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; h=r['hbm_kernels']
print('$1', j['ms_per_step'], r['frac'], 'ln_fwd', h['ln_fwd_kernel']['frac_of_8TBps'], h['ln_fwd_kernel']['us'], 'ln_bwd', h['ln_bwd_kernel']['frac_of_8TBps'], h['ln_bwd_kernel']['us'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  cp /tmp/new.so olmoasr_amd/liboasr.so; run dpp_$rep
  cp olmoasr_amd/liboasr_prev.bin olmoasr_amd/liboasr.so; run bpermute_$rep
done
cp /tmp/new.so olmoasr_amd/liboasr.so
