#!/bin/bash
# round 6, call 43: chip-wide decoder step: 8 / 10 / 12 compute waves per workgroup (576 / 704 / 832 threads; rebuilt on the box per setting)
O=gpurun_out/r06wt
mkdir -p $O
export OASR_TESTING_HOOKS=1
cd olmoasr_amd/csrc
for wt in 576 704 832 576; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -munsafe-fp-atomics -DDW_WT=$wt -c decode_wide.hip -o build/decode_wide.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liboasr.so build/*.o
  for v in medium small large; do echo "WT=$wt $(cd ../..; python scripts/decode_xcd_probe.py $v 1 32 -1 2>&1 | tail -1 | cut -c1-100)" | tee -a ../../$O/wt.txt; done
done
cd ../..
python -m pytest tests/test_gpu_decode_step.py -q --timeout 600 2>&1 | tail -2
