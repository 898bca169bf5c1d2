#!/bin/bash
# round 4, GPU call 23: phase cycle stamps inside the attention forward (scratch build), product library beside it for the wall time
mkdir -p gpurun_out/r04
( python scripts/probes/attn_fwd_stamps.py; OASR_LIB=$PWD/scratch/abl/liboasr_stamps.so python scripts/probes/attn_fwd_stamps.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/call23_attn_fwd_stamps.txt
