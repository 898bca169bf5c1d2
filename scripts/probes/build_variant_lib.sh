#!/bin/bash
# usage: build_abl.sh <src.hip> <tag> [extra -D flags]   -> scratch/abl/liboasr_<tag>.so
set -e
SRC=$1; TAG=$2; shift 2
cd /root/repo/olmoasr_amd/csrc
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics -I. "$@" -c $SRC -o /tmp/attn_$TAG.o
OBJS=$(ls build/*.o | grep -v attention.o | grep -v decode_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $(dirname $0)/../../scratch/abl/liboasr_$TAG.so $OBJS /tmp/attn_$TAG.o
echo built $TAG
