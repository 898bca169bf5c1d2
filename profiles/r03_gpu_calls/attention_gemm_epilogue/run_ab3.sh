cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" 2>&1 | tail -2
for rep in 1 2; do for tag in epinew idle; do echo "== $tag"; OASR_LIB=/root/repo/scratch/abl/liboasr_$tag.so python scripts/attn_bench.py 30 2>&1 | grep -v amdgpu | head -2; done; done
