cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for tag in cur brk; do echo "== $tag"; OASR_LIB=/root/repo/scratch/abl/liboasr_$tag.so python scripts/attn_bench.py 30 2>&1 | grep -v amdgpu; done; done
