cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/bench3; mkdir -p $O; rm -f $O/summary.txt
for p in 0 1 2 0 2; do
  OASR_LIB=/root/repo/scratch/abl/liboasr_noslp.so OASR_ATTN_PIPE=$p python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b$p.log 2>&1
  tail -1 $O/b$p.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline'].get('other_kernels',{})
print('PIPE=$p ms/step', d['ms_per_step'], 'value', d['value'])" | tee -a $O/summary.txt
done
