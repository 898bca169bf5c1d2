cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/bench5; mkdir -p $O; rm -f $O/summary.txt
for t in epiold epinew epiold epinew; do
OASR_LIB=/root/repo/scratch/abl/liboasr_$t.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/b_$t.log 2>&1
tail -1 $O/b_$t.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$t ms/step', d['ms_per_step'], 'value', d['value'])" | tee -a $O/summary.txt
done
