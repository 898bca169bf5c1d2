cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/bench6; mkdir -p $O; rm -f $O/summary.txt
for v in -1 40 -1 40; do
python scratch/bench_var.py $v --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/b_$v.log 2>&1
tail -1 $O/b_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('variant $v ms/step', d['ms_per_step'], 'value', d['value'])" | tee -a $O/summary.txt
done
