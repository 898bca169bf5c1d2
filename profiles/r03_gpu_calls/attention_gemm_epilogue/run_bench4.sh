cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/bench4; mkdir -p $O; rm -f $O/summary.txt
for m in 0 1 0 1; do
python scratch/bench_ab.py $m --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/b$m.log 2>&1
tail -1 $O/b$m.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('pingpong=$m ms/step', d['ms_per_step'], 'value', d['value'])" | tee -a $O/summary.txt
done
