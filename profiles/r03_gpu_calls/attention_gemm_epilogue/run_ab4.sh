cd /root/repo; export TMPDIR=/tmp
for t in epinew flags epinew flags; do
OASR_LIB=/root/repo/scratch/abl/liboasr_$t.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$t ms/step', d['ms_per_step'], 'value', d['value'])"
done
