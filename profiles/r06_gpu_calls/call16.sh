#!/bin/bash
# round 6, call 16: CU-masked side streams (OASR_SIDE_CUMASK=n, OASR_SIDE_CUMASK_PATTERN=p) against the unmasked lowest-priority side streams, whole step,
# same box, interleaved twice
O=gpurun_out/r06m
mkdir -p $O
: > $O/ab.txt
for rep in 1 2; do
  for cfg in "0 0" "64 0" "128 0" "192 0" "224 0" "128 1" "128 2" "192 1"; do
    set -- $cfg
    line=$(OASR_TESTING_HOOKS=1 OASR_SIDE_CUMASK=$1 OASR_SIDE_CUMASK_PATTERN=$2 python bench.py --steps 10 --warmup 2 --ab-steps 0 --no-cpu-baseline 2>>$O/err.log | tail -1)
    echo "$line" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('rep $rep cumask $1 pattern $2  ms_per_step %.2f  dominant %.4f  main_all %.4f' % (j['ms_per_step'], r['frac'], r['main_stream_all']['frac']))" >> $O/ab.txt
  done
done
cat $O/ab.txt; tail -5 $O/err.log
