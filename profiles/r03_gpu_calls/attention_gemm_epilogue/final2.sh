cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/final2; mkdir -p $O; rm -f $O/*.txt
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
for cfg in "base 256 128" "small 256 128" "large 512 64"; do set -- $cfg
python bench.py --variant $1 --per-gpu-batch $2 --micro-batch $3 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'step_model_tflops_per_gpu': r.get('step_model_tflops_per_gpu'), 'step_frac_of_mfma_peak': r.get('step_frac_of_mfma_peak'), 'workload': d['config']['workload']}))" | tee -a $O/variants.txt
done
python scripts/transcribe_bench.py small 600 20 2>/dev/null | tail -1 | tee -a $O/c5.txt
python scripts/transcribe_bench.py small 600 1 ts 2>/dev/null | tail -1 | tee -a $O/c5.txt
