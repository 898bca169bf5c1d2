#!/bin/bash
# round 6, call 11: evidence of the FINAL binary: the GPU suite, smoke(), the counter / trace passes of the default command (their traffic file first, so that the
# bench line's roofline.traffic comes from THIS binary), the default bench line, the driver-length line, the other variants (C2 base, small, large), the C5 lines
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -8 > $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash scripts/profile_round.sh r06 > $O/profile_round.log 2>&1
cp $O/r06_hbm_traffic.json profiles/r06_hbm_traffic.json
python scripts/mfma_util.py $O/r06_bench_default_kernel_stats.txt $O/r06_sq_counters.txt $O/r06_effective_clock.json > $O/r06_mfma_utilisation.txt 2>&1
python bench.py > $O/r06_bench_default.json 2> $O/err.log
python bench.py --steps 20 --warmup 3 > $O/r06_bench_steps20.json 2>> $O/err.log
for v in base small large; do
  pgb=256; [ $v = large ] && pgb=512
  python bench.py --variant $v --per-gpu-batch $pgb --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > $O/variant_$v.json
done
python scripts/transcribe_bench.py > $O/c5_random.log 2>&1
python scripts/transcribe_trained_bench.py 20 small > $O/c5_trained.log 2>&1
python scripts/mel_bench.py > $O/mel_bench.log 2>&1
tail -3 $O/suite.log; tail -2 $O/smoke.log; head -14 $O/r06_bench_default_kernel_stats.txt | cut -c1-200
python - <<PY
import json
for f in ("r06_bench_default", "r06_bench_steps20"):
    j=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f, j["ms_per_step"], j["value"], j["per_step_ms"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], r["frac"], r["main_stream_all"], r["traffic"], j["config"]["side_streams"], j["cpu_baseline"]["kind"] if "cpu_baseline" in j else None)
for v in ("base","small","large"):
    j=json.loads(open(f"$O/variant_{v}.json").read())
    print(v, j["ms_per_step"], j["value"], j["plain_step_ms"], j["span_bwd_ms"], j["step_frac_algorithmic"], j["step_frac_executed"], j["config"]["micro_batch"])
PY
tail -2 $O/c5_random.log; tail -3 $O/c5_trained.log; tail -1 $O/mel_bench.log
