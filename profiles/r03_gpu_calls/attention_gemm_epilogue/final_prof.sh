cd /root/repo; export TMPDIR=/tmp
bash scripts/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
python scripts/hostinfo.py > gpurun_out/r03/r03_hostinfo.txt 2>&1
python bench.py > gpurun_out/r03/bench_stdout.txt 2> gpurun_out/r03/bench_stderr.txt
tail -1 gpurun_out/r03/bench_stdout.txt > gpurun_out/r03/r03_bench_default.json
tail -1 gpurun_out/r03/bench_stdout.txt | cut -c1-600
ls gpurun_out/r03
