cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/final4; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -1 $O/bench_stdout.txt > $O/r03_bench_default.json
tail -1 $O/bench_stdout.txt | cut -c1-400
