set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c6; mkdir -p $O
for a in 0 1 2 4 8 15 3 7; do OASR_LOGMEL_ABL=$a python scripts/mel_bench.py 2>&1 | tail -1 | sed "s/^/ABL $a: /" | tee -a $O/summary.txt; done
