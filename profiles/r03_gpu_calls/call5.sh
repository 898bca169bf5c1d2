set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c5; mkdir -p $O
python scripts/quad_probe.py 4 2>&1 | grep "^code" | tee -a $O/summary.txt
for a in 0 1 2 3 4 7 8; do OASR_QUAD_ABL=$a python scripts/quad_probe.py 7 2>&1 | grep "^code" | tee -a $O/summary.txt; done
