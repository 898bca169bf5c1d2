import sys
sys.path.insert(0, "/root/repo")
from olmoasr_amd import _native as N
v = int(sys.argv[1])
N.lib().oasr_gemm_set_variant(v)
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
