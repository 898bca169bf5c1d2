import sys
sys.path.insert(0, "/root/repo")
from olmoasr_amd import _native as N
mode = int(sys.argv[1])
N.lib().oasr_attention_set_pingpong(mode)
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
