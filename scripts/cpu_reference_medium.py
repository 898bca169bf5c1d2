"""bench.py's `cpu_baseline` leg run where the reference tree is mounted (the build container): the UNMODIFIED reference module
(olmoasr.model.OLMoASR through the lines of scripts/training/train_timestamps.py:1440-1454, 1509-1512), OLMoASR-medium, one 30 s clip, full
train step, fp32 and CPU autocast(bfloat16) -> profiles/rNN_cpu_reference_medium.json.  On the GPU box the same leg times the oracle restatement
(kind "port"): this file is the "reference" number beside it, on this container's cores."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("oasr_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(bench)
sys.argv = argv
from oracle import model_oracle as mo  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "medium"
sd = mo.init_state_dict(mo.VARIANTS[variant], seed=0)
blk, loss, _ = bench.cpu_baseline(variant, sd, budget_s=120.0)
blk["first_loss"] = loss
blk["host"] = {"container_cores": len(os.sched_getaffinity(0)), "note": "build container (8 CPUs), not the GPU box's host"}
print(json.dumps(blk))
