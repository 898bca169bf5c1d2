"""Writes tests/golden/ref_param_order.json: ``named_parameters()`` order of the UNMODIFIED reference model -- the index
space of ``torch.optim.AdamW.state_dict()['state']`` in the reference's checkpoints (train_timestamps.py:949), which the
native model's ``optimizer_state_dict`` must reproduce.  Run in the build container (needs /root/reference)."""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_oracle as mo  # noqa: E402
from oracle import ref_import  # noqa: E402

ref_model, _, _ = ref_import.load()
out = {}
for v in ("tiny",):
    m = ref_model.OLMoASR(dims=types.SimpleNamespace(**mo.VARIANTS[v].__dict__))
    out[v] = [n for n, _ in m.named_parameters()]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_param_order.json")
json.dump(out, open(path, "w"), indent=0)
print(path, len(out["tiny"]))
