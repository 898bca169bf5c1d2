"""Times the fused cross-entropy kernel (fwd + in-place bwd) at the benchmarked shape: [57344, 51968] bf16 logits, all rows valid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

rows, V, Vp = 57344, 51865, 51968
lg = torch.randn(rows, Vp, device="cuda", dtype=torch.bfloat16)
for frac_valid in (1.0, 0.25):
    tgt = torch.randint(0, 50000, (rows,), device="cuda")
    if frac_valid < 1.0:
        tgt[torch.rand(rows, device="cuda") > frac_valid] = 51864
    ops.cross_entropy_(lg, V, tgt, 51864)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        lg.normal_()
        torch.cuda.synchronize()
        e0.record()
        ops.cross_entropy_(lg, V, tgt, 51864)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    nv = int((tgt != 51864).sum())
    actual = nv * 4 * Vp + (rows - nv) * 2 * Vp
    print(f"valid rows {frac_valid:.2f}: {best:.3f} ms = {rows * 4 * Vp / best / 1e9:.2f} TB/s algorithmic (4 V' B/row), {actual / best / 1e9:.2f} TB/s of bytes actually moved")
