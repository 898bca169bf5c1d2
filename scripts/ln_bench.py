"""LayerNorm forward / backward bandwidth on the encoder stream shape (96000 x 1024)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

rows, d = 96000, 1024
x = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
dy = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
dres = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
g, b = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
y, mean, rstd = ops.layernorm_fwd(x, g, b)


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


tf = t(lambda: ops.layernorm_fwd(x, g, b))
tb = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres))
u = rows * d * 2
print(f"LayerNorm {rows}x{d}: fwd {tf * 1e3:.1f} us = {2 * u / tf / 1e9:.2f} TB/s | bwd {tb * 1e3:.1f} us = {4 * u / tb / 1e9:.2f} TB/s")
