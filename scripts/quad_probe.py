"""Times the quad GEMM kernel (force code 7) on two NT shapes -- run with OASR_QUAD_ABL=n for the ablation variants."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
from olmoasr_amd import ops  # noqa: E402

BF = torch.bfloat16
M, d = 96000, 1024
x = torch.randn(M, 4 * d, device="cuda").to(BF)
w = (torch.randn(4 * d, 4 * d, device="cuda") * 0.02).to(BF)
out = torch.empty(M, 4 * d, device="cuda", dtype=BF)
cases = [("N=3072 K=1024", lambda: ops.gemm(x[:, :d], w[:3 * d, :d], M, 3 * d, d, out=out[:, :3 * d]), 2.0 * M * 3 * d * d),
         ("N=1024 K=4096", lambda: ops.gemm(x, w[:d], M, d, 4 * d, out=out[:, :d]), 2.0 * M * 4 * d * d)]
code = int(sys.argv[1]) if len(sys.argv) > 1 else 7
N.lib().oasr_gemm_force_general(code)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for name, fn, flops in cases:
    fn()
    best = 1e9
    for _ in range(5):
        ev[0].record()
        for _ in range(5):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) / 5)
    print(f"code {code} ABL {os.environ.get('OASR_QUAD_ABL', '0')} {name}: {best:.3f} ms = {flops / best / 1e9:.0f} TF/s")
N.lib().oasr_gemm_force_general(0)
