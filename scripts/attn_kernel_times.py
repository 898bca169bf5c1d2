"""Per-kernel times of the attention kernels on one problem (HIP events around single launches are not possible from Python: the
backward is one ABI call that launches two kernels), so this script is run under `rocprofv3 --kernel-trace --stats`.
usage: python scripts/attn_kernel_times.py [enc|cross|dec] [B] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "enc"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H, d = 16, 1024
Tq, Tk, causal = {"enc": (1500, 1500, False), "cross": (448, 1500, False), "dec": (448, 448, True)}[which]
BF = torch.bfloat16
torch.manual_seed(0)
if Tq == Tk:
    qkv = torch.randn(B, Tq, 3 * d, device="cuda").to(BF)
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(3))
else:
    qb = torch.randn(B, Tq, d, device="cuda").to(BF)
    kvb = torch.randn(B, Tk, 2 * d, device="cuda").to(BF)
    q = qb.unflatten(2, (H, 64))
    k, v = (kvb[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(2))
kv_len = torch.randint(8, 221, (B,), device="cuda", dtype=torch.int32) if causal else None
d_o = torch.randn(B, Tq, d, device="cuda").to(BF)
o, lse, o_lo = ops.attention_fwd(q, k, v, kv_len, causal, want_o_lo=True)
for _ in range(iters):
    ops.attention_fwd(q, k, v, kv_len, causal, want_o_lo=True)
    ops.attention_bwd(q, k, v, o, lse, d_o, kv_len, causal, o_lo=o_lo)
torch.cuda.synchronize()
