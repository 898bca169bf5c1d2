"""A/B probe: does a power-of-two leading dimension (row stride 2/8 KiB) hurt the direct-to-LDS GEMM through L2 channel
camping?  Same shapes, operands stored with ld = K vs ld = K + pad."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


M = 48000
for K, N in ((1024, 1024), (4096, 1024), (1024, 4096)):
    for pad_a, pad_b, pad_c in ((0, 0, 0), (64, 0, 0), (64, 64, 0), (64, 64, 64), (8, 8, 8), (32, 32, 32), (128, 128, 128)):
        xa = torch.randn(M, K + pad_a, device=DEV).to(BF)[:, :K]
        w = (torch.randn(N, K + pad_b, device=DEV) * 0.02).to(BF)[:, :K]
        out = torch.empty(M, N + pad_c, device=DEV, dtype=BF)[:, :N]
        ms = timeit(lambda: ops.gemm(xa, w, M, N, K, out=out))
        print(f"NT M={M} N={N} K={K} pad A/B/C = {pad_a}/{pad_b}/{pad_c}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
