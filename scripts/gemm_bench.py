"""Micro-benchmark of the bf16 MFMA GEMM (HIP events, random data) over the shapes of the OLMoASR-medium step.
Usage: python scripts/gemm_bench.py [geom ...]   (geom: 0 auto, 2 force 256x128, 3 force 256x256, 1 general kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
from olmoasr_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


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


def main():
    geoms = [int(a) for a in sys.argv[1:]] or [0, 2, 4]
    M, d = 48000, 1024
    x = torch.randn(M, 4 * d, device=DEV).to(BF)
    w = (torch.randn(4 * d, 4 * d, device=DEV) * 0.02).to(BF)
    bias = torch.randn(4 * d, device=DEV)
    out = torch.empty(M, 4 * d, device=DEV, dtype=BF)
    pre = torch.empty(M, 4 * d, device=DEV, dtype=BF)
    resid = torch.randn(M, d, device=DEV).to(BF)
    g32 = torch.zeros(4 * d, 4 * d, device=DEV)
    cases = []
    for K in (1024, 2048, 4096):
        cases.append((f"NT plain      M={M} N=1024 K={K}", lambda K=K: ops.gemm(x[:, :K], w[:1024, :K], M, 1024, K, out=out[:, :1024], a_view=None), 2.0 * M * 1024 * K))
    cases += [
        ("NT qkv bias   N=3072 K=1024", lambda: ops.gemm(x[:, :d], w[:3 * d, :d], M, 3 * d, d, bias=bias[:3 * d], out=out[:, :3 * d]), 2.0 * M * 3 * d * d),
        ("NT mlp1 gelu  N=4096 K=1024", lambda: ops.gemm(x[:, :d], w[:, :d], M, 4 * d, d, bias=bias, act=1, out=out, out_pre=pre), 2.0 * M * 4 * d * d),
        ("NT mlp2 resid N=1024 K=4096", lambda: ops.gemm(x, w[:d], M, d, 4 * d, bias=bias[:d], resid=resid, out=out[:, :d]), 2.0 * M * 4 * d * d),
        ("NN dgrad      N=1024 K=4096", lambda: ops.gemm(x, w[:, :d], M, d, 4 * d, tb=True, out=out[:, :d]), 2.0 * M * 4 * d * d),
        ("NN dgrad dgelu N=4096 K=1024", lambda: ops.gemm(x[:, :d], w[:d], M, 4 * d, d, tb=True, dgelu_u=pre, out=out), 2.0 * M * 4 * d * d),
        ("TN wgrad      [4096x1024] tokens=48000", lambda: ops.gemm(x, pre[:, :d], 4 * d, d, M, ta=True, tb=True, out_f32=g32[:, :d], atomic=True, split_k=8), 2.0 * M * 4 * d * d),
        ("TN wgrad      [1024x1024] tokens=48000", lambda: ops.gemm(x[:, :d], pre[:, :d], d, d, M, ta=True, tb=True, out_f32=g32[:d, :d], atomic=True, split_k=32), 2.0 * M * d * d),
        ("TN wgrad      [1024x1024] split 16", lambda: ops.gemm(x[:, :d], pre[:, :d], d, d, M, ta=True, tb=True, out_f32=g32[:d, :d], atomic=True, split_k=16), 2.0 * M * d * d),
    ]
    print(f"{'case':46s} " + " ".join(f"geom{g}: ms / TF/s   " for g in geoms))
    for name, fn, flops in cases:
        row = f"{name:46s} "
        for g in geoms:
            N.lib().oasr_gemm_force_general(g)
            ms = timeit(fn)
            row += f"{ms:8.3f} {flops / ms / 1e9:7.1f}     "
        print(row, flush=True)
    N.lib().oasr_gemm_force_general(0)


if __name__ == "__main__":
    main()
