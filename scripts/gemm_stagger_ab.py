"""A/B of the ping-pong GEMM's first-wave phase stagger on the encoder shapes of the benchmarked step (M = 192000)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
from olmoasr_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=8):
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
    M, d = 192000, 1024
    x = torch.randn(M, 4 * d, device=DEV).to(BF)
    w = (torch.randn(4 * d, 4 * d, device=DEV) * 0.02).to(BF)
    bias = torch.randn(4 * d, device=DEV)
    out = torch.empty(M, 4 * d, device=DEV, dtype=BF)
    pre = torch.empty(M, 4 * d, device=DEV, dtype=BF)
    resid = torch.randn(M, d, device=DEV).to(BF)
    cases = [
        ("NT attn.out resid N=1024 K=1024", lambda: ops.gemm(x[:, :d], w[:d, :d], M, d, d, bias=bias[:d], resid=resid, out=out[:, :d]), 2.0 * M * d * d),
        ("NT qkv bias   N=3072 K=1024", lambda: ops.gemm(x[:, :d], w[:3 * d, :d], M, 3 * d, d, bias=bias[:3 * d], out=out[:, :3 * d]), 2.0 * M * 3 * d * d),
        ("NT mlp1 gelu  N=4096 K=1024", lambda: ops.gemm(x[:, :d], w[:, :d], M, 4 * d, d, bias=bias, act=2, out=out, out_pre=pre), 2.0 * M * 4 * d * d),
        ("NT mlp2 resid N=1024 K=4096", lambda: ops.gemm(x, w[:d], M, d, 4 * d, bias=bias[:d], resid=resid, out=out[:, :d]), 2.0 * M * 4 * d * d),
        ("NN dgrad      N=1024 K=4096", lambda: ops.gemm(x, w[:, :d], M, d, 4 * d, tb=True, out=out[:, :d]), 2.0 * M * 4 * d * d),
        ("NN dgrad dgelu N=4096 K=1024", lambda: ops.gemm(x[:, :d], w[:d], M, 4 * d, d, tb=True, dgelu_u=pre, dgelu_deriv=True, out=out), 2.0 * M * 4 * d * d),
    ]
    if len(sys.argv) > 1 and sys.argv[1] == "colsum":  # what the fused bias-gradient column sums cost the GELU' dgrad
        cs = torch.zeros(4 * d, device=DEV)
        a = lambda: ops.gemm(x[:, :d], w[:d], M, 4 * d, d, tb=True, dgelu_u=pre, dgelu_deriv=True, out=out)
        b = lambda: ops.gemm(x[:, :d], w[:d], M, 4 * d, d, tb=True, dgelu_u=pre, dgelu_deriv=True, out=out, colsum=cs)
        for v in (2, 7):
            N.lib().oasr_gemm_set_variant(v)
            for _ in range(2):
                print(f"variant {v}: dgelu dgrad {timeit(a):.3f} ms, with fused colsum {timeit(b):.3f} ms", flush=True)
        N.lib().oasr_gemm_set_variant(-1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "persist":  # plain launches (24) against persistent ones (40), default kernels
        print(f"{'case':34s}  launch plain persistent plain persistent: ms TF/s")
        for name, fn, flops in cases:
            row = f"{name:34s} "
            for v in (24, 40, 24, 40):
                N.lib().oasr_gemm_set_variant(v)
                ms = timeit(fn)
                row += f"{ms:7.3f} {flops / ms / 1e9:6.0f} | "
            print(row, flush=True)
        N.lib().oasr_gemm_set_variant(-1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variant":  # A/B of where the ping-pong kernel issues its DMA pieces
        print(f"{'case':34s}  variants 7 15 7 15 (bit 3 = both wave groups in lockstep, no ping-pong offset) (bit 0 = DMA between the MFMAs, bit 1 = MFMA sections pinned, bit 2 = two 16-MFMA sections per K-tile): ms TF/s")
        for name, fn, flops in cases:
            row = f"{name:34s} "
            for v in (7, 15, 7, 15):
                N.lib().oasr_gemm_set_variant(v)
                ms = timeit(fn)
                row += f"{ms:7.3f} {flops / ms / 1e9:6.0f} | "
            print(row, flush=True)
        N.lib().oasr_gemm_set_variant(-1)
        return
    settings = [(0, 2), (2, 2), (5, 2), (2, 4), (3, 4), (1, 8)]
    print(f"{'case':34s} " + " ".join(f"s{a}p{b}: ms TF/s " for a, b in settings))
    for name, fn, flops in cases:
        row = f"{name:34s} "
        for a, b in settings:
            N.lib().oasr_gemm_set_stagger(a, b)
            ms = timeit(fn)
            row += f"{ms:7.3f} {flops / ms / 1e9:6.0f} | "
        print(row, flush=True)
    N.lib().oasr_gemm_set_stagger(0, 2)


if __name__ == "__main__":
    main()
