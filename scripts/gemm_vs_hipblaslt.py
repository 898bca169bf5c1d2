"""The layer GEMMs of the benchmarked step: this repo's kernels against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the
same box, same shapes, bf16, interleaved.  The vendor call is the PLAIN product (no bias / activation / residual epilogue), i.e. an
upper bound for it; ours carries the layer's epilogue.  Then 6 s of each back to back at N = K = 4096 (sustained, at the power cap)."""
import os
import sys
import time

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


def main():
    M, d = 192000, 1024
    x = torch.randn(M, 4 * d, device=DEV).to(BF)
    w = (torch.randn(4 * d, 4 * d, device=DEV) * 0.02).to(BF)
    bias = torch.randn(4 * d, device=DEV)
    out = torch.empty(M, 4 * d, device=DEV, dtype=BF)
    pre = torch.empty(M, 4 * d, device=DEV, dtype=BF)
    resid = torch.randn(M, d, device=DEV).to(BF)
    ref = torch.empty(M, 4 * d, device=DEV, dtype=BF)
    cases = [
        ("NT attn.out  N=1024 K=1024 (+bias+residual)", lambda: ops.gemm(x[:, :d], w[:d, :d], M, d, d, bias=bias[:d], resid=resid, out=out[:, :d]),
         lambda: torch.matmul(x[:, :d], w[:d, :d].t(), out=ref[:, :d]), 2.0 * M * d * d),
        ("NT qkv       N=3072 K=1024 (+bias)", lambda: ops.gemm(x[:, :d], w[:3 * d, :d], M, 3 * d, d, bias=bias[:3 * d], out=out[:, :3 * d]),
         lambda: torch.matmul(x[:, :d], w[:3 * d, :d].t(), out=ref[:, :3 * d]), 2.0 * M * 3 * d * d),
        ("NT mlp1      N=4096 K=1024 (+bias+GELU+GELU')", lambda: ops.gemm(x[:, :d], w[:, :d], M, 4 * d, d, bias=bias, act=2, out=out, out_pre=pre),
         lambda: torch.matmul(x[:, :d], w[:, :d].t(), out=ref), 2.0 * M * 4 * d * d),
        ("NT mlp2      N=1024 K=4096 (+bias+residual)", lambda: ops.gemm(x, w[:d], M, d, 4 * d, bias=bias[:d], resid=resid, out=out[:, :d]),
         lambda: torch.matmul(x, w[:d].t(), out=ref[:, :d]), 2.0 * M * 4 * d * d),
        ("NN dgrad     N=1024 K=4096", lambda: ops.gemm(x, w[:, :d], M, d, 4 * d, tb=True, out=out[:, :d]),
         lambda: torch.matmul(x, w[:, :d], out=ref[:, :d]), 2.0 * M * 4 * d * d),
        ("NT square    N=4096 K=4096 (plain)", lambda: ops.gemm(x, w, M, 4 * d, 4 * d, out=out), lambda: torch.matmul(x, w.t(), out=ref), 2.0 * M * 16 * d * d),
    ]
    print(f"{'case (M = 192000, bf16)':50s} {'this repo':>20s} {'torch.matmul (vendor, plain)':>30s}")
    for name, ours, theirs, flops in cases:
        a = [timeit(ours), timeit(theirs), timeit(ours), timeit(theirs)]
        print(f"{name:50s} {min(a[0], a[2]):7.3f} ms {flops / min(a[0], a[2]) / 1e9:6.0f} TF/s {min(a[1], a[3]):14.3f} ms {flops / min(a[1], a[3]) / 1e9:6.0f} TF/s", flush=True)
    name, ours, theirs, flops = cases[-1]
    for label, fn in (("this repo", ours), ("torch.matmul", theirs), ("this repo", ours), ("torch.matmul", theirs)):
        t0 = time.time()
        n = 0
        while time.time() - t0 < 6:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n += 20
        dt = time.time() - t0
        print(f"sustained 6 s, N = K = 4096, {label:13s}: {flops * n / dt / 1e12:6.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
