"""Interleaved A/B of the GEMM geometries (oasr_gemm_force_general codes) on OLMoASR-medium shapes.
Usage: python scripts/gemm_ab.py [rounds] [codes...]   -- prints median / min ms per (shape, code) over interleaved rounds."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
from olmoasr_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    codes = [int(a) for a in sys.argv[2:]] or [2, 4]
    d = 1024
    shapes = []
    for M in [int(a) for a in os.environ.get("MS", "96000,28672").split(",")]:
        x = torch.randn(M, 4 * d, device="cuda").to(BF)
        w = (torch.randn(4 * d, 4 * d, device="cuda") * 0.02).to(BF)
        bias = torch.randn(4 * d, device="cuda")
        out = torch.empty(M, 4 * d, device="cuda", dtype=BF)
        pre = torch.empty(M, 4 * d, device="cuda", dtype=BF)
        resid = torch.randn(M, d, device="cuda").to(BF)
        cs = torch.zeros(4 * d, device="cuda")
        shapes += [
            (f"M={M} qkv   N=3072 K=1024 bias", lambda x=x, w=w, bias=bias, out=out, M=M: ops.gemm(x[:, :d], w[:3 * d, :d], M, 3 * d, d, bias=bias[:3 * d], out=out[:, :3 * d]), 2.0 * M * 3 * d * d),
            (f"M={M} proj  N=1024 K=1024 bias+resid", lambda x=x, w=w, bias=bias, out=out, resid=resid, M=M: ops.gemm(x[:, :d], w[:d, :d], M, d, d, bias=bias[:d], resid=resid, out=out[:, :d]), 2.0 * M * d * d),
            (f"M={M} mlp1  N=4096 K=1024 gelu", lambda x=x, w=w, bias=bias, out=out, pre=pre, M=M: ops.gemm(x[:, :d], w[:, :d], M, 4 * d, d, bias=bias, act=1, out=out, out_pre=pre), 2.0 * M * 4 * d * d),
            (f"M={M} mlp1  N=4096 K=1024 gelu+gelu' (training)", lambda x=x, w=w, bias=bias, out=out, pre=pre, M=M: ops.gemm(x[:, :d], w[:, :d], M, 4 * d, d, bias=bias, act=2, out=out, out_pre=pre), 2.0 * M * 4 * d * d),
            (f"M={M} mlp2  N=1024 K=4096 bias+resid", lambda x=x, w=w, bias=bias, out=out, resid=resid, M=M: ops.gemm(x, w[:d], M, d, 4 * d, bias=bias[:d], resid=resid, out=out[:, :d]), 2.0 * M * 4 * d * d),
            (f"M={M} dgrad N=1024 K=1024", lambda x=x, w=w, out=out, M=M: ops.gemm(x[:, :d], w[:d, :d], M, d, d, tb=True, out=out[:, :d]), 2.0 * M * d * d),
            (f"M={M} dgrad N=1024 K=3072", lambda x=x, w=w, out=out, M=M: ops.gemm(x[:, :3 * d], w[:3 * d, :d], M, d, 3 * d, tb=True, out=out[:, :d]), 2.0 * M * 3 * d * d),
            (f"M={M} dgrad N=1024 K=4096 colsum", lambda x=x, w=w, out=out, cs=cs, M=M: ops.gemm(x, w[:, :d], M, d, 4 * d, tb=True, out=out[:, :d], colsum=cs[:d]), 2.0 * M * 4 * d * d),
            (f"M={M} dgrad N=4096 K=1024 dgelu", lambda x=x, w=w, out=out, pre=pre, M=M: ops.gemm(x[:, :d], w[:d], M, 4 * d, d, tb=True, dgelu_u=pre, out=out), 2.0 * M * 4 * d * d),
        ]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    res = {}
    for r in range(rounds + 1):
        for name, fn, flops in shapes:
            for c in codes:
                N.lib().oasr_gemm_force_general(c)
                fn()
                ev[0].record()
                for _ in range(3):
                    fn()
                ev[1].record()
                torch.cuda.synchronize()
                if r:
                    res.setdefault((name, c), []).append(ev[0].elapsed_time(ev[1]) / 3)
    N.lib().oasr_gemm_force_general(0)
    print(f"{'shape':52s} " + " ".join(f"code{c}: med/min ms (TF/s med)" for c in codes))
    for name, fn, flops in shapes:
        row = f"{name:52s} "
        for c in codes:
            v = res[(name, c)]
            med = statistics.median(v)
            row += f"{med:7.3f} {min(v):7.3f} ({flops / med / 1e9:6.0f})   "
        print(row)


if __name__ == "__main__":
    main()
