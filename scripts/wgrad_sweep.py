"""Split-K sweep of the weight-gradient GEMM (dW[N,K] += dY[tokens,N]^T X[tokens,K]) over geometries.
Usage: python scripts/wgrad_sweep.py [rounds]"""
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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    d = 1024
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for tokens in [int(a) for a in os.environ.get("TOKENS", "96000,28672").split(",")]:
        dy = (torch.randn(tokens, 4 * d, device="cuda") * 0.05).to(BF)
        x = torch.randn(tokens, 4 * d, device="cuda").to(BF)
        g32 = torch.zeros(4 * d, 4 * d, device="cuda")
        shapes = [tuple(int(v) for v in a.split('x')) for a in os.environ.get('SHAPES', '1024x1024,3072x1024,4096x1024,1024x4096').split(',')]
        for (n, k) in shapes:
            flops = 2.0 * tokens * n * k
            res = {}
            splits = [int(a) for a in os.environ.get('SPLITS', '1,2,4,8,16,24,32').split(',')]
            codes = [int(a) for a in os.environ.get('CODES', '2,4').split(',')]
            cfgs = [(c, s) for c in codes for s in splits if s * (n // 256) * (k // (128 if c == 2 else 256)) <= 2048]
            for r in range(rounds + 1):
                for c, s in cfgs:
                    N.lib().oasr_gemm_force_general(c)
                    fn = lambda: ops.gemm(dy[:, :n], x[:, :k], n, k, tokens, ta=True, tb=True, out_f32=g32[:n, :k], atomic=True, split_k=s)
                    fn()
                    ev[0].record()
                    for _ in range(3):
                        fn()
                    ev[1].record()
                    torch.cuda.synchronize()
                    if r:
                        res.setdefault((c, s), []).append(ev[0].elapsed_time(ev[1]) / 3)
            row = f"tokens={tokens} dW[{n}x{k}]: "
            for c in codes:
                row += f"\n     code{c}: " + " ".join(f"s{s}:{statistics.median(v):.3f}ms({flops / statistics.median(v) / 1e9:.0f})" for (cc, s), v in sorted(res.items()) if cc == c)
            print(row, flush=True)
    N.lib().oasr_gemm_force_general(0)


if __name__ == "__main__":
    main()
