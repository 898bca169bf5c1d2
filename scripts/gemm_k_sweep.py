"""Per-tile cost model of the ping-pong GEMM: time(K) at fixed M x N for several epilogues -> fixed cost per 256x256 tile
(prologue + epilogue + launch) and cost per K-tile, plain launches against persistent ones."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
from olmoasr_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


# oasr_gemm_set_variant: 8 = default kernels; +16 plain launch / +32 persistent; +64 non-temporal stores; +128 non-temporal side loads
MODES = (("plain-launch", 24), ("persistent", 40), ("plain nt-st", 24 + 64), ("plain nt-st+ld", 24 + 192), ("persist nt-st+ld", 40 + 192))


def few_tiles():
    """One round of 32 / 128 / 256 tiles: is a tile's fixed cost its own latency chain or the chip-wide burst of epilogue bytes?"""
    import numpy as np
    Nn = 4096
    Ks = [128, 512, 1024, 4096]
    N.lib().oasr_gemm_force_general(4)
    N.lib().oasr_gemm_set_variant(24)
    try:
        for rows in (512, 2048, 4096, 8192):
            M = rows
            x = torch.randn(M, 4096, device=DEV).to(BF)
            w = (torch.randn(4096, 4096, device=DEV) * 0.02).to(BF)
            bias = torch.randn(4096, device=DEV)
            out = torch.empty(M, 4096, device=DEV, dtype=BF)
            pre = torch.empty(M, 4096, device=DEV, dtype=BF)
            resid = torch.randn(M, 4096, device=DEV).to(BF)
            epis = {
                "bias": lambda K: ops.gemm(x[:, :K], w[:Nn, :K], M, Nn, K, bias=bias[:Nn], out=out[:, :Nn]),
                "bias+resid": lambda K: ops.gemm(x[:, :K], w[:Nn, :K], M, Nn, K, bias=bias[:Nn], resid=resid[:, :Nn], out=out[:, :Nn]),
                "bias+gelu(train)": lambda K: ops.gemm(x[:, :K], w[:Nn, :K], M, Nn, K, bias=bias[:Nn], act=2, out=out[:, :Nn], out_pre=pre[:, :Nn]),
            }
            tiles = (M // 256) * (Nn // 256)
            rounds = max(1.0, tiles / 256.0)
            for name, fn in epis.items():
                ts = [timeit(lambda: fn(K), iters=20) for K in Ks]
                A = np.stack([np.ones(len(Ks)), np.array(Ks) / 64.0], 1) * rounds
                a, b = np.linalg.lstsq(A, np.array(ts) * 1e3, rcond=None)[0]
                row = " ".join(f"K={K}:{t * 1e3:.1f}us" for K, t in zip(Ks, ts))
                print(f"tiles={tiles:4d} {name:18s} {row}  -> fixed {a:6.2f} us (incl. launch) + {b:5.3f} us/K-tile", flush=True)
    finally:
        N.lib().oasr_gemm_force_general(0)
        N.lib().oasr_gemm_set_variant(-1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "few":
        return few_tiles()
    M = 192000
    Ks = [128, 512, 1024, 4096]
    x = torch.randn(M, 4096, device=DEV).to(BF)
    w = (torch.randn(4096, 4096, device=DEV) * 0.02).to(BF)
    bias = torch.randn(4096, device=DEV)
    out = torch.empty(M, 4096, device=DEV, dtype=BF)
    pre = torch.empty(M, 4096, device=DEV, dtype=BF)
    resid = torch.randn(M, 4096, device=DEV).to(BF)
    N.lib().oasr_gemm_force_general(4)
    try:
        for Nn in (1024, 4096):
            tiles = (M // 256) * (Nn // 256)
            rounds = tiles / 256.0
            epis = {
                "bias": lambda K: ops.gemm(x[:, :K], w[:Nn, :K], M, Nn, K, bias=bias[:Nn], out=out[:, :Nn]),
                "bias+resid": lambda K: ops.gemm(x[:, :K], w[:Nn, :K], M, Nn, K, bias=bias[:Nn], resid=resid[:, :Nn], out=out[:, :Nn]),
                "bias+gelu(train)": lambda K: ops.gemm(x[:, :K], w[:Nn, :K], M, Nn, K, bias=bias[:Nn], act=2, out=out[:, :Nn], out_pre=pre[:, :Nn]),
            }
            for name, fn in epis.items():
                for mode, v in MODES:
                    N.lib().oasr_gemm_set_variant(v)
                    ts = [timeit(lambda: fn(K)) for K in Ks]
                    # least squares t = (a + b * K/64) * rounds
                    import numpy as np
                    A = np.stack([np.ones(len(Ks)), np.array(Ks) / 64.0], 1) * rounds
                    a, b = np.linalg.lstsq(A, np.array(ts) * 1e3, rcond=None)[0]
                    row = " ".join(f"K={K}:{t:.3f}" for K, t in zip(Ks, ts))
                    print(f"N={Nn:5d} {name:18s} {mode:17s} {row}  -> per tile: fixed {a:6.2f} us + {b:5.3f} us/K-tile (MFMA floor 0.860)", flush=True)
    finally:
        N.lib().oasr_gemm_force_general(0)
        N.lib().oasr_gemm_set_variant(-1)


if __name__ == "__main__":
    main()
