"""Reads the per-wave phase cycle sums of the stamped attention forward (scripts/probes/attn_fwd_stamps_patch.py; library through OASR_LIB) on the encoder
problem of scripts/attn_bench.py and prints mean cycles per 64-key tile and wave for each segment, next to the launch's wall time."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from olmoasr_amd import _native as N  # noqa: E402
from olmoasr_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    B, H, T = int(os.environ.get("PB", 32)), 16, 1500
    d = H * 64
    qkv = torch.randn(B, T, 3 * d, device="cuda").to(BF)
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(3))
    fn = lambda: ops.attention_fwd(q, k, v, None, False, want_o_lo=True)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    print(f"encoder self-attention forward B={B} H={H} T={T}: {ms * 1e3:.0f} us per launch = {4.0 * B * H * T * T * 64 / ms / 1e9:.0f} TF/s")
    lib = N.lib()
    if not hasattr(lib, "oasr_attn_dbg_read"):
        print("(product library: no stamps)")
        return
    nwg = ((T + 127) // 128) * B * H
    n = min(nwg * 4, 4 * 32768)
    buf = np.zeros(n * 8, dtype=np.uint64)
    lib.oasr_attn_dbg_read.argtypes = [C.c_void_p, C.c_size_t]
    rc = lib.oasr_attn_dbg_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, rc
    r = buf.reshape(n, 8).astype(np.float64)
    tiles = r[:, 6]
    ok = tiles > 0
    per = r[ok, :6] / tiles[ok, None]
    names = ["S^T MFMAs issued (K fragment reads + issue)", "row maximum known (waits for the matrix pipe; max3 chain, exchange, rescale test)",
             "exponentials + sums + packs (VALU)", "O^T MFMAs issued (V^T fragment reads + issue)", "next tile committed to LDS (global prefetch wait)",
             "workgroup barrier"]
    tot = per.sum(axis=1).mean()
    print(f"{int(ok.sum())} waves, {tiles[ok].mean():.1f} key tiles each; mean cycles per 64-key tile and wave (3 waves per SIMD share the SIMD): total {tot:.0f}")
    for i, nm in enumerate(names):
        print(f"  seg {i}: {per[:, i].mean():7.0f}  ({100 * per[:, i].mean() / tot:4.1f} %)  p10 {np.percentile(per[:, i], 10):6.0f}  p90 {np.percentile(per[:, i], 90):6.0f}   {nm}")


if __name__ == "__main__":
    main()
