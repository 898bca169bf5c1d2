"""Checksums of the attention forward's outputs (o, lse, o_lo) on fixed seeded problems -- encoder, cross, causal + key lengths, ragged sizes -- to compare two
builds of the library bit for bit (OASR_LIB selects the build)."""
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

BF = torch.bfloat16


def crc(t):
    return zlib.crc32(t.contiguous().view(torch.uint8).cpu().numpy().tobytes())


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, B, H, Tq, Tk, causal in (("encoder", 4, 16, 1500, 1500, False), ("cross", 4, 16, 448, 1500, False), ("causal", 6, 16, 448, 448, True),
                                       ("ragged", 3, 5, 200, 333, False), ("short", 2, 3, 70, 1, False), ("causal-ragged", 3, 4, 130, 130, True)):
        q = torch.randn(B, Tq, H, 64, device="cuda", generator=g).to(BF)
        k = torch.randn(B, Tk, H, 64, device="cuda", generator=g).to(BF)
        v = torch.randn(B, Tk, H, 64, device="cuda", generator=g).to(BF)
        kv_len = torch.randint(1, Tk + 1, (B,), device="cuda", generator=g, dtype=torch.int32) if causal else None
        o, lse, o_lo = ops.attention_fwd(q, k, v, kv_len, causal, want_o_lo=True)
        print(f"{name:14s} o {crc(o):08x} lse {crc(lse):08x} o_lo {crc(o_lo):08x}", flush=True)


if __name__ == "__main__":
    main()
