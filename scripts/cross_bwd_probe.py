"""Cross-attention backward under the span step's layout (chunked query rows, ~2.3 active 64-row chunks per sample of 7; keys / values plain
[B, 1500, 2d]) at the bench's launch shape, with the keys / values (a) distinct per sample (the real case: every tile is an HBM miss for the one
workgroup that reads it) and (b) SHARED by all samples (batch stride 0: tiles come out of L2) -- how much of these kernels is memory latency?
Run under `rocprofv3 --kernel-trace --stats` for per-kernel times; prints the pair's total by HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    B, H, S, Tk = int(os.environ.get("PB", 128)), 16, 448, 1500
    d = H * 64
    g = torch.Generator().manual_seed(0)
    span = torch.randint(24, 260, (B,), generator=g, dtype=torch.int32)
    nch = S // 64
    act = [(b, c) for c in range(nch) for b in range(B) if 64 * c < int(span[b])]
    ina = [(b, c) for c in range(nch) for b in range(B) if 64 * c >= int(span[b])]
    tab = ops.chunk_rows_table(act + ina, B, nch).cuda()
    span_d = span.cuda()
    qc = torch.randn(B * S, H, 64, device="cuda").to(BF)
    kv = torch.randn(B, Tk, 2, H, 64, device="cuda").to(BF)
    doc = torch.randn(B * S, d, device="cuda").to(BF)
    R = len(act) * 64
    doc[R:] = 0
    for shared in (False, True, False, True):
        src = kv[:1].expand(B, Tk, 2, H, 64) if shared else kv
        kc, vc = src[:, :, 0], src[:, :, 1]
        oc, lse, o_lo = ops.attention_fwd_rows(qc, kc, vc, B, H, S, Tk, tab, want_o_lo=True)
        fn = lambda: ops.attention_bwd_rows(qc, kc, vc, oc, lse, doc, B, H, S, Tk, tab, q_span=span_d, o_lo=o_lo)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        print(f"B={B} active rows {R} of {B * S}; keys/values {'SHARED (L2)' if shared else 'per sample (HBM)'}: dq + dkdv {ev[0].elapsed_time(ev[1]) / 10 * 1e3:.0f} us", flush=True)


if __name__ == "__main__":
    main()
