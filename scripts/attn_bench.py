"""Micro-benchmark of the flash-attention kernels on the OLMoASR-medium shapes (HIP events, random data)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=5):
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
    B, H = 32, 16
    d = H * 64
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    for name, Tq, Tk, causal in (("encoder self", 1500, 1500, False), ("cross", 448, 1500, False), ("decoder self", 448, 448, True)):
        if Tq == Tk:
            qkv = torch.randn(B, Tq, 3 * d, device=DEV).to(BF)
            q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(3))
        else:
            qb = torch.randn(B, Tq, d, device=DEV).to(BF)
            kvb = torch.randn(B, Tk, 2 * d, device=DEV).to(BF)
            q = qb.unflatten(2, (H, 64))
            k, v = (kvb[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(2))
        kv_len = torch.randint(8, 221, (B,), device=DEV, dtype=torch.int32) if causal else None
        d_o = torch.randn(B, Tq, d, device=DEV).to(BF)
        o, lse, o_lo = ops.attention_fwd(q, k, v, kv_len, causal, want_o_lo=True)
        flops = 4.0 * B * H * Tq * Tk * 64 * (0.5 if causal else 1.0)
        tf = timeit(lambda: ops.attention_fwd(q, k, v, kv_len, causal, want_o_lo=True), iters)
        tb = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, d_o, kv_len, causal, o_lo=o_lo), iters)
        print(f"{name:14s} Tq={Tq} Tk={Tk}: fwd {tf:7.3f} ms {flops / tf / 1e9:7.1f} TF/s | bwd {tb:7.3f} ms {2.5 * flops / tb / 1e9:7.1f} TF/s (algorithmic 2.5x)",
              flush=True)
        if os.environ.get("OASR_ATTN_VS_SDPA") and kv_len is None:
            # the vendor's flash attention (torch SDPA -> AOTriton / CK on ROCm) on the same problem, [B, H, T, 64] layout, no mask
            import torch.nn.functional as F
            qs, ks, vs = (t.permute(0, 2, 1, 3).contiguous().requires_grad_(True) for t in (q, k, v))
            go = d_o.unflatten(2, (H, 64)).permute(0, 2, 1, 3).contiguous()
            try:
                with torch.nn.attention.sdpa_kernel([torch.nn.attention.SDPBackend.FLASH_ATTENTION]):
                    ts = timeit(lambda: F.scaled_dot_product_attention(qs, ks, vs, is_causal=False), iters)
                    out = F.scaled_dot_product_attention(qs, ks, vs, is_causal=False)
                    tsb = timeit(lambda: torch.autograd.grad(out, (qs, ks, vs), go, retain_graph=True), iters)
                print(f"{'':14s} torch SDPA (flash backend): fwd {ts:7.3f} ms {flops / ts / 1e9:7.1f} TF/s | bwd {tsb:7.3f} ms {2.5 * flops / tsb / 1e9:7.1f} TF/s", flush=True)
            except Exception as e:  # backend not available for this shape on this build
                print(f"{'':14s} torch SDPA flash backend unavailable: {str(e)[:120]}", flush=True)


if __name__ == "__main__":
    main()
