"""KV-cached decoder step at B sequences: GPU time per step of every step engine (oasr_decode_set_ln_fold modes) at a fixed position, and the
HBM-roofline fraction of the streamed bytes (bench.py's `decode_step(B=1)` accounting).
    python scripts/decode_xcd_probe.py [variant=medium] [B=1] [pos=32] [modes=1,2,3,4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS  # noqa: E402
from olmoasr_amd.model import OLMoASR  # noqa: E402

NAMES = {-1: "library default", 0: "separate LayerNorm kernels", 1: "multi-launch, LayerNorm folded (round 2-4 default for B <= 4)", 2: "ONE launch, 32 CUs of one XCD",
         3: "ONE launch, 32 workgroups spread over the chip", 4: "ONE launch, 64 workgroups spread over the chip",
         5: "ONE launch, every CU (decode_wide.hip)"}


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "medium"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pos = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    modes = [int(m) for m in (sys.argv[4] if len(sys.argv) > 4 else "1,2,3,4").split(",")]
    N.enable_testing_hooks()
    dims = VARIANT_TO_DIMS[variant]
    net = OLMoASR(dims, device="cuda", seed=0, inference=True)
    d, L, V = dims.n_text_state, dims.n_text_layer, dims.n_vocab
    xa = torch.randn(B, dims.n_audio_ctx, d, device="cuda").to(torch.bfloat16)
    tok = torch.full((B,), 50257, device="cuda", dtype=torch.int64)
    dbytes = 2 * (L * 14 * d * d + V * d) + B * 2 * L * (dims.n_audio_ctx * 2 * d + pos * 2 * d) + 4 * V * B
    ref = None
    for mode in modes:
        N.lib().oasr_decode_set_ln_fold(mode)
        st = net.kv_cache_begin(xa)
        outs = []
        for _ in range(pos):
            outs.append(net.kv_cache_step(st, tok))
        net.kv_cache_check(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            st["pos"] = pos
            last = net.kv_cache_step(st, tok)
        e1.record()
        torch.cuda.synchronize()
        net.kv_cache_check(st)
        ms = e0.elapsed_time(e1) / n
        ctrl = st["cache"][-N.KV_TAIL_BYTES:].view(torch.int32)[:8].tolist()
        same = "" if ref is None else f"  logits == mode {modes[0]}: {bool(torch.equal(ref, last))} (max |d| {float((ref - last).abs().max()):.3g})"
        if ref is None:
            ref = last.clone()
        if mode == 5 and (int(os.environ.get("OASR_XCD_FLAGS", "0")) & 0x100):
            t = st["ws"][-512:].view(torch.int64).view(8, 8).cpu().tolist()
            names = ["qkv", "self-attn", "attn.out", "cross q", "cross-attn", "cross out", "mlp.0", "mlp.2"]
            print("   in-kernel s_memtime stamps, workgroup 0, decoder layer 1 (ticks): projection = poll | operand | units | epilogue+store ; attention = poll | rest")
            for p_, r in enumerate(t):
                nxt = t[p_ + 1][0] if p_ < 7 else None
                if p_ == 1:
                    print(f"     {names[p_]:10s} poll {r[1] - r[0]:6d}  q/K/V landed {r[3] - r[1]:6d}  attend {r[6] - r[3]:6d}  store {r[2] - r[6]:6d}  end {r[7] - r[2]:6d}"
                          + (f"  | to next phase {nxt - r[7]:6d}  | phase total {nxt - r[0]:6d}" if nxt else ""))
                elif p_ == 4:
                    print(f"     {names[p_]:10s} poll {r[1] - r[0]:6d}  rest {r[2] - r[1]:6d}  end {r[7] - r[2]:6d}" + (f"  | to next phase {nxt - r[7]:6d}  | phase total {nxt - r[0]:6d}" if nxt else ""))
                else:
                    print(f"     {names[p_]:10s} poll {r[1] - r[0]:6d}  operand {r[2] - r[1]:6d}  units {r[3] - r[2]:6d} (wave 1: weights landed +{r[5] - r[2]:5d}, sums done +{r[6] - r[2]:5d})  epilogue {r[4] - r[3]:6d}  end {r[7] - r[4]:6d}"
                          + (f"  | to next phase {nxt - r[7]:6d}  | phase total {nxt - r[0]:6d}" if nxt else ""))
            print(f"     layer total {t[7][4] - t[0][0]} ticks")
        elif mode >= 2 and (int(os.environ.get("OASR_XCD_FLAGS", "0")) & 0x100):
            t = st["ws"][-512:].view(torch.int64).view(8, 8).cpu().tolist()
            names = ["qkv", "self-attn", "attn.out", "cross q", "cross-attn", "cross out", "mlp.0", "mlp.2"]
            print("   in-kernel s_memtime stamps, workgroup 0, decoder layer 1 (ticks; helper: wait | operand | tiles+epilogue | arrive ; streaming wave 0: released->done):")
            for p_, r in enumerate(t):
                nxt = t[p_ + 1][0] if p_ < 7 else None
                print(f"     {names[p_]:10s} wait {r[1] - r[0]:6d}  operand {r[2] - r[1]:6d}  tiles {r[3] - r[2]:6d}  arrive {r[4] - r[3]:6d}  | stream {r[6] - r[5] if r[6] and r[5] else 0:6d}"
                      f"  | phase total {(nxt - r[0]) if nxt else r[4] - r[0]:6d}")
            print(f"     layer total {t[7][4] - t[0][0]} ticks")
        print(f"{variant} B={B} pos={pos} mode {mode} [{NAMES[mode]}]: {ms:.3f} ms/step = {dbytes / ms / 1e6:.0f} GB/s = {dbytes / ms / 1e6 / 8000:.3f} of 8 TB/s; "
              f"ctrl (counter, flag, epoch, xcc mask, wide epoch) = {ctrl[0]} {ctrl[1]:#x} {ctrl[2]} {ctrl[3]:#x} {ctrl[4]}{same}", flush=True)
    N.lib().oasr_decode_set_ln_fold(-1)


if __name__ == "__main__":
    main()
