"""Where a KV-cached decode step's time goes (OLMoASR-small, B windows): host enqueue time of oasr_decode_step vs GPU time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS  # noqa: E402
from olmoasr_amd.model import OLMoASR  # noqa: E402


def main():
    from olmoasr_amd import _native as N
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
    net = OLMoASR(VARIANT_TO_DIMS[os.environ.get("OASR_PROBE_MODEL", "small")], device="cuda", seed=0, inference=True)
    for B in [int(a) for a in sys.argv[1:]] or [20]:
        for fold in (0, 1):
            N.lib().oasr_decode_set_ln_fold(fold)
            print("--- " + {1: "LayerNorm folded into the projections (default for B <= 4)", 0: "separate LayerNorm kernels (default above)"}[fold])
            probe(net, B)
            net.kv_cache_check(probe.state)
    N.lib().oasr_decode_set_ln_fold(-1)


def probe(net, B):
    mel = torch.randn(B, 80, 3000, device="cuda")
    xa = net.embed_audio(mel)
    st = net.kv_cache_begin(xa)
    probe.state = st
    tok = torch.full((B,), 50257, device="cuda", dtype=torch.int64)
    for _ in range(5):
        net.kv_cache_step(st, tok)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        net.kv_cache_step(st, tok)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"B={B}: host enqueue {1e3 * t_enq / n:.3f} ms/step, wall (enqueue + drain) {1e3 * t_all / n:.3f} ms/step")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st["pos"] = 10
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        net.kv_cache_step(st, tok)
    e1.record()
    torch.cuda.synchronize()
    print(f"GPU time between events {e0.elapsed_time(e1) / n:.3f} ms/step")


if __name__ == "__main__":
    main()
