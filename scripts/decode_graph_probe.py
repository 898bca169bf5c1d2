"""Does replaying a decode step as a hipGraph shorten it?  (Timing experiment: the captured step has a FIXED position, so only the
duration of the replays is meaningful.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import _native as N  # noqa: E402
N.enable_testing_hooks()  # noqa: E402 -- this script steers kernel selection (include/oasr_testing.h)
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS  # noqa: E402
from olmoasr_amd.model import OLMoASR  # noqa: E402


def main():
    net = OLMoASR(VARIANT_TO_DIMS["small"], device="cuda", seed=0, inference=True)
    for B in [int(a) for a in sys.argv[1:]] or [1, 16]:
        for mode in (0, 1):
            N.lib().oasr_decode_set_ln_fold(mode)
            mel = torch.randn(B, 80, 3000, device="cuda")
            xa = net.embed_audio(mel)
            st = net.kv_cache_begin(xa)
            tok = torch.full((B,), 50257, device="cuda", dtype=torch.int64)
            for _ in range(12):
                net.kv_cache_step(st, tok)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st["pos"] = 12
            e0.record()
            for _ in range(100):
                st["pos"] = 12
                net.kv_cache_step(st, tok)
            e1.record()
            torch.cuda.synchronize()
            eager = e0.elapsed_time(e1) / 100
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                st["pos"] = 12
                net.kv_cache_step(st, tok)  # warm-up on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            st["pos"] = 12
            with torch.cuda.graph(g):
                out = net.kv_cache_step(st, tok)
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(100):
                g.replay()
            e1.record()
            host = (time.perf_counter() - t0) / 100
            torch.cuda.synchronize()
            print(f"B={B} mode={mode}: eager {eager:.3f} ms/step, hipGraph replay {e0.elapsed_time(e1) / 100:.3f} ms/step (host {1e3 * host:.3f} ms per replay)", flush=True)
    N.lib().oasr_decode_set_ln_fold(-1)


if __name__ == "__main__":
    main()
