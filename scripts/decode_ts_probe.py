"""Per-step cost of greedy decoding with and without the timestamp rules (B windows, 200 steps, eot suppressed so every step runs)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS  # noqa: E402
from olmoasr_amd.decoding import EOT, DecodingOptions, decode  # noqa: E402
from olmoasr_amd.model import OLMoASR  # noqa: E402


def main():
    net = OLMoASR(VARIANT_TO_DIMS["small"], device="cuda", seed=0, inference=True)
    bias = torch.zeros(net.dims.n_vocab, device="cuda")
    bias[EOT] = -float("inf")
    for B in [int(a) for a in sys.argv[1:]] or [1, 8]:
        mel = torch.randn(B, 80, 3000, device="cuda")
        for wts in (True, False):
            opt = DecodingOptions(sample_len=200, without_timestamps=wts, suppress_mask=bias)
            decode(net, mel, opt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = decode(net, mel, opt)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n = len(r[0].tokens)
            print(f"B={B} without_timestamps={wts}: {n} tokens, {1e3 * dt / max(n, 1):.3f} ms per step (incl. encoder + decode_begin {1e3 * dt:.0f} ms total)", flush=True)


if __name__ == "__main__":
    main()
