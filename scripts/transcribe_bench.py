"""BASELINE config 5: OLMoASR-small greedy long-form transcribe (inf_model path) on one GPU, 10 min of synthetic audio.
Random-init weights (no checkpoints offline): every window decodes the full sample_len = 224 tokens unless EOT is sampled,
i.e. the slowest case.  Prints audio-seconds per second and the per-token decode latency."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS  # noqa: E402
from olmoasr_amd.model import OLMoASR  # noqa: E402


# BASELINE config 5 is greedy (temperature 0, no fallback, no timestamps): the reference's defaults would walk the
# temperature tuple on a random-init model whose avg_logprob is far below the -1.0 threshold
GREEDY = dict(without_timestamps=True, temperature=0.0, logprob_threshold=None, no_speech_threshold=None)


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "small"
    seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    bw = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    opts = dict(GREEDY)
    if len(sys.argv) > 4 and sys.argv[4] == "ts":  # the reference's default: timestamp tokens on, windows strictly sequential (seek depends on them)
        opts["without_timestamps"] = False
    dev = torch.device("cuda", 0)
    net = OLMoASR(VARIANT_TO_DIMS[variant], device=dev, seed=0, inference=True)
    g = torch.Generator().manual_seed(0)
    audio = (torch.randn(seconds * 16000, generator=g) * 0.1).clamp_(-1, 1)
    net.transcribe(audio[:16000 * 60], batch_windows=bw, **opts)  # warm-up (workspaces, tables)
    torch.cuda.synchronize()
    t0 = time.time()
    out = net.transcribe(audio, batch_windows=bw, **opts)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ntok = sum(len(s["tokens"]) for s in out["segments"])
    print(json.dumps({"config": f"OLMoASR-{variant} greedy transcribe, {seconds} s synthetic audio, {bw} windows per decode batch, KV cache" + (", timestamp tokens (sequential windows)" if not opts["without_timestamps"] else ""),
                      "audio_seconds_per_second": round(seconds / dt, 1), "wall_s": round(dt, 3), "windows": len(out["segments"]),
                      "tokens": ntok, "ms_per_decode_step": round(1000 * dt / max(1, ntok / (bw if opts["without_timestamps"] else 1)), 3)}))


if __name__ == "__main__":
    main()
