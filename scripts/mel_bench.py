"""Times the log-mel front end (int16 PCM -> log-mel) per 30 s clip: the kernel alone (oasr_log_mel_raw: log10 mel power + per-clip maximum,
what the training step runs) and with whisper's floor / scale pass (oasr_log_mel).  MEL_CLIPS: clips per launch (default 128, the bench's micro-batch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

n = int(os.environ.get("MEL_CLIPS", "128"))
pcm = (torch.randn(n, 480000, device="cuda") * 0.1).clamp_(-1, 1).mul_(32767).round_().to(torch.int16)


def best_us(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 / n * 1000)
    return best


raw = best_us(lambda: ops.log_mel(pcm, finalize=False))
full = best_us(lambda: ops.log_mel(pcm))
print(f"log_mel kernel alone {raw:.3f} us/clip = {1.92e6 / raw / 1e6:.3f} TB/s = {1.92e6 / raw / 1e6 / 8:.3f} of 8 TB/s | with finalize {full:.3f} us/clip = "
      f"{1.92e6 / full / 1e6:.3f} TB/s ({n} clips per launch, lib {os.path.basename(os.environ.get('OASR_LIB', 'default'))}, OASR_LOGMEL={os.environ.get('OASR_LOGMEL', '-')})")
