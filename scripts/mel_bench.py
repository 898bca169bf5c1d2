"""Times oasr_log_mel (int16 PCM -> log-mel) per 30 s clip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import ops  # noqa: E402

pcm = (torch.randn(64, 480000, device="cuda") * 0.1).clamp_(-1, 1).mul_(32767).round_().to(torch.int16)
ops.log_mel(pcm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record()
    for _ in range(10):
        ops.log_mel(pcm)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10 / 64 * 1000)
print(f"log_mel {best:.2f} us/clip = {1.92e6 / best / 1e6:.3f} TB/s of algorithmic bytes (lib {os.environ.get('OASR_LIB', 'default')})")
