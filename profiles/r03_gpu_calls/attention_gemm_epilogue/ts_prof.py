import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, "/root/repo")
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
from olmoasr_amd.model import OLMoASR
opts = dict(without_timestamps=False, temperature=0.0, logprob_threshold=None, no_speech_threshold=None)
dev = torch.device("cuda", 0)
net = OLMoASR(VARIANT_TO_DIMS["small"], device=dev, seed=0, inference=True)
g = torch.Generator().manual_seed(0)
audio = (torch.randn(600 * 16000, generator=g) * 0.1).clamp_(-1, 1)
net.transcribe(audio[:16000 * 60], batch_windows=1, **opts)
torch.cuda.synchronize()
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
out = net.transcribe(audio, batch_windows=1, **opts)
torch.cuda.synchronize()
pr.disable()
dt = time.time() - t0
ntok = sum(len(s["tokens"]) for s in out["segments"])
print("wall", dt, "segments", len(out["segments"]), "tokens", ntok, "audio-s/s", 600 / dt)
print("first segments:", [(round(s["start"], 2), round(s["end"], 2), len(s["tokens"])) for s in out["segments"][:6]])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
