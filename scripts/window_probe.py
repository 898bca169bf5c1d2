"""Where a long-form transcribe window's time goes (OLMoASR-small, B = 1): encoder pass, decode_begin, decode steps, host profile."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
from olmoasr_amd.decoding import DecodingOptions, decode
from olmoasr_amd.model import OLMoASR
net = OLMoASR(VARIANT_TO_DIMS["small"], device="cuda", seed=0, inference=True)
mel = torch.randn(1, 80, 3000, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("embed_audio B=1       %.2f ms" % t(lambda: net.embed_audio(mel)))
xa = net.embed_audio(mel)
print("kv_cache_begin        %.2f ms" % t(lambda: net.kv_cache_begin(xa)))
print("decode sample_len=4   %.2f ms" % t(lambda: decode(net, mel, DecodingOptions(sample_len=4))))
print("decode sample_len=4 nots %.2f ms" % t(lambda: decode(net, mel, DecodingOptions(sample_len=4, without_timestamps=True))))
audio = (torch.randn(16000 * 120) * 0.1).clamp_(-1, 1)
G = dict(temperature=0.0, logprob_threshold=None, no_speech_threshold=None)
net.transcribe(audio, **G)
torch.cuda.synchronize(); t0 = time.perf_counter(); out = net.transcribe(audio, **G); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("transcribe 120 s: %.0f ms, %d segments -> %.1f ms per segment" % (1e3 * dt, len(out["segments"]), 1e3 * dt / max(1, len(out["segments"]))))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); net.transcribe(audio, **G); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
