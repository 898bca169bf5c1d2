"""BASELINE config 5 on a TRAINED-LIKE model: long-form transcribe in the reference's DEFAULT mode (timestamp tokens on, temperature
fallback tuple, compression / logprob / no-speech thresholds at their defaults: olmoasr/transcribe.py:47-66) on weights whose windows
close with <|30.00|> and end in eot -- what a real checkpoint does -- instead of random weights that emit 224 tokens per window and
never stop (profiles/r03_other_variants.txt: 117 audio-s/s, a property of the random-weight benchmark, not of the loop).

No checkpoint exists offline, so the model (OLMoASR-small dims, production bf16 engine) is first trained BY THIS ENGINE to memorise
`n_windows` 30 s windows of the seeded generator's audio with timestamp-format transcripts in the reference's training layout
(train_timestamps.py:401-506: <sot> <|s|> text <|e|> <|s'|> text <|e'|> ... <eot>; here three segments of 24 text tokens, the last closing
at <|30.00|>), until every supervised position leads by a margin far above bf16 noise.  Then transcribe() of the whole file is timed.
Prints one JSON line.  `python scripts/transcribe_trained_bench.py [n_windows=20] [variant=small]`"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import audio as A  # noqa: E402
from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS  # noqa: E402
from olmoasr_amd.decoding import EOT, NON_SPEECH_TOKENS_EN, SOT, TIMESTAMP_BEGIN  # noqa: E402
from olmoasr_amd.model import OLMoASR  # noqa: E402
from olmoasr_amd.synth import PAD_ID, synth_sample  # noqa: E402


def main():
    n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    variant = sys.argv[2] if len(sys.argv) > 2 else "small"
    dev = torch.device("cuda", 0)
    pcm = torch.cat([synth_sample(900 + i)[0] for i in range(n_win)])  # n_win x 30 s
    mel_padded = A.log_mel_spectrogram(pcm, padding=A.N_SAMPLES, device=dev)
    mel = torch.stack([mel_padded[:, 3000 * w:3000 * (w + 1)] for w in range(n_win)]).contiguous()
    g = torch.Generator().manual_seed(17)
    ok = torch.tensor([t for t in range(1000, 20000) if t not in set(NON_SPEECH_TOKENS_EN)])
    seqs = []
    for w in range(n_win):
        cuts = [0, int(torch.randint(350, 550, (1,), generator=g)), int(torch.randint(850, 1050, (1,), generator=g)), 1500]  # x 20 ms
        toks = [SOT]
        for s in range(3):
            toks += [TIMESTAMP_BEGIN + cuts[s]] + ok[torch.randint(0, len(ok), (24,), generator=g)].tolist() + [TIMESTAMP_BEGIN + cuts[s + 1]]
        seqs.append(toks + [EOT])
    L = len(seqs[0])
    ti = torch.full((n_win, 448), PAD_ID, dtype=torch.long)
    ty = ti.clone()
    for w, t in enumerate(seqs):
        ti[w, :L - 1] = torch.tensor(t[:-1])
        ty[w, :L - 1] = torch.tensor(t[1:])
    tl = torch.full((n_win,), L - 1, dtype=torch.int32)
    net = OLMoASR(VARIANT_TO_DIMS[variant], device=dev, seed=3)
    args = (mel, ti.to(dev), ty.to(dev), tl.to(dev))
    t_train = time.time()
    margin, step = 0.0, 0
    for step in range(1, 1501):
        log = step % 25 == 0 and step >= 100
        net.zero_grad()
        _, logits = net.loss_and_backward(*args, loss_scale=65536.0, return_logits=log, span=None if log else True)
        net.optim_step(step=step, lr=5e-4 * min(1.0, step / 20), inv_loss_scale=1.0 / 65536.0)
        if log:
            top2 = logits[:, :L - 1].float().topk(2, -1)
            right = bool((top2.indices[..., 0] == ty[:, :L - 1].to(dev)).all())
            margin = float((top2.values[..., 0] - top2.values[..., 1]).min()) if right else 0.0
            if margin > 4.0:
                break
    torch.cuda.synchronize()
    t_train = time.time() - t_train
    net._workspace = None
    torch.cuda.empty_cache()
    net.transcribe(pcm[:16000 * 60])  # warm-up (workspaces, tables), reference defaults
    torch.cuda.synchronize()
    t0 = time.time()
    out = net.transcribe(pcm)  # the reference's defaults: timestamps, temperature tuple, thresholds
    torch.cuda.synchronize()
    dt = time.time() - t0
    want = [[t for t in s[1:-1]] for s in seqs]  # what each window should decode to (sot and eot stripped)
    got_tokens = [t for s in out["segments"] for t in s["tokens"]]
    exact = got_tokens == [t for w in want for t in w]
    seeks = sorted({s["seek"] for s in out["segments"]})
    ntok = len(got_tokens) + n_win  # + one eot step per window
    print(json.dumps({
        "config": f"OLMoASR-{variant} long-form transcribe, reference defaults (timestamp tokens, temperature fallback tuple, thresholds), "
                  f"{30 * n_win} s of audio = {n_win} windows memorised by this engine ({step} AdamW steps, {t_train:.0f} s, smallest top-2 "
                  f"margin {margin:.2f}); bf16 engine, KV cache, one window per decode call (the seek depends on the timestamps)",
        "audio_seconds_per_second": round(30 * n_win / dt, 1), "wall_s": round(dt, 3), "segments": len(out["segments"]),
        "windows": len(seeks), "seek_step_is_3000": seeks == [3000 * w for w in range(n_win)], "tokens_exact": exact,
        "temperatures_used": sorted({s["temperature"] for s in out["segments"]}), "decode_steps": ntok,
        "ms_per_window": round(1000 * dt / max(1, len(seeks)), 2),
        "ms_per_decode_step_incl_encoder": round(1000 * dt / max(1, ntok), 3)}))


if __name__ == "__main__":
    main()
