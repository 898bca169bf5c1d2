"""Harness that pins ``oracle.decode_oracle.transcribe`` to the reference's OWN long-form driver.  TEST INFRASTRUCTURE ONLY.

``olmoasr/transcribe.py:47-523`` is in the reference tree and is executed here UNMODIFIED (``ref_import.load_transcribe``);
only the names it imports from the un-vendored ``whisper`` package are supplied:

  log_mel_spectrogram(audio, n_mels, padding)  -> the padded mel handed in as ``audio`` (transcribe.py:148)
  pad_or_trim(x, 3000)                         -> zero pad / cut of the last axis (whisper.audio.pad_or_trim, :295)
  get_tokenizer(...)                           -> .eot / .timestamp_begin / .encode / .decode(ids) (text = the ids, spelled)
  DecodingOptions(**kwargs)                    -> a plain namespace (:212)
  model.decode(segment, options)               -> the SAME callable the oracle's transcribe() is given

so the two seek loops (clip handling :177-186, temperature fallback :193-233, no-speech skip :305-320, timestamp-driven
seek and segment cutting :348-408, the "instantaneous or empty" rule :494-499, ids :501-512) run on identical decode
results and must produce identical seeks, segments and token streams.

Two kinds of decode callables:
  * ``scripted_decode(seed)`` -- a deterministic function of (window start, window size, temperature) that emits token
    patterns from a small grammar (closed <|t|>..<|t'|> pairs, an open tail, a single closing timestamp, no timestamps,
    nothing at all) with random avg_logprob / no_speech_prob: every branch of the loop is reached within a few cases.
    The window start is read back from the mel itself (frame f of the padded mel carries f in band 0).
  * ``model_decode(sd, dims, ...)`` -- ``decode_oracle.decode`` on a real (tiny) model.
"""
import random
import types
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import decode_oracle as do

N_FRAMES = 3000


def index_mel(content_frames: int) -> torch.Tensor:
    """[80, content + 3000] 'mel' whose band 0 holds the frame index + 1 (0 = the zero padding of pad_or_trim)."""
    mel = torch.zeros(80, content_frames + N_FRAMES)
    mel[0] = torch.arange(1, content_frames + N_FRAMES + 1, dtype=torch.float32)
    return mel


def scripted_decode(seed: int) -> Callable:
    """decode(segment [80,3000], temperature, options-dict) -> decode_oracle.Result, a pure function of its arguments."""
    def decode(segment: torch.Tensor, temperature: float, kw: dict) -> do.Result:
        seek = int(segment[0, 0].item()) - 1
        size = int((segment[0] > 0).sum().item())
        rng = random.Random(hash((seed, seek, size, round(temperature * 10), bool(kw.get("beam_size")), bool(kw.get("best_of")))) & 0xFFFFFFFF)
        toks: List[int] = []
        kind = rng.choice(["pairs", "pairs", "pairs_open", "pairs_single_end", "single_ts", "no_ts", "empty", "only_ts", "zero_len"])
        t = rng.randrange(0, 40)
        text = lambda: [rng.randrange(0, 50256) for _ in range(rng.randrange(1, 5))]  # noqa: E731
        if kind in ("pairs", "pairs_open", "pairs_single_end"):
            for _ in range(rng.randrange(1, 4)):
                t2 = min(1500, t + rng.randrange(1, 500))
                toks += [do.TIMESTAMP_BEGIN + t] + text() + [do.TIMESTAMP_BEGIN + t2]
                t = min(1500, t2 + rng.randrange(0, 20))
            if kind == "pairs_open":
                toks += [do.TIMESTAMP_BEGIN + t] + text()
            elif kind == "pairs_single_end":
                toks += [do.TIMESTAMP_BEGIN + t] + text() + [do.TIMESTAMP_BEGIN + min(1500, t + rng.randrange(1, 99))]
                toks = toks[:-1] if rng.random() < 0.5 else toks
                toks += text() + [do.TIMESTAMP_BEGIN + min(1500, t + 120)]
        elif kind == "single_ts":
            toks = [do.TIMESTAMP_BEGIN + t] + text() + ([do.TIMESTAMP_BEGIN + t + rng.randrange(0, 3)] if rng.random() < 0.7 else [])
        elif kind == "no_ts":
            toks = text()
        elif kind == "only_ts":
            toks = [do.TIMESTAMP_BEGIN + t, do.TIMESTAMP_BEGIN + t + rng.randrange(0, 2)]
        elif kind == "zero_len":
            toks = [do.TIMESTAMP_BEGIN + t] + text() + [do.TIMESTAMP_BEGIN + t] + [do.TIMESTAMP_BEGIN + t + 7] + text() + [do.TIMESTAMP_BEGIN + t + 9]
        lp = rng.choice([-0.2, -0.6, -0.99, -1.0, -1.01, -1.4, -2.5]) + 0.3 * temperature
        return do.Result(tokens=toks, avg_logprob=lp, no_speech_prob=rng.choice([0.01, 0.3, 0.59, 0.6, 0.61, 0.9]),
                         temperature=temperature, sum_logprob=lp * (len(toks) + 1))
    return decode


class Tok:
    """Stand-in for whisper.tokenizer.Tokenizer on both sides of the comparison: a token spells itself."""
    eot, timestamp_begin = do.EOT, do.TIMESTAMP_BEGIN

    def encode(self, s):
        return [int(x) for x in s.split()]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def scripted_decode_cr(seed: int) -> Callable:
    """Like ``scripted_decode`` but with TEXT: the result carries text / compression_ratio the way whisper.decoding.DecodingTask.run
    fills them (tokenizer.decode of the tokens without timestamps, stripped; gzip ratio of that) and the grammar has repetitive
    outputs (ratio 3-6) that cool down as the temperature rises, blank segments (a token that decodes to nothing does not exist for
    this stand-in tokenizer, so "blank" = no token below eot) and confident-but-repetitive windows where ONLY the compression ratio
    triggers the fallback."""
    base = scripted_decode(seed)

    def decode(segment: torch.Tensor, temperature: float, kw: dict) -> do.Result:
        seek = int(segment[0, 0].item()) - 1
        size = int((segment[0] > 0).sum().item())
        rng = random.Random(hash((770001, seed, seek, size, round(temperature * 10))) & 0xFFFFFFFF)  # (ints only: str hashes are salted per process)
        r = base(segment, temperature, kw)
        mode = rng.choice(["plain", "plain", "loop", "loop_confident", "loop_forever"])
        toks = list(r.tokens)
        lp = r.avg_logprob
        if mode != "plain" and (mode == "loop_forever" or temperature < rng.choice([0.2, 0.4, 0.6])):
            # a repetition loop: the same 1-3 tokens over and over inside one closed pair (or bare text without timestamps)
            unit = [rng.randrange(100, 200) for _ in range(rng.randrange(1, 4))]
            body = unit * rng.randrange(12, 40)
            t0 = rng.randrange(0, 200)
            toks = body if kw.get("without_timestamps") else [do.TIMESTAMP_BEGIN + t0] + body + [do.TIMESTAMP_BEGIN + t0 + rng.randrange(1, 600)]
            if mode == "loop_confident":
                lp = -0.1  # the log-probability test alone would accept it
        text = Tok().decode([t for t in toks if t < do.TIMESTAMP_BEGIN]).strip()
        return do.Result(tokens=toks, avg_logprob=lp, no_speech_prob=r.no_speech_prob, temperature=temperature,
                         sum_logprob=lp * (len(toks) + 1), text=text, compression_ratio=do.compression_ratio(text))
    return decode


def scripted_words(seed: int) -> Callable:
    """A scripted stand-in for ``whisper.timing.add_word_timestamps`` (keyword signature of the call at olmoasr/transcribe.py:410-419),
    supplied to BOTH sides of the word-timestamp pin: a pure function of what it is handed -- the window (read back from the mel), the
    segment's seek / start / token count, ``num_frames`` and ``last_speech_timestamp`` (so a wrong value of any of them changes the
    words) -- that fills ``segment["words"]`` with plausible, anomalous (very short / very long / improbable), late-starting or no
    words, punctuation-only words included, and -- like the real function -- moves the segment's own start / end onto its words."""
    def add_word_timestamps(*, segments, model, tokenizer, mel, num_frames, prepend_punctuations, append_punctuations, last_speech_timestamp,
                            **kwargs):
        assert prepend_punctuations == "\"'“¿([{-" and append_punctuations == "\"'.。,，!！?？:：”)]}、"
        if len(segments) == 0:
            return
        window = int(mel[0, 0].item()) - 1
        assert window == segments[0]["seek"], (window, segments[0]["seek"])
        for si, seg in enumerate(segments):
            text_tokens = [t for t in seg["tokens"] if t < do.EOT]
            rng = random.Random(hash((880002, seed, window, si, len(text_tokens), round(seg["start"] * 100), int(num_frames),
                                      round(last_speech_timestamp * 100))) & 0xFFFFFFFF)
            kind = rng.choice(["normal", "normal", "normal", "anomalous", "none", "late", "late", "tail"])
            words = []
            if text_tokens and kind != "none":
                t = seg["start"] + (rng.choice([0.5, 3.0, 6.5]) if kind == "late" else rng.choice([0.0, 0.02, 0.1]))
                for tok in text_tokens[:12]:
                    if kind == "anomalous":
                        dur, prob = rng.choice([0.02, 0.05, 0.1, 2.5, 4.0]), rng.choice([0.05, 0.1, 0.5])
                    else:
                        dur, prob = rng.choice([0.14, 0.2, 0.3, 0.46, 0.8]), rng.choice([0.3, 0.6, 0.9, 0.99])
                    word = str(tok) if rng.random() > 0.15 else rng.choice([".", ",", "?"])
                    words.append(dict(word=word, start=round(t, 2), end=round(t + dur, 2), probability=prob))
                    t += dur + rng.choice([0.0, 0.0, 0.04, 0.3, 1.2])
                if kind == "tail":  # the last word runs far past the segment
                    words[-1]["end"] = round(words[-1]["end"] + rng.choice([2.0, 6.0]), 2)
                if rng.random() < 0.8:
                    seg["start"], seg["end"] = words[0]["start"], words[-1]["end"]
            seg["words"] = words
    return add_word_timestamps


def scripted_words_cases() -> List[dict]:
    """Cases of the fixture's "scripted_words" list: transcribe(word_timestamps=True) with and without hallucination_silence_threshold."""
    cases = []
    for seed in range(36):
        rng = random.Random(9000 + seed)
        kw: dict = dict(temperature=rng.choice([(0.0, 0.2, 0.4, 0.6, 0.8, 1.0), (0.0, 0.4), 0.0]), logprob_threshold=rng.choice([-1.0, -1.0, None]),
                        no_speech_threshold=rng.choice([0.6, None]), compression_ratio_threshold=rng.choice([2.4, None]), word_timestamps=True,
                        hallucination_silence_threshold=rng.choice([None, 0.5, 2.0, 2.0, 5.0]))
        content = rng.choice([2999, 3001, 9000, 12345, 20000])
        if rng.random() < 0.25:
            pts = sorted(rng.sample(range(0, content // 100), rng.choice([1, 2, 3])))
            kw["clip_timestamps"] = [float(p) for p in pts]
        if rng.random() < 0.15:
            kw["without_timestamps"] = True
        cases.append(dict(seed=seed, content_frames=content, kw=kw))
    return cases


def model_decode(sd, dims, logit_bias: Optional[torch.Tensor] = None) -> Callable:
    def decode(segment: torch.Tensor, temperature: float, kw: dict) -> do.Result:
        kw = {k: v for k, v in kw.items() if k in do.Options.__dataclass_fields__}
        return do.decode(sd, dims, segment[None], do.Options(temperature=temperature, logit_bias=logit_bias, **kw))[0]
    return decode


def run_oracle(decode: Callable, mel_padded: torch.Tensor, **kw) -> dict:
    """decode_oracle.transcribe with its model call replaced by ``decode`` (same replacement the reference side gets)."""
    saved = do.decode

    def shim(sd, dims, seg, opt):
        fields = {k: getattr(opt, k) for k in opt.__dataclass_fields__ if k not in ("temperature", "logit_bias")}
        passed = {k: v for k, v in fields.items() if k in shim.kw}  # only what transcribe() forwarded
        return [decode(seg[0], opt.temperature, passed)]
    tkw = {k: kw.pop(k) for k in list(kw) if k in ("temperature", "logprob_threshold", "no_speech_threshold", "clip_timestamps", "tokenizer",
                                                   "compression_ratio_threshold", "initial_prompt", "word_timestamps", "add_word_timestamps",
                                                   "hallucination_silence_threshold")}
    shim.kw = dict(kw)
    do.decode = shim
    try:
        return do.transcribe(None, None, mel_padded, **tkw, **kw)
    finally:
        do.decode = saved


def run_reference(decode: Callable, mel_padded: torch.Tensor, **kw) -> dict:
    """The reference's transcribe() itself (needs /root/reference)."""
    from . import ref_import
    rt = ref_import.load_transcribe()

    class Opt(types.SimpleNamespace):
        pass

    def ref_decode(segment, options):
        d = dict(vars(options))
        t = d.pop("temperature")
        d.pop("language", None), d.pop("fp16", None)
        r = decode(segment, t, d)
        cr = r.compression_ratio if r.compression_ratio == r.compression_ratio else 1.0  # (NaN: a token-level script)
        return types.SimpleNamespace(tokens=list(r.tokens), avg_logprob=r.avg_logprob, no_speech_prob=r.no_speech_prob,
                                     temperature=r.temperature, compression_ratio=cr, text=r.text)

    model = types.SimpleNamespace(dims=types.SimpleNamespace(n_mels=80, n_audio_ctx=1500, n_text_ctx=448), device=torch.device("cpu"),
                                  is_multilingual=False, num_languages=0, decode=ref_decode)
    rt.log_mel_spectrogram = lambda audio, n_mels=80, padding=0: audio
    rt.pad_or_trim = lambda x, length: F.pad(x, (0, length - x.shape[-1])) if x.shape[-1] < length else x[..., :length]
    rt.get_tokenizer = lambda *a, **k: Tok()
    rt.DecodingOptions = Opt
    words = kw.pop("add_word_timestamps", None)
    if words is not None:  # the two whisper names the word-timestamp section calls (:410, :422)
        rt.add_word_timestamps = words
        rt.get_end = do.get_end
    if "clip_timestamps" in kw:
        kw["clip_timestamps"] = list(kw["clip_timestamps"])
    kw.pop("tokenizer", None)  # (the reference builds its own through get_tokenizer, patched above)
    kw.setdefault("compression_ratio_threshold", None)
    out = rt.transcribe(model, mel_padded, verbose=None, **kw)
    toks = [t for s in out["segments"] for t in s["tokens"]]
    return {"segments": out["segments"], "tokens": toks, "text": out["text"]}


def comparable(out: dict, text: bool = False, words: bool = False) -> dict:
    """The fields both sides define, JSON-ready.  ``text``: also the text-level fields (runs with a tokenizer)."""
    keys = ("id", "seek", "start", "end", "tokens", "temperature", "avg_logprob", "no_speech_prob") + (("text", "compression_ratio") if text else ())
    res = {"tokens": [int(t) for t in out["tokens"]],
           "segments": [{**{k: ([int(t) for t in s[k]] if k == "tokens" else s[k]) for k in keys}, **({"words": s["words"]} if words else {})}
                        for s in out["segments"]]}
    if text:
        res["text"] = out["text"]
    return res


def scripted_cases() -> List[dict]:
    """(seed, content_frames, transcribe kwargs) of the committed fixture (tests/golden/transcribe_ref.json)."""
    cases = []
    for seed in range(40):
        rng = random.Random(1000 + seed)
        kw: dict = dict(temperature=rng.choice([0.0, (0.0, 0.2, 0.4), (0.0, 0.2, 0.4, 0.6, 0.8, 1.0)]),
                        logprob_threshold=rng.choice([-1.0, -1.0, None, -0.5]), no_speech_threshold=rng.choice([0.6, 0.6, None]))
        content = rng.choice([700, 2999, 3000, 3001, 9000, 12345, 20000])
        if rng.random() < 0.3:
            pts = sorted(rng.sample(range(0, content // 100), rng.choice([1, 2, 3, 4])))
            kw["clip_timestamps"] = [float(p) for p in pts]
        if rng.random() < 0.4:
            kw["beam_size"], kw["best_of"] = 5, 5
        if rng.random() < 0.3:
            kw["without_timestamps"] = True
        cases.append(dict(seed=seed, content_frames=content, kw=kw))
    return cases


def scripted_cr_cases() -> List[dict]:
    """Cases of the fixture's "scripted_cr" list: the same loop WITH a tokenizer -- compression-ratio fallback, texts, initial_prompt."""
    cases = []
    for seed in range(24):
        rng = random.Random(5000 + seed)
        kw: dict = dict(temperature=rng.choice([(0.0, 0.2, 0.4, 0.6, 0.8, 1.0), (0.0, 0.2, 0.4, 0.6, 0.8, 1.0), (0.0, 0.4), 0.0]),
                        logprob_threshold=rng.choice([-1.0, -1.0, None]), no_speech_threshold=rng.choice([0.6, None]),
                        compression_ratio_threshold=rng.choice([2.4, 2.4, 2.4, 1.2, None]))
        content = rng.choice([2999, 3001, 9000, 12345])
        if rng.random() < 0.25:
            kw["without_timestamps"] = True
        if rng.random() < 0.3:
            kw["initial_prompt"] = " ".join(str(rng.randrange(0, 50000)) for _ in range(rng.randrange(1, 6)))
        if rng.random() < 0.25:
            kw["beam_size"], kw["best_of"] = 5, 5
        cases.append(dict(seed=seed, content_frames=content, kw=kw))
    return cases
