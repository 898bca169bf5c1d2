"""Pins the long-form driver against the reference's OWN ``olmoasr/transcribe.py:47-523`` (row a21).

* ``oracle.decode_oracle.transcribe`` == the unmodified reference file on 40 scripted-decode cases (every branch of the seek
  loop: clip_timestamps, temperature fallback with beam/best_of switching, no-speech skip, closed pairs / open tail / single
  closing timestamp / no timestamps / empty output, the instantaneous-or-empty rule, segment ids) and on a real tiny model
  with the timestamp bonus -- via the committed fixture tests/golden/transcribe_ref.json (oracle/gen_transcribe_golden.py)
  everywhere, and LIVE against /root/reference when it is mounted (build container).
* the same with a TOKENIZER (a stand-in whose tokens spell themselves, supplied to both sides): 24 scripted cases whose decode
  results carry text / compression_ratio like whisper.decoding fills them -- the compression-ratio fallback (:213-217, in half of
  the cases the only thing that changes the outcome), segment and result text, the blank-text rule (:494-499), initial_prompt
  (:258-264) -- oracle and product against the reference's own file (fixture + live).
* the PRODUCT's host-side seek loop (olmoasr_amd/transcribe.py) on the same scripted cases against the same fixture: its
  decode call is replaced by the scripted one, so this is pure host logic (no GPU, no native library)."""
import json
import os
import types

import pytest
import torch

from oracle import decode_oracle as do
from oracle import ref_import
from oracle import ref_transcribe_harness as H


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "transcribe_ref.json")) as f:
        return json.load(f)


def _eq(got, want, what, text=False, words=False):
    assert got["tokens"] == want["tokens"], what
    assert len(got["segments"]) == len(want["segments"]), what
    for a, b in zip(got["segments"], want["segments"]):
        if words:
            assert a["words"] == b["words"], (what, a["id"], a["words"][:3], b["words"][:3])
        for k in ("id", "seek", "tokens", "temperature") + (("text",) if text else ()):
            assert a[k] == b[k], (what, k, a, b)
        for k in ("start", "end", "avg_logprob", "no_speech_prob") + (("compression_ratio",) if text else ()):
            assert abs(a[k] - b[k]) < 1e-9, (what, k, a, b)
    if text:
        assert got["text"] == want["text"], what


def test_oracle_transcribe_equals_reference_fixture_scripted(golden):
    cases = H.scripted_cases()
    assert len(cases) == len(golden["scripted"]) == 40
    fallback = seek_driven = cleared = skipped = clipped = 0
    for c, want in zip(cases, golden["scripted"]):
        calls = []
        dec = H.scripted_decode(c["seed"])

        def logged(seg, t, kw, dec=dec, calls=calls):
            calls.append((int(seg[0, 0]) - 1, t))
            return dec(seg, t, kw)
        got = H.comparable(H.run_oracle(logged, H.index_mel(c["content_frames"]), **dict(c["kw"])))
        _eq(got, want, f"scripted case {c['seed']} {c['kw']}")
        fallback += any(s["temperature"] > 0 for s in got["segments"])
        seek_driven += any(s["seek"] % 3000 not in (0,) and "clip_timestamps" not in c["kw"] for s in got["segments"])
        cleared += any(s["tokens"] == [] for s in got["segments"])
        skipped += len({s for s, _ in calls}) > len({s["seek"] for s in got["segments"]})
        clipped += "clip_timestamps" in c["kw"]
    # the fixture reaches every branch of the loop
    assert fallback >= 5 and seek_driven >= 5 and cleared >= 5 and skipped >= 3 and clipped >= 5, (fallback, seek_driven, cleared, skipped, clipped)


def test_oracle_transcribe_with_tokenizer_equals_reference_fixture(golden):
    """Text level: compression-ratio fallback, texts, initial_prompt (a21's text half)."""
    cases = H.scripted_cr_cases()
    assert len(cases) == len(golden["scripted_cr"]) == 24
    cr_decides = prompts = fallback = 0
    for c, want in zip(cases, golden["scripted_cr"]):
        dec, mel = H.scripted_decode_cr(c["seed"]), H.index_mel(c["content_frames"])
        got = H.comparable(H.run_oracle(dec, mel, tokenizer=H.Tok(), **dict(c["kw"])), text=True)
        _eq(got, want, f"scripted_cr case {c['seed']} {c['kw']}", text=True)
        no_cr = H.comparable(H.run_oracle(dec, mel, tokenizer=H.Tok(), **{**c["kw"], "compression_ratio_threshold": None}), text=True)
        cr_decides += no_cr != got
        prompts += "initial_prompt" in c["kw"]
        fallback += any(s["temperature"] > 0 for s in got["segments"])
    assert cr_decides >= 8 and prompts >= 3 and fallback >= 8, (cr_decides, prompts, fallback)


def test_oracle_transcribe_word_timestamps_equals_reference_fixture(golden):
    """transcribe(word_timestamps=True[, hallucination_silence_threshold=...]) (:409-486): the seek-to-last-word rule, the three silence /
    hallucination skipping rules, ``last_speech_timestamp`` and the ``words`` of cleared segments, with ``add_word_timestamps`` (third party,
    whisper.timing) scripted identically on both sides."""
    cases = H.scripted_words_cases()
    assert len(cases) == len(golden["scripted_words"]) == 36
    hal_decides = word_seek = with_words = dropped = 0
    for c, want in zip(cases, golden["scripted_words"]):
        dec, mel, words = H.scripted_decode_cr(c["seed"]), H.index_mel(c["content_frames"]), H.scripted_words(c["seed"])
        got = H.comparable(H.run_oracle(dec, mel, tokenizer=H.Tok(), add_word_timestamps=words, **dict(c["kw"])), text=True, words=True)
        _eq(got, want, f"scripted_words case {c['seed']} {c['kw']}", text=True, words=True)
        plain = H.comparable(H.run_oracle(dec, mel, tokenizer=H.Tok(), **{**c["kw"], "word_timestamps": False, "hallucination_silence_threshold": None}), text=True)
        word_seek += [s["seek"] for s in plain["segments"]] != [s["seek"] for s in got["segments"]]
        if c["kw"]["hallucination_silence_threshold"] is not None:
            no_hal = H.comparable(H.run_oracle(dec, mel, tokenizer=H.Tok(), add_word_timestamps=words, **{**c["kw"], "hallucination_silence_threshold": None}),
                                  text=True, words=True)
            hal_decides += no_hal != got
            dropped += len(no_hal["segments"]) != len(got["segments"])
        with_words += any(s["words"] for s in got["segments"])
    assert hal_decides >= 8 and word_seek >= 15 and with_words >= 25 and dropped >= 5, (hal_decides, word_seek, with_words, dropped)


def test_oracle_transcribe_equals_reference_fixture_real_model(golden):
    from oracle.gen_transcribe_golden import model_case
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sd, dims, mel_padded, bias, kw = model_case()
    got = do.transcribe(sd, dims, mel_padded, logit_bias=bias, **kw)
    _eq(H.comparable(got), golden["model"], "tiny model, timestamp bonus")
    assert len(got["seeks"]) >= 2 and any(s % 3000 for s in got["seeks"])  # the seek was driven by timestamps


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not mounted (build container only)")
def test_oracle_transcribe_equals_reference_live():
    for c in H.scripted_cases() + [dict(seed=99, content_frames=45000, kw=dict(temperature=(0.0, 0.4, 0.8), logprob_threshold=-0.8,
                                                                               no_speech_threshold=0.5, clip_timestamps=[3.0, 200.0, 210.5]))]:
        dec, mel = H.scripted_decode(c["seed"]), H.index_mel(c["content_frames"])
        _eq(H.comparable(H.run_oracle(dec, mel, **dict(c["kw"]))), H.comparable(H.run_reference(dec, mel, **dict(c["kw"]))), f"live {c}")
    for c in H.scripted_cr_cases():
        dec, mel = H.scripted_decode_cr(c["seed"]), H.index_mel(c["content_frames"])
        _eq(H.comparable(H.run_oracle(dec, mel, tokenizer=H.Tok(), **dict(c["kw"])), text=True),
            H.comparable(H.run_reference(dec, mel, **dict(c["kw"])), text=True), f"live, with tokenizer {c}", text=True)
    for c in H.scripted_words_cases():
        dec, mel, words = H.scripted_decode_cr(c["seed"]), H.index_mel(c["content_frames"]), H.scripted_words(c["seed"])
        _eq(H.comparable(H.run_oracle(dec, mel, tokenizer=H.Tok(), add_word_timestamps=words, **dict(c["kw"])), text=True, words=True),
            H.comparable(H.run_reference(dec, mel, add_word_timestamps=words, **dict(c["kw"])), text=True, words=True),
            f"live, word timestamps {c}", text=True, words=True)


def test_product_seek_loop_equals_reference_fixture(golden, monkeypatch):
    """olmoasr_amd/transcribe.py's host loop with its decode() replaced by the scripted one (batch of one window in timestamp
    mode; ``batch_windows`` windows ahead without timestamps -- results must not depend on it)."""
    from olmoasr_amd import transcribe as T
    model = types.SimpleNamespace(dims=types.SimpleNamespace(n_mels=80, n_audio_ctx=1500, n_text_ctx=448), device="cpu")
    for c, want in zip(H.scripted_cases(), golden["scripted"]):
        dec = H.scripted_decode(c["seed"])
        monkeypatch.setattr(T, "decode", _fake_decode(T, dec))
        monkeypatch.setattr(T, "log_mel_spectrogram", lambda audio, n_mels=80, padding=0, device=None: audio)
        monkeypatch.setattr(T, "resolve_tokenizer", lambda model, tokenizer=None, *a, **k: tokenizer)  # (no whisper package lookup)
        kw = dict(c["kw"])
        if "clip_timestamps" in kw:
            kw["clip_timestamps"] = ",".join(str(x) for x in kw["clip_timestamps"])
        for bw in (1, 5):
            out = T.transcribe(model, H.index_mel(c["content_frames"]), compression_ratio_threshold=None, batch_windows=bw, **kw)
            assert out["text"] is None and all(s["text"] is None for s in out["segments"])  # token level: no tokenizer, no text
            _eq(H.comparable(out), want, f"product loop, case {c['seed']}, batch_windows {bw}")


def _fake_decode(T, dec):
    def fake_decode(_model, segments, options, tokenizer=None):
        kw = {k: getattr(options, k) for k in ("beam_size", "best_of", "without_timestamps", "patience")}
        out = []
        for seg in segments:
            r = dec(seg, options.temperature, kw)
            out.append(T.DecodingResult(audio_features=None, tokens=list(r.tokens), avg_logprob=r.avg_logprob, no_speech_prob=r.no_speech_prob,
                                        temperature=r.temperature, text=r.text, compression_ratio=r.compression_ratio))
        return out
    return fake_decode


def test_product_seek_loop_with_tokenizer_equals_reference_fixture(golden, monkeypatch):
    """olmoasr_amd/transcribe.py with the tokenizer PLUG: compression-ratio fallback, texts, the blank-text rule and initial_prompt
    against what the reference's own transcribe() produced on the same decode results (fixture "scripted_cr")."""
    from olmoasr_amd import transcribe as T
    model = types.SimpleNamespace(dims=types.SimpleNamespace(n_mels=80, n_audio_ctx=1500, n_text_ctx=448), device="cpu")
    monkeypatch.setattr(T, "log_mel_spectrogram", lambda audio, n_mels=80, padding=0, device=None: audio)
    for c, want in zip(H.scripted_cr_cases(), golden["scripted_cr"]):
        monkeypatch.setattr(T, "decode", _fake_decode(T, H.scripted_decode_cr(c["seed"])))
        for bw in (1, 4):
            out = T.transcribe(model, H.index_mel(c["content_frames"]), batch_windows=bw, tokenizer=H.Tok(), **dict(c["kw"]))
            _eq(H.comparable(out, text=True), want, f"product loop with tokenizer, case {c['seed']}, batch_windows {bw}", text=True)


def test_product_seek_loop_word_timestamps_equals_reference_fixture(golden, monkeypatch):
    """olmoasr_amd/transcribe.py with word_timestamps / hallucination_silence_threshold against what the reference's own transcribe()
    produced with the same scripted decode results and the same scripted add_word_timestamps (fixture "scripted_words")."""
    from olmoasr_amd import transcribe as T
    model = types.SimpleNamespace(dims=types.SimpleNamespace(n_mels=80, n_audio_ctx=1500, n_text_ctx=448), device="cpu")
    monkeypatch.setattr(T, "log_mel_spectrogram", lambda audio, n_mels=80, padding=0, device=None: audio)
    for c, want in zip(H.scripted_words_cases(), golden["scripted_words"]):
        monkeypatch.setattr(T, "decode", _fake_decode(T, H.scripted_decode_cr(c["seed"])))
        monkeypatch.setattr(T, "add_word_timestamps", H.scripted_words(c["seed"]))
        kw = dict(c["kw"])
        if "clip_timestamps" in kw:
            kw["clip_timestamps"] = ",".join(str(x) for x in kw["clip_timestamps"])
        out = T.transcribe(model, H.index_mel(c["content_frames"]), batch_windows=4, tokenizer=H.Tok(), **kw)
        _eq(H.comparable(out, text=True, words=True), want, f"product loop, word timestamps, case {c['seed']}", text=True, words=True)
    with pytest.raises(ValueError, match="tokenizer"):
        monkeypatch.setattr(T, "resolve_tokenizer", lambda model, tokenizer=None, *a, **k: tokenizer)
        T.transcribe(model, H.index_mel(3000), word_timestamps=True)


def test_decode_fills_text_and_compression_ratio_like_whisper():
    """olmoasr_amd.decoding: compression_ratio == the oracle's restatement of whisper.utils.compression_ratio; resolve_tokenizer
    returns the plug, or None when whisper is not installed."""
    from olmoasr_amd import decoding as D
    for text in ("", "a", "hello hello hello hello hello hello hello hello hello hello", "The quick brown fox."):
        if text:
            assert D.compression_ratio(text) == do.compression_ratio(text)
    assert D.compression_ratio("ab " * 200) > 2.4 > D.compression_ratio("The quick brown fox jumps over the lazy dog.")
    tok = H.Tok()
    assert D.resolve_tokenizer(None, tok) is tok
    try:
        import whisper  # noqa: F401
    except Exception:
        assert D.resolve_tokenizer(types.SimpleNamespace(is_multilingual=False, num_languages=0)) is None
