"""Pins the long-form driver against the reference's OWN ``olmoasr/transcribe.py:47-523`` (row a21).

* ``oracle.decode_oracle.transcribe`` == the unmodified reference file on 40 scripted-decode cases (every branch of the seek
  loop: clip_timestamps, temperature fallback with beam/best_of switching, no-speech skip, closed pairs / open tail / single
  closing timestamp / no timestamps / empty output, the instantaneous-or-empty rule, segment ids) and on a real tiny model
  with the timestamp bonus -- via the committed fixture tests/golden/transcribe_ref.json (oracle/gen_transcribe_golden.py)
  everywhere, and LIVE against /root/reference when it is mounted (build container).
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


def _eq(got, want, what):
    assert got["tokens"] == want["tokens"], what
    assert len(got["segments"]) == len(want["segments"]), what
    for a, b in zip(got["segments"], want["segments"]):
        for k in ("id", "seek", "tokens", "temperature"):
            assert a[k] == b[k], (what, k, a, b)
        for k in ("start", "end", "avg_logprob", "no_speech_prob"):
            assert abs(a[k] - b[k]) < 1e-9, (what, k, a, b)


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


def test_product_seek_loop_equals_reference_fixture(golden, monkeypatch):
    """olmoasr_amd/transcribe.py's host loop with its decode() replaced by the scripted one (batch of one window in timestamp
    mode; ``batch_windows`` windows ahead without timestamps -- results must not depend on it)."""
    from olmoasr_amd import transcribe as T
    model = types.SimpleNamespace(dims=types.SimpleNamespace(n_mels=80, n_audio_ctx=1500, n_text_ctx=448), device="cpu")
    for c, want in zip(H.scripted_cases(), golden["scripted"]):
        dec = H.scripted_decode(c["seed"])

        def fake_decode(_model, segments, options, dec=dec):
            kw = {k: getattr(options, k) for k in ("beam_size", "best_of", "without_timestamps", "patience")}
            out = []
            for seg in segments:
                r = dec(seg, options.temperature, kw)
                out.append(T.DecodingResult(audio_features=None, tokens=list(r.tokens), avg_logprob=r.avg_logprob,
                                            no_speech_prob=r.no_speech_prob, temperature=r.temperature))
            return out
        monkeypatch.setattr(T, "decode", fake_decode)
        monkeypatch.setattr(T, "log_mel_spectrogram", lambda audio, n_mels=80, padding=0, device=None: audio)
        kw = dict(c["kw"])
        if "clip_timestamps" in kw:
            kw["clip_timestamps"] = ",".join(str(x) for x in kw["clip_timestamps"])
        for bw in (1, 5):
            out = T.transcribe(model, H.index_mel(c["content_frames"]), compression_ratio_threshold=None, batch_windows=bw, **kw)
            _eq(H.comparable(out), want, f"product loop, case {c['seed']}, batch_windows {bw}")
