"""Pins oracle/mel_oracle.py (CPU restatement of whisper.audio.log_mel_spectrogram) against
(1) fixtures produced by transformers.WhisperFeatureExtractor (independent implementation) and
(2) the same extractor run live, (3) torch.stft-based float32 pipeline as whisper writes it."""
import os

import numpy as np
import torch

from oracle import mel_oracle as me
from oracle import model_oracle as mo


def _clips():
    pcm, *_ = mo.synthetic_batch([0, 1])
    return pcm.numpy()


def test_mel_filters_match_hf():
    from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor
    f = me.mel_filters(80)
    assert f.shape == (80, 201) and f.dtype == np.float32
    assert np.abs(f - WhisperFeatureExtractor().mel_filters.T).max() < 1e-8


def test_oracle_vs_golden_hf(golden_dir):
    g = np.load(os.path.join(golden_dir, "mel_hf.npz"))
    m = me.log_mel_batch(_clips())
    assert m.shape == (2, 80, 3000) and m.dtype == np.float32
    tol = 5e-6  # fp32 rounding of HF's float32 pipeline vs our float64 internals (values in [-1.2, 1.5])
    assert np.abs(m[:, :, :96] - g["first"]).max() < tol
    assert np.abs(m[:, :, -96:] - g["last"]).max() < tol
    assert np.abs(m[:, :, ::37] - g["strided"]).max() < tol
    assert np.abs(m.astype(np.float64).sum(-1) - g["band_sum"]).max() < 3e-3
    assert np.abs(m.max((1, 2)) - g["vmax"]).max() < tol


def test_oracle_vs_torch_stft_pipeline():
    a = _clips()[0].astype(np.float32) / 32768.0
    st = torch.stft(torch.from_numpy(a), 400, 160, window=torch.hann_window(400), return_complex=True)
    mag = st[..., :-1].abs() ** 2
    x = torch.clamp(torch.from_numpy(me.mel_filters()) @ mag, min=1e-10).log10()
    x = (torch.maximum(x, x.max() - 8.0) + 4.0) / 4.0
    assert np.abs(x.numpy() - me.log_mel_spectrogram(a)).max() < 5e-6


def test_int16_equals_scaled_float():
    c = _clips()[1]
    assert np.array_equal(me.log_mel_spectrogram(c), me.log_mel_spectrogram(c.astype(np.float32) / 32768.0))


def test_pad_or_trim_edges():
    x = np.arange(10, dtype=np.float32)
    assert np.array_equal(me.pad_or_trim(x, 4), x[:4])
    p = me.pad_or_trim(x, 16)
    assert p.shape == (16,) and np.array_equal(p[:10], x) and not p[10:].any()
    assert me.pad_or_trim(np.zeros((0,), np.float32), 8).shape == (8,)
    y = np.ones((3, 5), np.float32)
    assert me.pad_or_trim(y, 7, axis=0).shape == (7, 5)
    assert me.pad_or_trim(x, 10) is x or np.array_equal(me.pad_or_trim(x, 10), x)


def test_silence_and_short_clip():
    z = np.zeros(480000, np.float32)
    m = me.log_mel_spectrogram(z)
    assert m.shape == (80, 3000) and np.allclose(m, (-10.0 + 4.0) / 4.0)
    s = me.log_mel_spectrogram(np.random.default_rng(0).standard_normal(1600).astype(np.float32))
    assert s.shape == (80, 10)
    assert me.log_mel_spectrogram(z[:1600], padding=480000).shape == (80, 3010)
