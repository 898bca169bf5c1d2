"""olmoasr_amd/timing.py (word-level timestamps = what olmoasr/transcribe.py:409-419 calls as whisper.timing.add_word_timestamps) -- host logic.

whisper is not installable here, so the published algorithm is checked through independent statements of what each piece computes: the DTW
against a brute-force minimum over ALL monotonic paths, the median filter against scipy's, punctuation merging and the segment / word
clipping rules on hand-built alignments.  The GPU half (cross-attention scores, find_alignment on a model) is tests/test_gpu_timing.py."""
import itertools

import numpy as np
import pytest
import torch

from olmoasr_amd import timing as T


def _all_paths(n, m):
    """every monotonic path (0,0) -> (n-1,m-1) with steps (1,1), (1,0), (0,1)"""
    def rec(i, j):
        if (i, j) == (n - 1, m - 1):
            yield [(i, j)]
            return
        for di, dj in ((1, 1), (1, 0), (0, 1)):
            if i + di < n and j + dj < m:
                for rest in rec(i + di, j + dj):
                    yield [(i, j)] + rest
    return rec(0, 0)


def test_dtw_path_is_a_minimum_cost_monotonic_path():
    rng = np.random.default_rng(0)
    for n, m in ((1, 1), (1, 5), (4, 1), (3, 4), (5, 5), (4, 7), (6, 3)):
        x = rng.normal(size=(n, m)).astype(np.float32)
        ti, fi = T.dtw(x)
        path = list(zip(ti.tolist(), fi.tolist()))
        assert path[0] == (0, 0) and path[-1] == (n - 1, m - 1)
        assert all((b[0] - a[0], b[1] - a[1]) in ((1, 1), (1, 0), (0, 1)) for a, b in zip(path, path[1:]))
        best = min(sum(float(x[i, j]) for i, j in p) for p in _all_paths(n, m))
        assert abs(sum(float(x[i, j]) for i, j in path) - best) < 1e-5, (n, m)


def test_dtw_tie_rule_and_a_diagonal_ridge():
    # equal costs everywhere: the trace prefers "right" (same token, next frame) unless another step is STRICTLY cheaper, so the backtrace
    # from the end walks left along the last token first and the path hugs the first column / last row
    ti, fi = T.dtw(np.zeros((3, 4), dtype=np.float32))
    assert list(zip(ti.tolist(), fi.tolist())) == [(0, 0), (1, 0), (2, 0), (2, 1), (2, 2), (2, 3)]
    # a strong diagonal (token k speaks during frames 3k .. 3k+2) is followed exactly
    x = np.ones((4, 12), dtype=np.float32)
    for k in range(4):
        x[k, 3 * k:3 * k + 3] = -1.0
    ti, fi = T.dtw(x)
    assert all(ti[fi == f][0] == f // 3 for f in range(12))
    jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
    assert fi[jumps].tolist() == [0, 3, 6, 9]  # the frame at which each token starts


def test_median_filter_matches_scipy_mirror_mode():
    from scipy.ndimage import median_filter as sp
    g = torch.Generator().manual_seed(0)
    for shape, w in (((3, 5, 40), 7), ((2, 9), 3), ((17,), 5), ((4, 6, 8), 1)):
        x = torch.randn(*shape, generator=g)
        size = [1] * (x.dim() - 1) + [w]
        assert np.allclose(T.median_filter(x, w).numpy(), sp(x.numpy(), size=size, mode="mirror"), atol=0)
    short = torch.randn(2, 3)
    assert torch.equal(T.median_filter(short, 7), short)  # shorter than the padding: returned as is
    with pytest.raises(AssertionError):
        T.median_filter(torch.randn(10), 4)


def _w(word, start, end, tokens=None, p=0.9):
    return T.WordTiming(word, list(tokens if tokens is not None else [1]), start, end, p)


def test_merge_punctuations():
    a = [_w(" (", 0, 1, [1]), _w("hello", 1, 2, [2]), _w(",", 2, 3, [3]), _w(" world", 3, 4, [4]), _w(".", 4, 5, [5]), _w(")", 5, 6, [6]),
         _w(" \"", 6, 7, [7]), _w(" '", 7, 8, [8]), _w("x", 8, 9, [9])]
    T.merge_punctuations(a, "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、")
    assert [(t.word, t.tokens) for t in a] == [("", []), (" (hello,", [1, 2, 3]), ("", []), (" world.)", [4, 5, 6]), ("", []), ("", []),
                                               ("", []), ("", []), (" \" 'x", [7, 8, 9])]


class _Tok:
    eot, sot_sequence, no_timestamps = 50256, (50257,), 50362


def test_add_word_timestamps_fills_words_and_moves_segment_bounds(monkeypatch):
    segs = [dict(seek=3000, start=30.0, end=34.0, tokens=[50363, 11, 12, 13, 50500]),
            dict(seek=3000, start=34.0, end=40.0, tokens=[50500, 21, 22, 50800]),
            dict(seek=3000, start=40.0, end=41.0, tokens=[50800, 50801])]  # no text tokens: no words

    def fake_alignment(model, tokenizer, text_tokens, mel, num_frames, **kw):
        assert text_tokens == [11, 12, 13, 21, 22] and num_frames == 2500
        return [_w(" a", 0.5, 0.9, [11]), _w(" b", 0.9, 1.3, [12]), _w(".", 1.3, 1.5, [13]), _w(" c", 4.5, 4.9, [21]), _w(" d", 4.9, 9.9, [22], p=0.2)]
    monkeypatch.setattr(T, "find_alignment", fake_alignment)
    T.add_word_timestamps(segments=segs, model=None, tokenizer=_Tok(), mel=None, num_frames=2500, last_speech_timestamp=29.0)
    # window offset 30 s.  "." is merged into " b" (word and tokens; a merged word keeps its own times).  Median word 0.4 s -> max 0.8 s.
    assert [w["word"] for w in segs[0]["words"]] == [" a", " b."] and segs[0]["words"][0]["start"] == 30.5 and segs[0]["words"][1]["end"] == 31.3
    assert segs[0]["start"] == 30.5 and segs[0]["end"] == 31.3  # the segment's bounds follow its words
    # second segment: its first word ends 3.6 s after the last speech (> 4 medians) and the second word is 5 s long: the boundary between them
    # moves to (end of the second word - max) = 39.1 and the first word becomes max long
    w = segs[1]["words"]
    assert [x["word"] for x in w] == [" c", " d"] and w[1]["probability"] == 0.2
    assert abs(w[0]["start"] - 38.3) < 1e-9 and abs(w[0]["end"] - 39.1) < 1e-9 and abs(w[1]["start"] - 39.1) < 1e-9 and w[1]["end"] == 39.9
    assert segs[1]["start"] == w[0]["start"] and segs[1]["end"] == 39.9
    assert segs[2]["words"] == []
    # nothing to do
    T.add_word_timestamps(segments=[], model=None, tokenizer=_Tok(), mel=None, num_frames=1, last_speech_timestamp=0.0)


def test_add_word_timestamps_clips_a_long_first_word_after_a_pause(monkeypatch):
    segs = [dict(seek=0, start=10.0, end=14.0, tokens=[1, 2, 3])]

    def fake_alignment(model, tokenizer, text_tokens, mel, num_frames, **kw):
        return [_w(" long", 2.0, 10.4, [1]), _w(" b", 10.4, 10.8, [2]), _w(" c", 10.8, 11.2, [3])]
    monkeypatch.setattr(T, "find_alignment", fake_alignment)
    T.add_word_timestamps(segments=segs, model=None, tokenizer=_Tok(), mel=None, num_frames=3000, last_speech_timestamp=0.0)
    w = segs[0]["words"]
    # median duration 0.4 (0.4, 0.4, 8.4) -> max 0.8; first word ends 10.4 s after the last speech (> 4 medians) and is longer than max: start = end - max
    assert w[0]["end"] == 10.4 and abs(w[0]["start"] - 9.6) < 1e-9 and segs[0]["start"] == w[0]["start"] and segs[0]["end"] == 11.2


def test_alignment_heads_default_is_the_upper_half_of_the_decoder():
    import types
    m = types.SimpleNamespace(dims=types.SimpleNamespace(n_text_layer=4, n_text_head=3))
    assert T.alignment_heads(m) == [(l, h) for l in (2, 3) for h in range(3)]
    mask = torch.zeros(4, 3, dtype=torch.bool)
    mask[1, 2] = mask[3, 0] = True
    m.alignment_heads = mask.to_sparse()
    assert T.alignment_heads(m) == [(1, 2), (3, 0)]


def test_set_alignment_heads_accepts_a_mask_or_whispers_serialised_form():
    import base64
    import gzip
    import types
    from olmoasr_amd.model import OLMoASR
    m = types.SimpleNamespace(dims=types.SimpleNamespace(n_text_layer=4, n_text_head=6), alignment_heads=None)
    mask = np.zeros((4, 6), dtype=bool)
    mask[2, 1] = mask[3, 5] = True
    OLMoASR.set_alignment_heads(m, mask)
    assert T.alignment_heads(m) == [(2, 1), (3, 5)]
    OLMoASR.set_alignment_heads(m, base64.b85encode(gzip.compress(mask.tobytes())))
    assert T.alignment_heads(m) == [(2, 1), (3, 5)]
