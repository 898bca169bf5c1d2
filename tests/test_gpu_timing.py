"""GPU half of rows a8 / a21: the attention score matrix on request (csrc/scores.hip, ``oasr_attention_scores``), ``qk`` out of
MultiHeadAttention.forward like the reference returns it (olmoasr/model.py:313-345), and word-level timestamps on top of it
(olmoasr_amd/timing.py; the reference calls whisper.timing.add_word_timestamps at olmoasr/transcribe.py:410-419)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_scores(q, k, kv_len, causal):
    s = torch.einsum("bihc,bjhc->bhij", q.double(), k.double()) * 0.125
    Tq, Tk = s.shape[-2:]
    if causal:
        s = s.masked_fill(torch.ones(Tq, Tk, dtype=torch.bool, device=s.device).triu(1), -math.inf)
    if kv_len is not None:
        pad = torch.arange(Tk, device=s.device)[None, :] >= kv_len[:, None].to(s.device)
        s = s.masked_fill(pad[:, None, None, :], -math.inf)
    return s


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,Tq,Tk,causal,ragged", [(2, 6, 7, 1500, False, False), (3, 4, 130, 130, True, False), (2, 8, 65, 200, False, True),
                                                     (1, 16, 448, 1500, False, False), (2, 2, 1, 1, True, False)])
def test_attention_scores_match_a_float64_product(dtype, B, H, Tq, Tk, causal, ragged):
    from olmoasr_amd import ops
    g = torch.Generator().manual_seed(Tq * 31 + Tk)
    # operands inside a fused q|k|v projection output (token stride 3 H 64), as the engine and MultiHeadAttention.forward hold them
    qkv = torch.randn(B, max(Tq, Tk), 3 * H * 64, generator=g).to(DEV, dtype)
    q = qkv[:, :Tq, : H * 64].view(B, Tq, H, 64)
    k = qkv[:, :Tk, H * 64: 2 * H * 64].view(B, Tk, H, 64)
    kv_len = torch.tensor([Tk - 3 * b - 1 for b in range(B)], dtype=torch.int32, device=DEV) if ragged else None
    got = ops.attention_scores(q, k, kv_len, causal)
    want = _ref_scores(q, k, kv_len, causal)
    assert got.dtype == torch.float32 and got.shape == (B, H, Tq, Tk)
    assert torch.equal(torch.isinf(got), torch.isinf(want))
    fin = torch.isfinite(want)
    assert float((got.double() - want)[fin].abs().max()) < 1e-4 * (1 + float(want[fin].abs().max()))  # fp32 accumulation of exact products


def _dims(mo_dims):
    from olmoasr_amd.config.model_dims import ModelDimensions
    return ModelDimensions(**{k: getattr(mo_dims, k) for k in ModelDimensions.__dataclass_fields__})


def test_multi_head_attention_returns_qk_like_the_reference(tiny_case):
    """2-D mask (the reference's manual path) -> fp32 pre-softmax scores with the mask added; mask-free / 3-D mask (its SDPA path) -> None,
    unless ``return_qk`` asks.  Values against the oracle's qkv_attention (pinned bit-exactly to the reference's module on the CPU)."""
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims, sd = tiny_case["dims"], tiny_case["sd"]
    net = OLMoASR(_dims(dims), device=DEV, seed=0)
    net.load_state_dict(sd)
    blk = net.decoder.blocks[1]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 9, dims.n_text_state, generator=g)
    xa = torch.randn(2, 1500, dims.n_text_state, generator=g)
    mask = torch.full((9, 9), -math.inf).triu_(1)
    out, qk = blk.attn(x.to(DEV), mask=mask.to(DEV))
    cfg = mo._Cfg(True)
    pre = "decoder.blocks.1.attn"
    q = mo.linear(x, sd[pre + ".query.weight"], sd[pre + ".query.bias"], cfg)
    k = mo.linear(x, sd[pre + ".key.weight"], None, cfg)
    v = mo.linear(x, sd[pre + ".value.weight"], sd[pre + ".value.bias"], cfg)
    _, qk_o = mo.qkv_attention(q.float(), k.float(), v.float(), dims.n_text_head, mask)
    assert qk.dtype == torch.float32 and qk.shape == (2, dims.n_text_head, 9, 9) and torch.equal(torch.isinf(qk.cpu()), torch.isinf(qk_o))
    fin = torch.isfinite(qk_o)
    assert float((qk.cpu() - qk_o)[fin].abs().max()) < 2e-2 * (1 + float(qk_o[fin].abs().max()))  # bf16 projections on both sides, fp32 scores
    assert blk.attn(x.to(DEV))[1] is None and blk.cross_attn(x.to(DEV), xa.to(DEV))[1] is None
    assert blk.attn(x.to(DEV), mask=torch.zeros(2, 9, 9, device=DEV))[1] is None  # 3-D padding mask: the reference's SDPA path
    blk.cross_attn.return_qk = True
    try:
        _, cqk = blk.cross_attn(x.to(DEV), xa.to(DEV))
    finally:
        del blk.cross_attn.return_qk
    assert cqk.shape == (2, dims.n_text_head, 9, 1500) and torch.isfinite(cqk).all() and blk.cross_attn(x.to(DEV), xa.to(DEV))[1] is None


class WordTok:
    """Scripted tokenizer with whisper's attribute names: every text token is one word."""
    eot, sot_sequence, no_timestamps, timestamp_begin = 50256, (50257,), 50362, 50363

    def decode(self, ids):
        return "".join(f" w{int(i)}" for i in ids if i < self.eot)

    def encode(self, s):
        return [int(x[1:]) for x in s.split()]

    def split_to_word_tokens(self, tokens):
        return [f" w{t}" if t < self.eot else "<|eot|>" for t in tokens], [[t] for t in tokens]


def test_cross_attention_scores_and_alignment_against_the_oracle(tiny_case):
    from olmoasr_amd import timing
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims, sd = tiny_case["dims"], tiny_case["sd"]
    net = OLMoASR(_dims(dims), device=DEV, seed=0)
    net.load_state_dict(sd)
    mel = tiny_case["mel"][:1].to(DEV)
    text = [1000, 2000, 3000, 4000, 5000]
    toks = torch.tensor([50257, 50362, *text, 50256], device=DEV)
    xa = net.embed_audio(mel)
    layers = sorted({l for l, _ in timing.alignment_heads(net)})
    assert layers == list(range(dims.n_text_layer // 2, dims.n_text_layer))
    got = timing.cross_attention_scores(net, toks, xa, layers)
    want = mo.cross_attention_scores(sd, dims, toks.cpu()[None], mo.encoder_forward(sd, dims, tiny_case["mel"][:1], True).float(), layers, True)
    for l in layers:
        a, b = got[l].cpu(), want[l][0]
        assert a.shape == b.shape == (dims.n_text_head, 8, 1500)
        assert float((a - b).abs().max()) < 0.05 * (1 + float(b.abs().max())), (l, float((a - b).abs().max()), float(b.abs().max()))
    assert not any(hasattr(b.cross_attn, "__dict__") and "return_qk" in b.cross_attn.__dict__ for b in net.decoder.blocks)  # switches restored
    # the alignment itself: one WordTiming per word, monotone, inside the window, probabilities in (0, 1]
    words = timing.find_alignment(net, WordTok(), text, mel[0], 3000)
    assert [w.word for w in words] == [f" w{t}" for t in text] and [w.tokens for w in words] == [[t] for t in text]  # (the eot "word" has no end time)
    times = [(w.start, w.end) for w in words]
    assert all(0.0 <= s <= e <= 30.0 for s, e in times) and all(a[1] <= b[0] + 1e-9 for a, b in zip(times, times[1:]))
    assert all(0.0 < w.probability <= 1.0 for w in words)
    assert timing.find_alignment(net, WordTok(), [], mel[0], 3000) == []


def test_transcribe_with_word_timestamps_runs_end_to_end(tiny_case):
    """transcribe(word_timestamps=True) on the native model: every text-bearing segment gets ``words`` that tile its tokens, with times
    inside the audio; the same call without a tokenizer is refused (words are a property of the text)."""
    from olmoasr_amd.model import OLMoASR
    dims, sd = tiny_case["dims"], tiny_case["sd"]
    net = OLMoASR(_dims(dims), device=DEV, seed=0)
    net.load_state_dict(sd)
    pcm = tiny_case["pcm"][0].float() / 32768.0
    kw = dict(tokenizer=WordTok(), word_timestamps=True, temperature=0.0, logprob_threshold=None, no_speech_threshold=None,
              compression_ratio_threshold=None, sample_len=12)
    # (a random-weight model's words are improbable: with the hallucination rules on, its segments may all be dropped -- only that it runs)
    hal = net.transcribe(pcm, hallucination_silence_threshold=2.0, **kw)
    assert isinstance(hal["segments"], list) and all("words" in s for s in hal["segments"])
    out = net.transcribe(pcm, **kw)
    assert out["segments"]
    seen = 0
    for s in out["segments"]:
        assert "words" in s
        for w in s["words"]:
            assert set(w) == {"word", "start", "end", "probability"} and 0.0 <= w["start"] <= w["end"] <= 31.0
            seen += 1
    assert seen > 0
    with pytest.raises(ValueError, match="tokenizer"):
        net.transcribe(pcm, word_timestamps=True)
