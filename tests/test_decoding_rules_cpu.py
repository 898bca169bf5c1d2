"""CPU tests of the token-level logit filters of olmoasr_amd.decoding (no device needed): ApplyTimestampRules as
whisper.decoding defines them, with the English-only ids the reference's Dataset writes."""
import pytest
import torch

from olmoasr_amd.decoding import EOT, NO_TIMESTAMPS, SOT, TIMESTAMP_BEGIN, _timestamp_rules

V = 51864
NEG = -float("inf")


def _rules(prefix, logits=None, sample_begin=1, max_initial_index=50):
    toks = torch.tensor([prefix], dtype=torch.int64)
    lg = torch.zeros(1, V) if logits is None else logits.clone()
    _timestamp_rules(lg, toks, sample_begin, max_initial_index)
    return lg[0]


def test_first_token_must_be_an_early_timestamp():
    lg = _rules([SOT])
    assert torch.isinf(lg[:TIMESTAMP_BEGIN]).all()                      # no text, no eot, no <|notimestamps|>
    assert torch.isfinite(lg[TIMESTAMP_BEGIN:TIMESTAMP_BEGIN + 51]).all()  # <|0.00|> .. <|1.00|>
    assert torch.isinf(lg[TIMESTAMP_BEGIN + 51:]).all()


def test_timestamps_come_in_pairs_and_do_not_decrease():
    t = TIMESTAMP_BEGIN
    lg = torch.zeros(1, V)
    lg[0, 100] = 20.0  # a text token clearly beats the total timestamp mass (logsumexp of ~1500 ids at logit 0 = 7.3)
    # after "<|0.20|> text": anything but a timestamp earlier than the last one (plus <|notimestamps|>)
    out = _rules([SOT, t + 10, 100], lg)
    assert torch.isinf(out[t:t + 11]).all() and torch.isfinite(out[t + 11:]).all()
    assert torch.isfinite(out[:EOT + 1]).all() and out[NO_TIMESTAMPS] == NEG
    # after "text <|0.40|>" (an opening/closing single timestamp): no text tokens, a timestamp >= the last or eot
    lg2 = lg.clone()
    lg2[0, EOT] = 20.0  # (eot counts as a "text" token in the mass test: keep it above the timestamp mass)
    out = _rules([SOT, t + 10, 100, t + 20], lg2)
    assert torch.isinf(out[:EOT]).all() and torch.isfinite(out[EOT]) and torch.isinf(out[t:t + 20]).all()
    assert torch.isfinite(out[t + 20:]).all()
    # after a timestamp PAIR: the next token has to be text (or eot), never a third timestamp
    out = _rules([SOT, t + 10, 100, t + 20, t + 20], lg)
    assert torch.isinf(out[t:]).all() and torch.isfinite(out[100])


def test_timestamp_mass_forces_a_timestamp():
    t = TIMESTAMP_BEGIN
    lg = torch.full((1, V), -10.0)
    lg[0, 100] = 0.0          # best text token
    lg[0, t + 30:t + 40] = -1.0  # ten timestamps, each less likely than it, together more likely
    out = _rules([SOT, t + 10, 100], lg)
    assert torch.isinf(out[:t]).all() and torch.isfinite(out[t + 30])
    lg[0, t + 30:t + 40] = -8.0  # now their mass does not beat the text token
    out = _rules([SOT, t + 10, 100], lg)
    assert torch.isfinite(out[100])


def test_product_rules_equal_the_oracle_restatement_on_random_states():
    """olmoasr_amd.decoding._timestamp_rules (vectorised) against oracle.decode_oracle.apply_timestamp_rules (row by row)
    on random logits and random well-formed / ill-formed prefixes: identical masks, bit-identical surviving logits."""
    from oracle import decode_oracle as do
    g = torch.Generator().manual_seed(0)
    t = TIMESTAMP_BEGIN
    prefixes = [[SOT], [SOT, t + 3], [SOT, t + 3, 11, 12], [SOT, t + 3, 11, t + 40], [SOT, t + 3, 11, t + 40, t + 40],
                [SOT, t + 3, 11, t + 40, t + 40, 99], [SOT, 5, 6, 7], [SOT, t, t], [SOT, t + 1499]]
    for pre in prefixes:
        for scale in (0.5, 4.0):
            lg = torch.randn(3, V, generator=g) * scale
            lg[1, t:] += 6.0  # one row where the timestamp mass wins
            toks = torch.tensor([pre] * 3, dtype=torch.int64)
            a, b = lg.clone(), lg.clone()
            _timestamp_rules(a, toks, 1, 50)
            do.apply_timestamp_rules(b, toks, 1, 50)
            assert torch.equal(a, b), pre


def test_suppress_lists_agree():
    from olmoasr_amd.decoding import DecodingOptions, suppress_list
    from oracle import decode_oracle as do
    specials = [50257, 50357, 50358, 50359, 50360, 50361]
    assert suppress_list(DecodingOptions(non_speech_tokens=())) == do.suppress_list(do.Options(non_speech_tokens=())) == specials
    # the default ("-1") expands to the tokenizer's non-speech symbol list; independent source: transformers' constant for the
    # English-only checkpoints (= the suppress_tokens of openai/whisper-*.en), which also carries sot / sot_lm / sot_prev / no_speech
    from transformers.models.whisper.configuration_whisper import NON_SPEECH_TOKENS
    from olmoasr_amd.decoding import NON_SPEECH_TOKENS_EN
    assert sorted(NON_SPEECH_TOKENS_EN) == sorted(do.NON_SPEECH_EN) == [t for t in NON_SPEECH_TOKENS if t < 50257] and len(NON_SPEECH_TOKENS_EN) == 84
    assert set(NON_SPEECH_TOKENS) - set(NON_SPEECH_TOKENS_EN) == {50257, 50359, 50360, 50361}
    assert suppress_list(DecodingOptions()) == do.suppress_list(do.Options()) == sorted(list(NON_SPEECH_TOKENS_EN) + specials)
    assert all(ord(c) - 33 in NON_SPEECH_TOKENS_EN for c in '"#()*+/:;<=>@[\\]^_`{|}~')  # ids 0..93 = printable ASCII bytes
    assert suppress_list(DecodingOptions(suppress_tokens="-1,7", non_speech_tokens=(1, 2))) == \
        do.suppress_list(do.Options(suppress_tokens=(-1, 7), non_speech_tokens=(1, 2))) == [1, 2, 7, 50257, 50357, 50358, 50359, 50360, 50361]
    assert suppress_list(DecodingOptions(suppress_tokens=None)) == []


# ---- an implementation independent of ours: transformers' generation processors (installed offline) ------------------------------
# openai-whisper is not vendored, so whisper.decoding's ApplyTimestampRules / SuppressTokens / SuppressBlank cannot be imported;
# transformers ships its own restatement of the same rules for WhisperForConditionalGeneration.  Both of ours (the oracle's and the
# product's) must mask exactly the logits it masks.
def _hf_timestamp_processor(sample_begin, max_initial_index):
    from types import SimpleNamespace
    from transformers.generation.logits_process import WhisperTimeStampLogitsProcessor
    cfg = SimpleNamespace(no_timestamps_token_id=50362, eos_token_id=50256, bos_token_id=50256, max_initial_timestamp_index=max_initial_index,
                          _detect_timestamp_from_logprob=True)
    return WhisperTimeStampLogitsProcessor(cfg, begin_index=sample_begin)


def _random_history(g, n, length, sample_begin):
    """Token rows [n, sample_begin + length] that exercise every branch: text runs, single and paired timestamps, repeats."""
    rows = []
    for _ in range(n):
        seq, t = [50257] * sample_begin, 50363 + int(torch.randint(0, 40, (1,), generator=g))
        while len(seq) < sample_begin + length:
            kind = int(torch.randint(0, 4, (1,), generator=g))
            if kind == 0:
                seq.append(int(torch.randint(0, 50256, (1,), generator=g)))
            elif kind == 1:
                seq.append(t)
            elif kind == 2:
                t += int(torch.randint(0, 30, (1,), generator=g))
                seq += [t, t]
            else:
                t += int(torch.randint(1, 30, (1,), generator=g))
                seq.append(min(t, 51863))
        rows.append(seq[:sample_begin + length])
    return torch.tensor(rows, dtype=torch.int64)


@pytest.mark.parametrize("length", [0, 1, 2, 3, 7, 16])
@pytest.mark.parametrize("max_initial_index", [None, 50])
def test_timestamp_rules_equal_transformers_processor(length, max_initial_index):
    import torch as T
    from olmoasr_amd import decoding as dec
    from oracle import decode_oracle as do
    g = T.Generator().manual_seed(100 + length)
    sample_begin, n, V = 1, 24, 51864
    tokens = _random_history(g, n, length, sample_begin)
    logits = T.randn(n, V, generator=g) * 3.0
    logits[: n // 2, 50363:] += 6.0  # half the rows: enough timestamp mass to trigger the "sample a timestamp" rule
    want = _hf_timestamp_processor(sample_begin, max_initial_index)(tokens, logits.clone())
    a, b = logits.clone(), logits.clone()
    do.apply_timestamp_rules(a, tokens, sample_begin, max_initial_index)
    dec._timestamp_rules(b, tokens, sample_begin, max_initial_index)
    for got, name in ((a, "oracle"), (b, "product")):
        assert T.equal(T.isinf(got), T.isinf(want)), name
        assert T.equal(got[~T.isinf(got)], want[~T.isinf(want)]), name


def test_suppress_processors_equal_transformers():
    """SuppressTokens = transformers' SuppressTokensLogitsProcessor on our suppress list; SuppressBlank = its
    SuppressTokensAtBeginLogitsProcessor([blank, eot]) at the first sampled position only."""
    import torch as T
    from transformers.generation.logits_process import SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor
    from olmoasr_amd import decoding as dec
    opt = dec.DecodingOptions()
    sup = dec.suppress_list(opt)
    g = T.Generator().manual_seed(3)
    logits = T.randn(4, 51864, generator=g)
    want = SuppressTokensLogitsProcessor(sup)(T.zeros(4, 2, dtype=T.long), logits.clone())
    mask = T.zeros(51864)
    mask[sup] = -float("inf")
    assert T.equal(logits + mask, want)
    begin = SuppressTokensAtBeginLogitsProcessor([220, 50256], begin_index=2)
    first = T.zeros(51864)
    first[[220, 50256]] = -float("inf")
    assert T.equal(logits + first, begin(T.zeros(4, 2, dtype=T.long), logits.clone()))          # at sample_begin
    assert T.equal(logits, begin(T.zeros(4, 3, dtype=T.long), logits.clone()))                  # later: untouched
