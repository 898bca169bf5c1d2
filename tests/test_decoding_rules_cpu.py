"""CPU tests of the token-level logit filters of olmoasr_amd.decoding (no device needed): ApplyTimestampRules as
whisper.decoding defines them, with the English-only ids the reference's Dataset writes."""
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
    assert suppress_list(DecodingOptions()) == do.suppress_list(do.Options()) == [50257, 50357, 50358, 50359, 50360, 50361]
    assert suppress_list(DecodingOptions(suppress_tokens="-1,7", non_speech_tokens=(1, 2))) == \
        do.suppress_list(do.Options(suppress_tokens=(-1, 7), non_speech_tokens=(1, 2))) == [1, 2, 7, 50257, 50357, 50358, 50359, 50360, 50361]
    assert suppress_list(DecodingOptions(suppress_tokens=None)) == []
