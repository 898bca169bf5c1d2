"""Independent pin of the beam-search / ranker / sampling restatement in oracle/decode_oracle.py (row f2).

whisper.decoding's BeamSearchDecoder + MaximumLikelihoodRanker is a heuristic; in the regime where its beam is wide enough to
keep EVERY live prefix (beam_size >= live prefixes at each depth, beam_size + 1 >= admissible tokens per step, patience large
enough that the candidate list never fills) it is an exhaustive search, and its answer must equal a brute-force enumeration
of all admissible sequences scored by the ranker's rule.  The enumeration below shares nothing with the oracle's search: the
logit filters are transformers' own processors (WhisperTimeStampLogitsProcessor, SuppressTokens(AtBegin)LogitsProcessor -- an
implementation independent of ours), the scoring is written from the definition (sum of log-softmax values of the chosen
tokens; sum / length, or sum / ((5 + length) / 6) ** alpha; avg_logprob = sum / (length + 1)), and the "language model" is a
synthetic prefix-hash model patched under both.  What this pins: candidate merging and source bookkeeping, finished-sequence
handling, finalize (open beams closed with eot), the ranker and its length penalty, avg_logprob, and their interplay with the
timestamp rules.  What it cannot pin is which prefixes a NARROW beam drops; for that regime the test asserts the invariants
any correct pruning keeps (scores are true model log-probabilities; wider beams never rank worse).
Sampling (temperature > 0, best_of) is pinned by its invariants: reported scores are the UN-tempered log-probabilities of the
returned tokens, and token frequencies follow softmax(logits / T)."""
import itertools
import math
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

from oracle import decode_oracle as do
from oracle import model_oracle as mo

V = 51864
T0 = do.TIMESTAMP_BEGIN


def _prefix_logits(prefix, scale):
    g = torch.Generator().manual_seed(hash(tuple(int(t) for t in prefix)) & 0x7FFFFFFF)
    return torch.randn(V, generator=g) * scale


@pytest.fixture()
def synthetic_lm(monkeypatch):
    """decoder_forward(tokens)[i, l] = a fixed pseudo-random function of tokens[i, :l+1]; the encoder is a no-op."""
    state = {"scale": 2.0}
    monkeypatch.setattr(mo, "encoder_forward", lambda sd, dims, mel: torch.zeros(mel.shape[0], 1, 1))

    def decoder_forward(sd, dims, tokens, xa):
        return torch.stack([torch.stack([_prefix_logits(row[: l + 1].tolist(), state["scale"]) for l in range(tokens.shape[1])]) for row in tokens])
    monkeypatch.setattr(mo, "decoder_forward", decoder_forward)
    return state


def _hf_filters(sample_begin, with_timestamps, sup):
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    procs = [SuppressTokensAtBeginLogitsProcessor([do.BLANK, do.EOT], begin_index=sample_begin), SuppressTokensLogitsProcessor(sup)]
    if with_timestamps:
        cfg = SimpleNamespace(no_timestamps_token_id=do.NO_TIMESTAMPS, eos_token_id=do.EOT, bos_token_id=do.EOT,
                              max_initial_timestamp_index=50, _detect_timestamp_from_logprob=True)
        procs.append(WhisperTimeStampLogitsProcessor(cfg, begin_index=sample_begin))
    return procs


def brute_force(init, bias, scale, depth, with_timestamps, length_penalty):
    """Every admissible continuation of ``init`` up to ``depth`` tokens -> (tokens, sum_logprob), best by the ranker's rule."""
    procs = _hf_filters(len(init), with_timestamps, do.suppress_list(do.Options()))
    done = []

    def expand(prefix, lp):
        body = prefix[len(init):]
        if len(body) == depth:
            done.append((body, lp))  # still open at the length limit: closed with eot, score unchanged (finalize)
            return
        lg = (_prefix_logits(prefix, scale) + bias)[None]
        ids = torch.tensor([prefix])
        for p in procs:
            lg = p(ids, lg)
        logp = F.log_softmax(lg[0].float(), -1)
        for t in torch.nonzero(torch.isfinite(lg[0])).flatten().tolist():
            if t == do.EOT:
                done.append((body, lp + float(logp[t])))
            else:
                expand(prefix + [t], lp + float(logp[t]))
    expand(list(init), 0.0)

    def score(c):
        n = len(c[0])
        pen = n if length_penalty is None else ((5 + n) / 6) ** length_penalty
        return c[1] / pen if pen else -math.inf
    return max(done, key=score), done


CASES = [  # (allowed token ids, with timestamps, beam, patience, depth, length_penalty)
    ([11, 22, do.EOT], False, 8, 2.0, 3, None),
    ([11, 22, do.EOT], False, 8, 2.0, 3, 0.6),
    ([11, 22, 33, do.EOT], False, 27, 2.0, 3, None),
    ([11, 22, do.EOT], False, 16, 2.0, 4, 1.0),
    ([11, 22, T0 + 5, T0 + 9, T0 + 30, do.EOT], True, 24, 3.0, 3, None),
    ([11, T0 + 2, T0 + 3, T0 + 40, do.EOT], True, 24, 3.0, 4, 0.3),
]


@pytest.mark.parametrize("seed_scale", [1.0, 3.0])
@pytest.mark.parametrize("case", CASES)
def test_wide_beam_equals_brute_force(synthetic_lm, case, seed_scale):
    allowed, with_ts, beam, patience, depth, alpha = case
    synthetic_lm["scale"] = seed_scale
    bias = torch.full((V,), -math.inf)
    bias[allowed] = 0.0
    init = [do.SOT] if with_ts else [do.SOT, do.NO_TIMESTAMPS]
    (best, best_lp), done = brute_force(init, bias, seed_scale, depth, with_ts, alpha)
    assert len(done) >= 6
    dims = SimpleNamespace(n_text_ctx=448)
    got = do.decode(None, dims, torch.zeros(1, 80, 3000), do.Options(beam_size=beam, patience=patience, sample_len=depth, length_penalty=alpha,
                                                                      without_timestamps=not with_ts, logit_bias=bias))[0]
    assert got.tokens == best, (got.tokens, best)
    assert abs(got.sum_logprob - best_lp) < 1e-4 and abs(got.avg_logprob - best_lp / (len(best) + 1)) < 1e-5


def test_narrow_beam_invariants(synthetic_lm):
    """With a beam too narrow to be exhaustive: the reported score is still the true log-probability of the returned tokens,
    the returned sequence is admissible, and widening the beam never lowers the ranker's score of the answer."""
    synthetic_lm["scale"] = 2.5
    allowed = [11, 22, 33, 44, 55, do.EOT]
    bias = torch.full((V,), -math.inf)
    bias[allowed] = 0.0
    init = [do.SOT, do.NO_TIMESTAMPS]
    _, done = brute_force(init, bias, 2.5, 4, False, None)
    table = {tuple(b): lp for b, lp in done}
    prev = -math.inf
    for beam in (1, 2, 3, 5, 8, 40):
        r = do.decode(None, SimpleNamespace(n_text_ctx=448), torch.zeros(1, 80, 3000),
                      do.Options(beam_size=beam, patience=4.0, sample_len=4, without_timestamps=True, logit_bias=bias))[0]
        assert tuple(r.tokens) in table and abs(table[tuple(r.tokens)] - r.sum_logprob) < 1e-4
        score = r.sum_logprob / max(len(r.tokens), 1)
        assert score >= prev - 1e-6, (beam, score, prev)
        prev = score
    best = max(done, key=lambda c: c[1] / len(c[0]) if c[0] else -math.inf)
    assert abs(prev - best[1] / len(best[0])) < 1e-5  # beam 40 is exhaustive here


def test_sampling_scores_and_frequencies(synthetic_lm):
    """GreedyDecoder with temperature: tokens ~ softmax(logits / T); sum_logprob accumulates the UN-tempered log-softmax;
    best_of keeps the candidate with the best average."""
    synthetic_lm["scale"] = 1.5
    allowed = [11, 22, 33]
    bias = torch.full((V,), -math.inf)
    bias[allowed] = 0.0
    init = [do.SOT, do.NO_TIMESTAMPS]
    temp, n = 0.7, 600
    lg = (_prefix_logits(init, 1.5) + bias)
    want = F.softmax(lg[allowed] / temp, -1)
    counts = {t: 0 for t in allowed}
    for seed in range(n // 6):
        r = do.decode(None, SimpleNamespace(n_text_ctx=448), torch.zeros(6, 80, 3000),
                      do.Options(temperature=temp, sample_len=2, without_timestamps=True, logit_bias=bias, seed=seed))
        for x in r:
            counts[x.tokens[0]] += 1
            lp = 0.0  # recompute the un-tempered log-probability of what was returned
            pre = list(init)
            for t in x.tokens:
                lp += float(F.log_softmax((_prefix_logits(pre, 1.5) + bias).float(), -1)[t])
                pre.append(t)
            assert abs(lp - x.sum_logprob) < 1e-4 and abs(x.avg_logprob - lp / (len(x.tokens) + 1)) < 1e-5
    for t, p in zip(allowed, want.tolist()):
        sigma = math.sqrt(p * (1 - p) / n)
        assert abs(counts[t] / n - p) < 4.5 * sigma + 1e-3, (t, counts[t] / n, p)
    # best_of: the winner's average log-probability is the maximum over the group's samples (same seed -> same samples)
    grp = do.decode(None, SimpleNamespace(n_text_ctx=448), torch.zeros(1, 80, 3000),
                    do.Options(temperature=temp, best_of=5, sample_len=3, without_timestamps=True, logit_bias=bias, seed=7))[0]
    assert grp.temperature == temp and len(grp.tokens) == 3
