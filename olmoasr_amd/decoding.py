"""Decoding on the native engine: ``OLMoASR.decode`` (reference binding olmoasr/model.py:966-968, inf_model.py:455-457).

The reference calls ``whisper.decoding.decode`` -- at scripts/training/train_timestamps.py:1916-1919 (greedy,
``DecodingOptions(language="en", without_timestamps=True)``), scripts/eval/eval.py:1846-1847 and, through
``decode_with_fallback``, olmoasr/transcribe.py:193-233 (eval.py:2077-2084: beam 5 + temperature fallback).
``openai-whisper`` is not vendored by the reference (requirements.txt:21) and its tokenizer (tiktoken) is unavailable
offline, so the TOKEN-level behaviour of its ``DecodingTask`` is restated here (``oracle/decode_oracle.py`` holds an
independent CPU restatement; tests compare token ids and scores exactly in fp32 validation mode); text decoding, the
text-based compression ratio and word timestamps are left to the caller.  Parity for this file is "unpinned by the
reference": it ships no tests or vectors for decoding.

  DecodingTask.run:  initial tokens [sot] (+ [notimestamps] when without_timestamps) -- the English-only specials the
  reference's Dataset writes (train_timestamps.py:345-506); up to sample_len (= n_text_ctx // 2) steps of
      logits = decoder(tokens, audio_features)[:, -1]          (every row the head has: the training model's pad class too)
      SuppressBlank (blank " " and eot at the first sampled position), SuppressTokens ("-1" = the tokenizer's non-speech
      symbol ids, NON_SPEECH_TOKENS_EN, + transcribe/translate/sot/sot_prev/sot_lm/no_speech), ApplyTimestampRules (unless without_timestamps)
      GreedyDecoder (argmax, or Categorical(logits / T) with best_of samples) or BeamSearchDecoder (beam_size, patience)
  then MaximumLikelihoodRanker (sum_logprob / length, or the length_penalty form) and avg_logprob = sum / (len + 1);
  no_speech_prob = softmax(logits at the sot position)[no_speech].

Step engines: greedy / sampling run on the engine-owned KV cache (``kv_cache_begin/kv_cache_step`` = the reference's
``install_kv_cache_hooks`` (model.py:925-964) + a one-token decoder step; the per-step argmax + log-softmax gather with the
suppress masks is one HIP kernel, ``oasr_pick_tokens`` -- in timestamp mode ``oasr_pick_tokens_ts``, which also applies
ApplyTimestampRules from the device-resident token history, so the greedy loop never waits for the GPU); beam search runs on the
same cache, its rows re-gathered in place between steps (``kv_cache_reorder`` = whisper's ``rearrange_kv_cache``,
inf_model.py:422-453 hooks); ``use_kv_cache=False`` re-runs the decoder on the whole prefix each step (the pattern of
notebooks/ow_decoding.py:42-72).
"""
import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import torch

from . import ops

EOT = 50256
SOT = 50257
TRANSLATE, TRANSCRIBE, SOT_LM, SOT_PREV = 50357, 50358, 50359, 50360
NO_SPEECH = 50361
NO_TIMESTAMPS = 50362
TIMESTAMP_BEGIN = 50363  # <|0.00|>; id = TIMESTAMP_BEGIN + ms // 20 (train_timestamps.py:236)
BLANK = 220              # GPT-2 BPE id of " ": what whisper's SuppressBlank masks (tokenizer.encode(" "))
# whisper.tokenizer.Tokenizer.non_speech_tokens evaluated on the GPT-2 (English-only) vocabulary: what suppress_tokens="-1" -- the
# default of DecodingOptions, hence of the reference's DecodingOptions(language="en", without_timestamps=True) at
# train_timestamps.py:1916 and of transcribe() -- expands to.  The rule: every symbol of  " # ( ) * + / : ; < = > @ [ \ ] ^ _ ` { | } ~
# and the bracket / dash / music-note strings that encodes to ONE token, bare and with a leading space, plus the first token of " -"
# and " '".  A constant of the vocabulary (ids 0..93 are the printable ASCII bytes '!'..'~': '"' = 1, '#' = 2, '(' = 7, ...); the same
# 84 ids ship as the ``suppress_tokens`` of every openai/whisper-*.en config and as transformers'
# ``models/whisper/configuration_whisper.py::NON_SPEECH_TOKENS`` (minus its four specials), against which the tests pin this table.
NON_SPEECH_TOKENS_EN = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 357, 366, 438, 532, 685, 705, 796, 930,
    1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635, 1782, 1875, 2162, 2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855,
    6329, 7203, 9609, 9959, 10563, 10786, 11420, 11709, 11907, 13163, 13697, 13700, 14808, 15306, 16410, 16791, 17992, 19203, 19510,
    20724, 22305, 22935, 27007, 30109, 30420, 33409, 34949, 40283, 40493, 40549, 47282, 49146)


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = "en"
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Sequence[int]] = None   # accepted for signature parity; the reference has prompt conditioning commented out
    prefix: Optional[Sequence[int]] = None
    suppress_tokens: Optional[Union[str, Sequence[int]]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True
    # --- not in whisper: what its tokenizer would have supplied, and engine switches -----------------------------------
    non_speech_tokens: Optional[Sequence[int]] = None  # what "-1" expands to; None = NON_SPEECH_TOKENS_EN (the tokenizer's list)
    initial_tokens: Optional[Sequence[int]] = None   # override of the sot sequence
    suppress_mask: Optional[torch.Tensor] = None     # extra additive mask [rows] (0 / -inf)
    seed: Optional[int] = None                       # sampling generator seed (temperature > 0)
    use_kv_cache: bool = True


@dataclass
class DecodingResult:
    audio_features: torch.Tensor
    language: str = "en"
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = float("nan")
    no_speech_prob: float = float("nan")
    temperature: float = float("nan")
    compression_ratio: float = float("nan")  # needs text (gzip of the decoded string): not available without the tokenizer


def suppress_list(options: DecodingOptions) -> List[int]:
    """DecodingTask._get_suppress_tokens."""
    sup = options.suppress_tokens
    if sup is None:
        return []
    if isinstance(sup, str):
        sup = [int(t) for t in sup.split(",") if t.strip()]
    sup = list(sup)
    if -1 in sup:
        sup = [t for t in sup if t >= 0] + list(NON_SPEECH_TOKENS_EN if options.non_speech_tokens is None else options.non_speech_tokens)
    sup += [TRANSCRIBE, TRANSLATE, SOT, SOT_PREV, SOT_LM, NO_SPEECH]
    return sorted(set(sup))


def _timestamp_rules(logits: torch.Tensor, tokens: torch.Tensor, sample_begin: int, max_initial_index: Optional[int]):
    """whisper.decoding.ApplyTimestampRules on logits [n, rows], in place -- evaluated with tensor ops on the device the logits live
    on (no per-row Python, no device -> host copy: the decode loop never waits for the GPU here).  The greedy path does not come
    through here at all: ``oasr_pick_tokens_ts`` applies the same rules inside the pick kernel."""
    ninf = -float("inf")
    n, rows = logits.shape
    dev = logits.device
    logits[:, NO_TIMESTAMPS] = ninf
    seq = tokens[:, sample_begin:].to(dev)
    m = seq.shape[1]
    cols = torch.arange(rows, device=dev)[None, :]
    if m >= 1:
        is_ts = seq >= TIMESTAMP_BEGIN
        last_was_ts = is_ts[:, -1]
        penultimate_was_ts = is_ts[:, -2] if m >= 2 else torch.ones(n, dtype=torch.bool, device=dev)
        # a timestamp pair has to be followed by text; an opening timestamp + text has to be closed before more text ... :
        dead = (last_was_ts & penultimate_was_ts)[:, None] & (cols >= TIMESTAMP_BEGIN)
        dead |= (last_was_ts & ~penultimate_was_ts)[:, None] & (cols < EOT)
        # ... and timestamps neither decrease nor close a zero-length segment
        pos = torch.arange(1, m + 1, device=dev)[None, :]
        last_pos = (is_ts * pos).max(dim=1).values                       # 1-based position of the last timestamp, 0 = none
        last_ts = seq.gather(1, (last_pos - 1).clamp(min=0)[:, None])[:, 0]
        lo = last_ts + (~(last_was_ts & ~penultimate_was_ts)).to(last_ts.dtype)
        dead |= (last_pos > 0)[:, None] & (cols >= TIMESTAMP_BEGIN) & (cols < lo[:, None])
        logits.masked_fill_(dead, ninf)
    else:
        logits[:, :TIMESTAMP_BEGIN] = ninf           # the first sampled token is a timestamp ...
        if max_initial_index is not None:
            logits[:, TIMESTAMP_BEGIN + max_initial_index + 1:] = ninf  # ... no later than max_initial_timestamp
    logprobs = torch.log_softmax(logits.float(), dim=-1)
    force = logprobs[:, TIMESTAMP_BEGIN:].logsumexp(dim=-1) > logprobs[:, :TIMESTAMP_BEGIN].max(dim=-1).values
    logits[:, :TIMESTAMP_BEGIN].masked_fill_(force[:, None], ninf)  # the timestamp mass beats every text token: sample a timestamp


MAX_BEAM_SIZE = 15  # oasr_topk_tokens: K = beam_size + 1 <= 16


def compression_ratio(text: str) -> float:
    """whisper.utils.compression_ratio: utf-8 bytes over their zlib-compressed size -- high for repetitive text (the "too repetitive"
    test of olmoasr/transcribe.py:213-217)."""
    b = text.encode("utf-8")
    return len(b) / len(zlib.compress(b))


def resolve_tokenizer(model, tokenizer=None, language: str = "en", task: str = "transcribe"):
    """The tokenizer PLUG: openai-whisper is not vendored by the reference (requirements.txt:21) and is absent offline, so text is
    optional here.  ``tokenizer``: any object with ``decode(list[int]) -> str`` (and ``encode(str) -> list[int]`` for
    ``initial_prompt``), e.g. ``whisper.tokenizer.get_tokenizer(...)``; ``None``: whisper's own when the package is importable (the
    reference's environment, olmoasr/transcribe.py:167-172), else no text -- token-level results only."""
    if tokenizer is not None:
        return tokenizer
    try:
        from whisper.tokenizer import get_tokenizer  # noqa: PLC0415 -- optional dependency of the reference's environment
    except Exception:
        return None
    return get_tokenizer(getattr(model, "is_multilingual", False), num_languages=getattr(model, "num_languages", 0), language=language, task=task)


@torch.no_grad()
def decode(model, mel: torch.Tensor, options: Optional[DecodingOptions] = None, *, tokenizer=None, **kwargs):
    """``mel``: [80, 3000] or [n, 80, 3000] windows (or already-encoded audio features).  Returns DecodingResult / list.
    With a ``tokenizer`` (see ``resolve_tokenizer``) the results carry ``text`` and ``compression_ratio`` exactly as
    whisper.decoding.DecodingTask.run fills them: text = tokenizer.decode(tokens without timestamps).strip()."""
    out = _decode(model, mel, options, tokenizer, kwargs)
    if out is _RETRY:  # oasr_decode_check switched the context to the multi-launch step engine (model.kv_cache_check): same window again
        out = _decode(model, mel, options, tokenizer, kwargs)
        if out is _RETRY:
            raise RuntimeError("decode: the step engine asked for a second retry of one window")
    return out


_RETRY = object()


def _decode(model, mel, options, tokenizer, kwargs):
    if options is None:
        options = DecodingOptions(**kwargs)
    elif kwargs:
        options = DecodingOptions(**{**options.__dict__, **kwargs})
    if options.beam_size is not None and options.best_of is not None:
        raise ValueError("beam_size and best_of can't be given together")
    if options.temperature == 0.0 and options.best_of is not None:
        raise ValueError("best_of with greedy sampling (T=0) is not compatible")
    if options.patience is not None and options.beam_size is None:
        raise ValueError("patience requires beam_size to be given")
    if options.beam_size is not None and not 1 <= options.beam_size <= MAX_BEAM_SIZE:
        # the selection kernel keeps beam_size + 1 candidates per row in registers (csrc/loss.hip: topk_ts_kernel, K <= 16); whisper and
        # the reference's eval use 5 (eval.py:2077-2084).  Refused up front instead of failing inside the first step.
        raise ValueError(f"beam_size must be in [1, {MAX_BEAM_SIZE}] on the native selection kernel, got {options.beam_size}")
    if options.length_penalty is not None and not (0 <= options.length_penalty <= 1):
        raise ValueError("length_penalty (alpha) should be a value between 0 and 1")
    single = mel.dim() == 2
    if single:
        mel = mel[None]
    dims = model.dims
    rows = model._n_rows
    n_audio = mel.shape[0]
    xa = model.embed_audio(mel) if mel.shape[-2:] == (dims.n_mels, 2 * dims.n_audio_ctx) else mel
    dev = xa.device
    init = list(options.initial_tokens) if options.initial_tokens is not None else (
        [SOT, NO_TIMESTAMPS] if options.without_timestamps else [SOT])
    sample_begin = len(init)
    sot_index = init.index(SOT) if SOT in init else 0
    sample_len = min(options.sample_len or dims.n_text_ctx // 2, dims.n_text_ctx - sample_begin)
    n_group = options.beam_size or options.best_of or 1
    beam = options.beam_size
    max_candidates = round(beam * (options.patience or 1.0)) if beam else None
    gen = None
    if options.temperature > 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(options.seed if options.seed is not None else 0)
    max_init_idx = None
    if not options.without_timestamps and options.max_initial_timestamp is not None:
        max_init_idx = round(options.max_initial_timestamp / 0.02)
    # additive masks: every step / first sampled position only
    base_mask = torch.zeros(rows, device=dev)
    sup = [t for t in suppress_list(options) if t < rows]
    if sup:
        base_mask[sup] = -float("inf")
    if options.suppress_mask is not None:
        base_mask[: options.suppress_mask.numel()] += options.suppress_mask.to(dev).float()
    first_mask = None
    if options.suppress_blank:
        first_mask = torch.zeros(rows, device=dev)
        first_mask[[BLANK, EOT]] = -float("inf")

    xa_g = xa.repeat_interleave(n_group, dim=0) if n_group > 1 else xa
    n = n_audio * n_group
    tokens = torch.tensor([init] * n, dtype=torch.int64, device=dev)
    sum_logprobs = torch.zeros(n, device=dev)
    finished = [dict() for _ in range(n_audio)] if beam else None
    cached = options.use_kv_cache
    state = None
    # position sot_index gives no_speech_prob; with the cache it is the logits returned while the prompt is being fed
    if cached:
        state = model.kv_cache_begin(xa_g)
        step_logits = None
        for p in range(sample_begin):
            step_logits = model.kv_cache_step(state, tokens[:, p])
            if p == sot_index:
                no_speech = torch.softmax(step_logits.float(), dim=-1)[::n_group, NO_SPEECH].tolist()
    for i in range(sample_len):
        if cached:
            lg = step_logits if i == 0 else model.kv_cache_step(state, tokens[:, -1])
        elif i == 0:
            full = model.logits(tokens, xa_g)
            no_speech = torch.softmax(full[:, sot_index].float(), dim=-1)[::n_group, NO_SPEECH].tolist()
            lg = full[:, -1].contiguous()
        else:
            lg = model.logits(tokens, xa_g, last_only=True)
        fm = first_mask if i == 0 else None
        simple = not beam and options.temperature == 0
        if simple and options.without_timestamps:  # argmax + log-softmax gather + masks in ONE kernel
            nxt, cur = ops.pick_tokens(lg, base_mask, fm)
        elif simple:  # ... and, in timestamp mode, ApplyTimestampRules from the device-resident history in the same kernel
            nxt, cur = ops.pick_tokens_ts(lg.float().contiguous(), tokens[:, sample_begin:], tokens.shape[1] - sample_begin,
                                          timestamp_begin=TIMESTAMP_BEGIN, eot=EOT, no_timestamps=NO_TIMESTAMPS,
                                          max_initial_index=max_init_idx, mask=base_mask, mask2=fm)
        # beam search and sampling select on the device too (round 4): one kernel applies the masks and -- in timestamp mode -- the
        # rules from the device-resident history, then returns the beam_size + 1 best log-probabilities / draws by inverse CDF
        sel = dict(history=tokens[:, sample_begin:], n_history=None if options.without_timestamps else tokens.shape[1] - sample_begin,
                   timestamp_begin=TIMESTAMP_BEGIN, eot=EOT, no_timestamps=NO_TIMESTAMPS, max_initial_index=max_init_idx, mask=base_mask, mask2=fm)
        if beam:  # BeamSearchDecoder.update
            top_lp, top_tok = ops.topk_tokens(lg.float().contiguous(), beam + 1, **sel)
            top_lp, top_tok = top_lp.cpu(), top_tok.cpu()
            prev_sum, tok_cpu = sum_logprobs.cpu(), tokens.cpu()
            next_tokens, new_sum, sources = [], [], []
            for a in range(n_audio):
                scores, newly, origin = {}, {}, {}
                for j in range(beam):
                    idx = a * beam + j
                    prefix = tok_cpu[idx].tolist()
                    for lp, t in zip(top_lp[idx].tolist(), top_tok[idx].tolist()):
                        seq = tuple(prefix + [t])
                        scores[seq] = float(prev_sum[idx]) + lp
                        origin[seq] = idx
                saved = 0
                for seq in sorted(scores, key=scores.get, reverse=True):
                    if seq[-1] == EOT:
                        newly[seq] = scores[seq]
                    else:
                        new_sum.append(scores[seq])
                        next_tokens.append(list(seq))
                        sources.append(origin[seq])
                        saved += 1
                        if saved == beam:
                            break
                for seq in sorted(newly, key=newly.get, reverse=True):
                    if len(finished[a]) >= max_candidates:
                        break
                    finished[a][seq] = newly[seq]
            tokens = torch.tensor(next_tokens, dtype=torch.int64, device=dev)
            sum_logprobs = torch.tensor(new_sum, device=dev)
            if cached and sources != list(range(n)):  # PyTorchInference.rearrange_kv_cache: row j continues row sources[j]
                model.kv_cache_reorder(state, sources)
            completed = all(len(f) >= max_candidates for f in finished)
        else:  # GreedyDecoder.update
            if not simple:  # temperature > 0: Categorical(logits / T) by inverse CDF on uniforms of this decode's generator
                nxt, cur = ops.sample_tokens(lg.float().contiguous(), options.temperature, torch.rand(lg.shape[0], device=dev, generator=gen), **sel)
            alive = tokens[:, -1] != EOT
            sum_logprobs = sum_logprobs + cur * alive
            nxt = torch.where(alive, nxt, torch.full_like(nxt, EOT))
            tokens = torch.cat([tokens, nxt[:, None]], dim=1)
            # The all-rows-finished test is a device -> host sync.  Finished rows only ever append eot and add 0 to their score, so
            # testing every 8th step decodes at most 7 surplus positions and changes no result -- but lets the host run ahead of
            # the GPU in between (a KV-cached step is ~1 ms of GPU time behind ~0.9 ms of enqueue).
            completed = bool((tokens[:, -1] == EOT).all()) if (not cached or i % 8 == 7 or i == sample_len - 1) else False
        if completed or tokens.shape[-1] > dims.n_text_ctx:
            break

    # ---- finalize + rank (MaximumLikelihoodRanker): sum_logprob / length, or the Google-NMT penalty when given
    def rank(cands):  # [(tokens between the prompt and eot, sum_logprob)]
        def score(c):
            length = len(c[0])
            pen = length if options.length_penalty is None else ((5 + length) / 6) ** options.length_penalty
            return c[1] / pen if pen else -float("inf")
        return max(cands, key=score)

    results = []
    if state is not None and not model.kv_cache_check(state):
        return _RETRY
    tok_cpu, sums = tokens.cpu(), sum_logprobs.cpu()
    for a in range(n_audio):
        if beam:
            f = dict(finished[a])
            if len(f) < beam:  # BeamSearchDecoder.finalize: close the best unfinished beams with eot
                for j in sorted(range(beam), key=lambda j: float(sums[a * beam + j]), reverse=True):
                    f[tuple(tok_cpu[a * beam + j].tolist() + [EOT])] = float(sums[a * beam + j])
                    if len(f) >= beam:
                        break
            seqs = [(list(seq), lp) for seq, lp in f.items()]
        else:
            seqs = [(tok_cpu[a * n_group + j].tolist() + [EOT], float(sums[a * n_group + j])) for j in range(n_group)]
        cands = []
        for seq, lp in seqs:
            body = seq[sample_begin:]
            cands.append((body[:body.index(EOT)], lp))
        best, lp = rank(cands)
        res = DecodingResult(audio_features=xa[a], tokens=best, avg_logprob=lp / (len(best) + 1), no_speech_prob=no_speech[a],
                             temperature=options.temperature)
        if tokenizer is not None:  # (whisper's Tokenizer.decode drops the timestamp tokens itself; any other plug gets them dropped here)
            res.text = tokenizer.decode([t for t in best if t < TIMESTAMP_BEGIN]).strip()
            res.compression_ratio = compression_ratio(res.text)
        results.append(res)
    return results[0] if single else results


def detect_language(model, mel: torch.Tensor, tokenizer=None):
    """whisper.decoding.detect_language as bound at olmoasr/model.py:966: the OLMoASR checkpoints are English-only
    (n_vocab 51864: no language tokens), for which whisper raises exactly this error."""
    raise ValueError("This model doesn't have language tokens so it can't perform lang id")


def greedy_token_matrix(model, mel: torch.Tensor, max_new: int, initial_tokens=(SOT, NO_TIMESTAMPS)) -> torch.Tensor:
    """[B, len(initial)+n] token matrix exactly as oracle.model_oracle.greedy_decode produces it (plain argmax over the
    n_vocab classes, no suppression -- the notebooks/ow_decoding.py:42-72 loop; parity tests)."""
    xa = model.embed_audio(mel)
    B = mel.shape[0]
    toks = torch.tensor([list(initial_tokens)] * B, dtype=torch.int64, device=xa.device)
    done = torch.zeros(B, dtype=torch.bool, device=xa.device)
    for _ in range(max_new):
        nxt = model.logits(toks, xa, last_only=True)[:, :model.dims.n_vocab].argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, EOT), nxt)
        toks = torch.cat([toks, nxt[:, None]], dim=1)
        done |= nxt == EOT
        if bool(done.all()):
            break
    return toks
