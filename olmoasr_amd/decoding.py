"""Greedy decoding on the native engine.

Replaces the greedy subset of ``whisper.decoding.decode`` that the reference binds as ``OLMoASR.decode``
(olmoasr/model.py:966-968, olmoasr/inf_model.py:455-457) and calls at scripts/training/train_timestamps.py:1916-1919
(``DecodingOptions(language="en", without_timestamps=True)``, temperature 0) and scripts/eval/eval.py:1846-1847.
``openai-whisper`` is not vendored by the reference (requirements.txt:21) and its tokenizer (tiktoken) is unavailable
offline, so the *token-level* behaviour of ``DecodingTask``/``GreedyDecoder`` is restated here and text decoding is left to
the caller:

  tokens = initial_tokens (default [sot 50257, notimestamps 50362], the English-only specials the reference's Dataset
  writes at train_timestamps.py:345-506); repeat up to ``sample_len`` (= n_text_ctx // 2 = 224) times:
      logits = decoder(tokens, audio_features)[:, -1, :n_vocab]   (the pad class of the training head is never sampled)
      logits += suppress_mask (SuppressBlank / SuppressTokens as an explicit additive mask; -inf entries)
      next = argmax(logits); rows that already emitted eot keep emitting eot; stop when every row has.
  sum_logprobs accumulates log_softmax(logits)[next] of the sampled (non-eot-padding) tokens -> avg_logprob, as whisper.

Beyond greedy, restated from the published algorithm of ``whisper.decoding`` at token level
(no tokenizer is needed for any of it; text decoding and the text-based compression-ratio test are the caller's):
  * ``beam_size`` (+ ``patience``): BeamSearchDecoder -- per audio, expand every beam by its top ``beam_size + 1`` tokens,
    keep the best ``beam_size`` unfinished continuations, collect finished ones until ``round(beam_size * patience)``;
    unfinished beams are closed with eot at the end; the winner maximises sum_logprob / length (MaximumLikelihoodRanker with
    length_penalty None).
  * ``temperature > 0`` (+ ``best_of``): multinomial sampling from softmax(logits / T), ``best_of`` independent samples per
    audio ranked the same way.
  * ``without_timestamps=False``: ApplyTimestampRules with the English-only ids the reference's Dataset writes
    (timestamp_begin 50363 = ``<|0.00|>``, 20 ms per id, train_timestamps.py:236): timestamps come in pairs, are
    non-decreasing, ``<|notimestamps|>`` is never sampled, the first sampled token is a timestamp no later than
    ``max_initial_timestamp``, and a timestamp is forced when the timestamp mass beats every text token.
  * ``no_speech_prob``: softmax probability of ``<|nospeech|>`` (50361) at the sot position.

Two step engines: (default) the KV-cached one -- ``OLMoASR.kv_cache_begin/kv_cache_step`` = the reference's
``install_kv_cache_hooks`` (model.py:925-964) + a one-token decoder step, cross-attention K/V computed once per window;
and, for cross-checking, the cache-less one that re-runs the decoder on the whole prefix (the reference-internal pattern
of notebooks/ow_decoding.py:42-72), computing only the last position's logits.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

SOT = 50257
EOT = 50256
NO_SPEECH = 50361
NO_TIMESTAMPS = 50362
TIMESTAMP_BEGIN = 50363  # <|0.00|>; id = TIMESTAMP_BEGIN + ms // 20 (train_timestamps.py:236)


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = "en"
    temperature: float = 0.0
    sample_len: Optional[int] = None
    without_timestamps: bool = True
    initial_tokens: Optional[Sequence[int]] = None  # default [sot, notimestamps]
    suppress_mask: Optional[torch.Tensor] = None    # additive [n_vocab] (0 / -inf)
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    best_of: Optional[int] = None
    length_penalty: Optional[float] = None
    max_initial_timestamp: Optional[float] = 1.0
    seed: Optional[int] = None  # sampling generator seed (temperature > 0)
    fp16: bool = True
    use_kv_cache: bool = True  # False: re-run the decoder on the whole prefix every step (ow_decoding.py style)


@dataclass
class DecodingResult:
    audio_features: torch.Tensor
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = float("nan")
    no_speech_prob: float = float("nan")
    temperature: float = 0.0
    language: str = "en"


@torch.no_grad()
def decode(model, mel: torch.Tensor, options: Optional[DecodingOptions] = None, **kwargs):
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
    if options.temperature != 0.0 or options.beam_size or options.best_of or not options.without_timestamps:
        return _decode_general(model, mel, options)
    single = mel.dim() == 2
    if single:
        mel = mel[None]
    dims = model.dims
    B = mel.shape[0]
    xa = model.embed_audio(mel) if mel.shape[-2:] == (dims.n_mels, 2 * dims.n_audio_ctx) else mel
    init = list(options.initial_tokens) if options.initial_tokens is not None else [SOT, NO_TIMESTAMPS]
    sample_len = options.sample_len or dims.n_text_ctx // 2
    sample_len = min(sample_len, dims.n_text_ctx - len(init))
    toks = torch.tensor([init] * B, dtype=torch.int64, device=xa.device)
    done = torch.zeros(B, dtype=torch.bool, device=xa.device)
    sum_logprobs = torch.zeros(B, device=xa.device)
    sup = options.suppress_mask.to(xa.device) if options.suppress_mask is not None else None
    # no_speech_prob: P(<|nospeech|>) at the sot position (one extra 1-token decoder pass)
    no_speech = [float("nan")] * B
    if SOT in init:
        k = init.index(SOT) + 1
        p0 = torch.softmax(model.logits(toks[:, :k], xa, last_only=True)[:, :dims.n_vocab].float(), dim=-1)[:, NO_SPEECH]
        no_speech = p0.tolist()
    state = None
    if options.use_kv_cache:
        state = model.kv_cache_begin(xa)
        for p in range(len(init) - 1):  # prefill the prompt; the last prompt token is fed by the first loop iteration
            model.kv_cache_step(state, toks[:, p])
    for _ in range(sample_len):
        if state is not None:
            lg = model.kv_cache_step(state, toks[:, -1])[:, :dims.n_vocab]
        else:
            lg = model.logits(toks, xa, last_only=True)[:, :dims.n_vocab]
        if sup is not None:
            lg = lg + sup
        logp = torch.log_softmax(lg.float(), dim=-1)
        nxt = lg.argmax(-1)
        cur = logp.gather(1, nxt[:, None])[:, 0]
        sum_logprobs += torch.where(done, torch.zeros_like(cur), cur)
        nxt = torch.where(done, torch.full_like(nxt, EOT), nxt)
        toks = torch.cat([toks, nxt[:, None]], dim=1)
        done |= nxt == EOT
        if bool(done.all()):
            break
    results = []
    for b in range(B):
        row = toks[b, len(init):].tolist()
        if EOT in row:
            row = row[:row.index(EOT)]
        results.append(DecodingResult(audio_features=xa[b], tokens=row, avg_logprob=float(sum_logprobs[b]) / (len(row) + 1),  # whisper: sum / (len(tokens) + 1)
                                      no_speech_prob=no_speech[b], temperature=0.0))
    return results[0] if single else results


def _timestamp_rules(logits: torch.Tensor, tokens: torch.Tensor, sample_begin: int, n_vocab: int, max_initial_index: Optional[int]):
    """whisper.decoding.ApplyTimestampRules on additive logits [n, n_vocab], in place."""
    logits[:, NO_TIMESTAMPS] = -float("inf")
    for k in range(tokens.shape[0]):
        seq = tokens[k, sample_begin:].tolist()
        last_was_ts = len(seq) >= 1 and seq[-1] >= TIMESTAMP_BEGIN
        penultimate_was_ts = len(seq) < 2 or seq[-2] >= TIMESTAMP_BEGIN
        if last_was_ts:
            if penultimate_was_ts:
                logits[k, TIMESTAMP_BEGIN:] = -float("inf")   # has to be non-timestamp
            else:
                logits[k, :EOT] = -float("inf")               # cannot be normal text tokens
        ts = [t for t in seq if t >= TIMESTAMP_BEGIN]
        if ts:  # timestamps shouldn't decrease; also force each segment to have a nonzero length
            last = ts[-1] if (last_was_ts and not penultimate_was_ts) else ts[-1] + 1
            logits[k, TIMESTAMP_BEGIN:last] = -float("inf")
    if tokens.shape[1] == sample_begin:
        logits[:, :TIMESTAMP_BEGIN] = -float("inf")           # suppress generating non-timestamp tokens at the beginning
        if max_initial_index is not None:
            logits[:, TIMESTAMP_BEGIN + max_initial_index + 1:] = -float("inf")
    logprobs = torch.log_softmax(logits.float(), dim=-1)
    ts_lp = logprobs[:, TIMESTAMP_BEGIN:].logsumexp(dim=-1)
    max_text = logprobs[:, :TIMESTAMP_BEGIN].max(dim=-1).values
    force = ts_lp > max_text                                    # timestamp mass beats every text token: sample a timestamp
    logits[force, :TIMESTAMP_BEGIN] = -float("inf")


def _decode_general(model, mel: torch.Tensor, options: DecodingOptions):
    """Beam search / temperature sampling / timestamp rules (DecodingTask of whisper.decoding at token level).  Runs the
    cache-less step engine: every beam is a batch row and rows are re-gathered each step."""
    single = mel.dim() == 2
    if single:
        mel = mel[None]
    dims = model.dims
    V = dims.n_vocab
    n_audio = mel.shape[0]
    xa = model.embed_audio(mel) if mel.shape[-2:] == (dims.n_mels, 2 * dims.n_audio_ctx) else mel
    dev = xa.device
    init = list(options.initial_tokens) if options.initial_tokens is not None else (
        [SOT, NO_TIMESTAMPS] if options.without_timestamps else [SOT])
    sample_begin = len(init)
    sample_len = min(options.sample_len or dims.n_text_ctx // 2, dims.n_text_ctx - sample_begin)
    n_group = options.beam_size or options.best_of or 1
    beam = options.beam_size
    max_candidates = round(beam * (options.patience or 1.0)) if beam else None
    gen = None
    if options.temperature > 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(options.seed if options.seed is not None else 0)
    sup = options.suppress_mask.to(dev) if options.suppress_mask is not None else None
    max_init_idx = None
    if not options.without_timestamps and options.max_initial_timestamp is not None:
        max_init_idx = round(options.max_initial_timestamp / 0.02)

    xa_g = xa.repeat_interleave(n_group, dim=0)
    tokens = torch.tensor([init] * (n_audio * n_group), dtype=torch.int64, device=dev)
    sum_logprobs = torch.zeros(n_audio * n_group, device=dev)
    no_speech = [float("nan")] * n_audio
    finished = [dict() for _ in range(n_audio)] if beam else None

    for i in range(sample_len):
        if i == 0:  # no_speech_prob from the sot position of the first forward
            full = model.logits(tokens, xa_g)  # [n, len(init), V(+1)]
            sot_index = init.index(SOT) if SOT in init else 0
            p0 = torch.softmax(full[:, sot_index, :V].float(), dim=-1)[:, NO_SPEECH]
            no_speech = p0[::n_group].tolist()
            lg = full[:, -1, :V].float().clone()
        else:
            lg = model.logits(tokens, xa_g, last_only=True)[:, :V].float().clone()
        if sup is not None:
            lg = lg + sup
        if not options.without_timestamps:
            _timestamp_rules(lg, tokens, sample_begin, V, max_init_idx)
        if beam:
            logprobs = torch.log_softmax(lg, dim=-1)
            top_lp, top_tok = logprobs.topk(beam + 1, dim=-1)
            top_lp, top_tok = top_lp.cpu(), top_tok.cpu()
            prev_sum = sum_logprobs.cpu()
            tok_cpu = tokens.cpu()
            next_tokens, source, new_sum = [], [], []
            for a in range(n_audio):
                scores, sources, newly = {}, {}, {}
                for j in range(beam):
                    idx = a * beam + j
                    prefix = tok_cpu[idx].tolist()
                    for lp, t in zip(top_lp[idx].tolist(), top_tok[idx].tolist()):
                        seq = tuple(prefix + [t])
                        scores[seq] = float(prev_sum[idx]) + lp
                        sources[seq] = idx
                saved = 0
                for seq in sorted(scores, key=scores.get, reverse=True):
                    if seq[-1] == EOT:
                        newly[seq] = scores[seq]
                    else:
                        new_sum.append(scores[seq])
                        next_tokens.append(list(seq))
                        source.append(sources[seq])
                        saved += 1
                        if saved == beam:
                            break
                for seq in sorted(newly, key=newly.get, reverse=True):
                    if len(finished[a]) >= max_candidates:
                        break
                    finished[a][seq] = newly[seq]
            tokens = torch.tensor(next_tokens, dtype=torch.int64, device=dev)
            sum_logprobs = torch.tensor(new_sum, device=dev)
            completed = all(len(f) >= max_candidates for f in finished)
        else:
            if options.temperature == 0:
                nxt = lg.argmax(-1)
            else:
                probs = torch.softmax(lg / options.temperature, dim=-1)
                nxt = torch.multinomial(probs, 1, generator=gen)[:, 0]
            logprobs = torch.log_softmax(lg, dim=-1)
            cur = logprobs.gather(1, nxt[:, None])[:, 0]
            alive = tokens[:, -1] != EOT
            sum_logprobs = sum_logprobs + cur * alive
            nxt = torch.where(alive, nxt, torch.full_like(nxt, EOT))
            tokens = torch.cat([tokens, nxt[:, None]], dim=1)
            completed = bool((tokens[:, -1] == EOT).all())
        if completed or tokens.shape[-1] > dims.n_text_ctx:
            break

    # ---- finalize + rank (MaximumLikelihoodRanker): sum_logprob / length, or Google-NMT penalty when given
    def rank(cands):  # [(tokens after the prompt incl. eot, sum_logprob)]
        def score(c):
            length = len(c[0]) - (1 if c[0] and c[0][-1] == EOT else 0)  # whisper ranks on the tokens before eot
            pen = length if options.length_penalty is None else ((5 + length) / 6) ** options.length_penalty
            return c[1] / max(pen, 1e-6) if length else -float("inf")
        return max(cands, key=score)

    results = []
    tok_cpu, sums = tokens.cpu(), sum_logprobs.cpu()
    for a in range(n_audio):
        cands = []
        if beam:
            f = dict(finished[a])
            if len(f) < beam:  # close the best unfinished beams with eot
                order = sorted(range(beam), key=lambda j: float(sums[a * beam + j]), reverse=True)
                for j in order:
                    f[tuple(tok_cpu[a * beam + j].tolist() + [EOT])] = float(sums[a * beam + j])
                    if len(f) >= beam:
                        break
            cands = [(list(seq[sample_begin:]), lp) for seq, lp in f.items()]
        else:
            for j in range(n_group):
                row = tok_cpu[a * n_group + j, sample_begin:].tolist()
                if EOT in row:
                    row = row[:row.index(EOT) + 1]
                cands.append((row, float(sums[a * n_group + j])))
        best, lp = rank(cands)
        toks = best[:best.index(EOT)] if EOT in best else best
        results.append(DecodingResult(audio_features=xa[a], tokens=toks, avg_logprob=lp / (len(toks) + 1), no_speech_prob=no_speech[a],
                                      temperature=options.temperature))
    return results[0] if single else results


def greedy_token_matrix(model, mel: torch.Tensor, max_new: int, initial_tokens=(SOT, NO_TIMESTAMPS)) -> torch.Tensor:
    """[B, len(initial)+n] token matrix exactly as oracle.model_oracle.greedy_decode produces it (parity tests)."""
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
