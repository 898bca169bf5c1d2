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
NO_TIMESTAMPS = 50362


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
    best_of: Optional[int] = None
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
    if options.temperature != 0.0 or options.beam_size or options.best_of:
        raise NotImplementedError("only greedy decoding (temperature 0, no beam) is implemented on the native path")
    if not options.without_timestamps:
        raise NotImplementedError("timestamp rules need the tokenizer's timestamp ids; decode without_timestamps=True")
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
    n_sampled = torch.zeros(B, device=xa.device)
    sup = options.suppress_mask.to(xa.device) if options.suppress_mask is not None else None
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
        n_sampled += (~done).float()
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
        results.append(DecodingResult(audio_features=xa[b], tokens=row, avg_logprob=float(sum_logprobs[b] / n_sampled[b].clamp(min=1)),
                                      temperature=0.0))
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
