"""Word-level timestamps: what ``olmoasr/transcribe.py:409-419`` calls as ``whisper.timing.add_word_timestamps`` when
``transcribe(word_timestamps=True)``.

The reference takes the function from the third-party ``openai-whisper`` package (unpinned: requirements.txt:21; not installable here);
this module restates that package's published algorithm (``whisper/timing.py``: ``find_alignment``, ``median_filter``, ``dtw``,
``merge_punctuations``, ``add_word_timestamps``) on top of this repository's model:

  1. teacher-force ``<sot..> <notimestamps> text... <eot>`` through the decoder and collect the CROSS-attention score matrices ``qk`` of the
     alignment heads (whisper's default when a checkpoint names none: every head of the upper half of the decoder layers);
  2. softmax over the audio frames, normalise each head over the token axis, median-filter along time (width 7), average the heads;
  3. dynamic time warping of tokens against frames on the negated matrix; the frame at which the path moves to the next token is that
     token's start time (20 ms per frame pair);
  4. group tokens into words with the tokenizer, merge punctuation into its neighbours, clip implausibly long words at sentence and
     segment boundaries, and write ``segment["words"]`` / tighten ``segment["start"]`` / ``segment["end"]``.

With the REFERENCE's model class step 1 cannot work: its cross-attention runs through ``F.scaled_dot_product_attention`` and returns
``qk = None`` (olmoasr/model.py:313, 328-345), and the class has no ``alignment_heads`` -- ``word_timestamps=True`` fails there.  Here the
score matrix is computed on request (``MultiHeadAttention.return_qk`` -> ``oasr_attention_scores``, csrc/scores.hip), so the option works.

The tokenizer is a plug with whisper's attribute names: ``sot_sequence``, ``no_timestamps``, ``eot``, ``split_to_word_tokens(tokens) ->
(words, word_tokens)``.
"""
import itertools
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

HOP_LENGTH, SAMPLE_RATE = 160, 16000
TOKENS_PER_SECOND = SAMPLE_RATE // HOP_LENGTH // 2  # 20 ms per audio token (two mel frames)


@dataclass
class WordTiming:
    word: str
    tokens: List[int]
    start: float
    end: float
    probability: float


def median_filter(x: torch.Tensor, filter_width: int) -> torch.Tensor:
    """Median over a sliding window of ``filter_width`` (odd) along the last axis, reflect-padded at both ends."""
    assert filter_width > 0 and filter_width % 2 == 1, "`filter_width` should be an odd number"
    pad = filter_width // 2
    if x.shape[-1] <= pad:  # (reflect padding needs pad < length)
        return x
    shape = x.shape
    y = torch.nn.functional.pad(x.reshape(1, -1, shape[-1]), (pad, pad), mode="reflect")
    return y.unfold(-1, filter_width, 1).sort()[0][..., pad].reshape(shape)


def dtw(x: np.ndarray):
    """Monotonic alignment of rows (tokens) to columns (frames) minimising the summed cost ``x``: returns (row indices, column indices)
    of the path from (0, 0) to (N-1, M-1); steps are diagonal, down (next token, same frame) or right (same token, next frame).
    Ties resolve as in whisper's reference implementation: diagonal only when strictly cheapest, else down only when strictly cheapest,
    else right.  Anti-diagonals are independent, so each is one vectorised numpy step."""
    x = np.asarray(x, dtype=np.float64)
    N, M = x.shape
    cost = np.full((N + 1, M + 1), np.inf, dtype=np.float32)
    trace = np.full((N + 1, M + 1), -1, dtype=np.int8)
    cost[0, 0] = 0
    xf = x.astype(np.float32)
    for d in range(2, N + M + 1):
        i = np.arange(max(1, d - M), min(N, d - 1) + 1)
        j = d - i
        c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
        t = np.full(i.shape, 2, dtype=np.int8)
        t[(c1 < c0) & (c1 < c2)] = 1
        t[(c0 < c1) & (c0 < c2)] = 0
        c = np.where(t == 0, c0, np.where(t == 1, c1, c2))
        cost[i, j] = xf[i - 1, j - 1] + c
        trace[i, j] = t
    trace[0, :] = 2
    trace[:, 0] = 1
    i, j, path = N, M, []
    while i > 0 or j > 0:
        path.append((i - 1, j - 1))
        t = trace[i, j]
        if t == 0:
            i, j = i - 1, j - 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    path = np.array(path[::-1], dtype=np.int64).T
    return path[0], path[1]


def alignment_heads(model):
    """(layer, head) pairs whose cross-attention is averaged: ``model.alignment_heads`` (a bool [n_text_layer, n_text_head] mask, set by
    ``set_alignment_heads`` where a checkpoint provides one) or whisper's default -- all heads of the upper half of the decoder."""
    dims = model.dims
    mask = getattr(model, "alignment_heads", None)
    if mask is None:
        mask = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
        mask[dims.n_text_layer // 2:] = True
    if mask.is_sparse:
        mask = mask.to_dense()
    return [(int(l), int(h)) for l, h in mask.nonzero().tolist()]


@torch.no_grad()
def cross_attention_scores(model, tokens: torch.Tensor, xa: torch.Tensor, layers):
    """Teacher-forced decoder pass, module by module, returning {layer: qk fp32 [H, n_tokens, n_audio_ctx]} for ``layers`` -- the forward
    hooks whisper installs on ``block.cross_attn`` (``outs[-1]``), with the score matrix switched on only for the layers that are read."""
    dec = model.decoder
    n = tokens.shape[-1]
    x = (dec.token_embedding.weight.detach()[tokens] + dec.positional_embedding.detach()[:n]).to(torch.bfloat16)[None]
    causal = torch.full((n, n), -float("inf"), device=x.device).triu_(1)
    got, hooks = {}, []
    for i, blk in enumerate(dec.blocks):
        if i in layers:
            blk.cross_attn.return_qk = True
            hooks.append(blk.cross_attn.register_forward_hook(lambda _m, _i, outs, index=i: got.__setitem__(index, outs[-1][0])))
    try:
        for i, blk in enumerate(dec.blocks):
            if i > max(layers):
                break  # nothing above the last alignment layer is read
            x = blk(x, xa, mask=causal)
    finally:
        for h in hooks:
            h.remove()
        for i in layers:
            del dec.blocks[i].cross_attn.return_qk  # back to the class default
    return got


@torch.no_grad()
def find_alignment(model, tokenizer, text_tokens: List[int], mel: torch.Tensor, num_frames: int, *, medfilt_width: int = 7,
                   qk_scale: float = 1.0) -> List[WordTiming]:
    if len(text_tokens) == 0:
        return []
    n_sot = len(tokenizer.sot_sequence)
    tokens = torch.tensor([*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot], device=model.device)
    xa = model.embed_audio(mel[None] if mel.dim() == 2 else mel)
    logits = model.logits(tokens[None], xa)[0]
    probs = logits[n_sot:, : tokenizer.eot].float().softmax(dim=-1)
    text_token_probs = probs[torch.arange(len(text_tokens)), torch.tensor(text_tokens, device=probs.device)].tolist()

    heads = alignment_heads(model)
    qks = cross_attention_scores(model, tokens, xa, sorted({l for l, _ in heads}))
    weights = torch.stack([qks[l][h] for l, h in heads])            # heads x tokens x frames
    weights = weights[:, :, : num_frames // 2]
    weights = (weights * qk_scale).softmax(dim=-1)
    std, mean = torch.std_mean(weights, dim=-2, keepdim=True, unbiased=False)
    weights = (weights - mean) / std
    weights = median_filter(weights, medfilt_width)
    matrix = weights.mean(dim=0)[n_sot:-1]                           # text tokens (+ <notimestamps>) x frames
    text_indices, time_indices = dtw(-matrix.double().cpu().numpy())

    words, word_tokens = tokenizer.split_to_word_tokens(text_tokens + [tokenizer.eot])
    if len(word_tokens) <= 1:  # only the eot "word": nothing to time
        return []
    word_boundaries = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    jump_times = time_indices[jumps] / TOKENS_PER_SECOND
    start_times = jump_times[word_boundaries[:-1]]
    end_times = jump_times[word_boundaries[1:]]
    word_probs = [float(np.mean(text_token_probs[i:j])) for i, j in zip(word_boundaries[:-1], word_boundaries[1:])]
    return [WordTiming(w, t, float(s), float(e), p) for w, t, s, e, p in zip(words, word_tokens, start_times, end_times, word_probs)]


def merge_punctuations(alignment: List[WordTiming], prepended: str, appended: str) -> None:
    """Opening punctuation joins the word after it (right to left), closing punctuation the word before it (left to right); the absorbed
    entries stay in the list with an empty word and no tokens."""
    i, j = len(alignment) - 2, len(alignment) - 1
    while i >= 0:
        prev, nxt = alignment[i], alignment[j]
        if prev.word.startswith(" ") and prev.word.strip() in prepended:
            nxt.word, nxt.tokens = prev.word + nxt.word, prev.tokens + nxt.tokens
            prev.word, prev.tokens = "", []
        else:
            j = i
        i -= 1
    i, j = 0, 1
    while j < len(alignment):
        prev, nxt = alignment[i], alignment[j]
        if not prev.word.endswith(" ") and nxt.word in appended:
            prev.word, prev.tokens = prev.word + nxt.word, prev.tokens + nxt.tokens
            nxt.word, nxt.tokens = "", []
        else:
            i = j
        j += 1


def add_word_timestamps(*, segments: List[dict], model, tokenizer, mel: torch.Tensor, num_frames: int,
                        prepend_punctuations: str = "\"'“¿([{-", append_punctuations: str = "\"'.。,，!！?？:：”)]}、",
                        last_speech_timestamp: float, **kwargs) -> None:
    """Fills ``segment["words"]`` = [{word, start, end, probability}] for the segments of ONE window (keyword signature of the call at
    olmoasr/transcribe.py:410-419) and pulls the segments' own start / end onto their first / last word."""
    if len(segments) == 0:
        return
    per_segment = [[t for t in s["tokens"] if t < tokenizer.eot] for s in segments]
    text_tokens = list(itertools.chain.from_iterable(per_segment))
    alignment = find_alignment(model, tokenizer, text_tokens, mel, num_frames, **kwargs)
    durations = np.array([t.end - t.start for t in alignment])
    durations = durations[durations.nonzero()]
    median_duration = min(0.7, float(np.median(durations))) if len(durations) > 0 else 0.0
    max_duration = median_duration * 2
    if len(durations) > 0:  # words at sentence boundaries are not allowed to run longer than twice the median word
        marks = ".。!！?？"
        for i in range(1, len(alignment)):
            if alignment[i].end - alignment[i].start > max_duration:
                if alignment[i].word in marks:
                    alignment[i].end = alignment[i].start + max_duration
                elif alignment[i - 1].word in marks:
                    alignment[i].start = alignment[i].end - max_duration
    merge_punctuations(alignment, prepend_punctuations, append_punctuations)

    time_offset = segments[0]["seek"] * HOP_LENGTH / SAMPLE_RATE
    wi = 0
    for segment, toks in zip(segments, per_segment):
        saved, words = 0, []
        while wi < len(alignment) and saved < len(toks):
            t = alignment[wi]
            if t.word:
                words.append(dict(word=t.word, start=round(time_offset + t.start, 2), end=round(time_offset + t.end, 2), probability=t.probability))
            saved += len(t.tokens)
            wi += 1
        if len(words) > 0:
            # the first (and second) word after a pause must not be longer than twice the median word
            if words[0]["end"] - last_speech_timestamp > median_duration * 4 and (
                    words[0]["end"] - words[0]["start"] > max_duration
                    or (len(words) > 1 and words[1]["end"] - words[0]["start"] > max_duration * 2)):
                if len(words) > 1 and words[1]["end"] - words[1]["start"] > max_duration:
                    boundary = max(words[1]["end"] / 2, words[1]["end"] - max_duration)
                    words[0]["end"] = words[1]["start"] = boundary
                words[0]["start"] = max(0, words[0]["end"] - max_duration)
            # the decoder's own segment boundaries win where the first / last word is implausibly long
            if segment["start"] < words[0]["end"] and segment["start"] - 0.5 > words[0]["start"]:
                words[0]["start"] = max(0, min(words[0]["end"] - median_duration, segment["start"]))
            else:
                segment["start"] = words[0]["start"]
            if segment["end"] > words[-1]["start"] and segment["end"] + 0.5 < words[-1]["end"]:
                words[-1]["end"] = max(words[-1]["start"] + median_duration, segment["end"])
            else:
                segment["end"] = words[-1]["end"]
            last_speech_timestamp = segment["end"]
        segment["words"] = words
