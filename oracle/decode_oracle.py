"""CPU oracle for decoding and the long-form window driver.  TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu leg).

What is restated and from where.  The reference binds ``whisper.decoding.decode`` as ``OLMoASR.decode``
(olmoasr/model.py:966-968, inf_model.py:455-457) and drives it from ``olmoasr/transcribe.py:193-233`` (temperature
fallback) inside the seek loop ``olmoasr/transcribe.py:281-517``; evaluation uses beam 5 + the fallback tuple
(scripts/eval/eval.py:2077-2084), training-time evaluation greedy (scripts/training/train_timestamps.py:1916-1919).
``openai-whisper`` itself is an un-vendored, unpinned dependency (requirements.txt:21; the model code cites upstream commit
ba3f3cd54b0e5b8ce1ab3de13e32122d0d5f98ab): its ``DecodingTask`` -- GreedyDecoder, BeamSearchDecoder,
MaximumLikelihoodRanker, SuppressBlank, SuppressTokens, ApplyTimestampRules -- is restated here from the published
algorithm at TOKEN level (no tokenizer offline: text, the compression-ratio test and word timestamps are outside).
**Parity unpinned by the reference**: it ships neither tests nor golden vectors for decoding; this file is anchored on
the reference's call sites and on ``transcribe.py``, which IS in the reference tree and is followed line by line.

Everything runs on ``oracle.model_oracle`` (pinned to the unmodified reference model), plain Python + torch CPU, written
for clarity, independent of olmoasr_amd/*.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import model_oracle as mo

# English-only GPT-2 specials (consistent with train_timestamps.py:236,354,543)
EOT, SOT, TRANSLATE, TRANSCRIBE, SOT_LM, SOT_PREV, NO_SPEECH, NO_TIMESTAMPS, TIMESTAMP_BEGIN = (
    50256, 50257, 50357, 50358, 50359, 50360, 50361, 50362, 50363)
BLANK = 220  # GPT-2 BPE id of " " (tokenizer.encode(" ")), what SuppressBlank masks at the first sampled position
N_FRAMES, HOP_LENGTH, SAMPLE_RATE, FRAMES_PER_SECOND = 3000, 160, 16000, 100
# Tokenizer.non_speech_tokens on the GPT-2 English vocabulary (what suppress_tokens="-1" expands to).  The oracle's own copy of the
# constant, pinned in tests/test_decoding_rules_cpu.py against transformers' configuration_whisper.NON_SPEECH_TOKENS.
NON_SPEECH_EN = tuple(int(t) for t in """1 2 7 8 9 10 14 25 26 27 28 29 31 58 59 60 61 62 63 90 91 92 93 357 366 438 532 685 705 796 930 1058 1220
1267 1279 1303 1343 1377 1391 1635 1782 1875 2162 2361 2488 3467 4008 4211 4600 4808 5299 5855 6329 7203 9609 9959 10563 10786 11420 11709
11907 13163 13697 13700 14808 15306 16410 16791 17992 19203 19510 20724 22305 22935 27007 30109 30420 33409 34949 40283 40493 40549 47282
49146""".split())


@dataclass
class Options:  # whisper.decoding.DecodingOptions (token-level fields)
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    suppress_tokens: Optional[Sequence[int]] = (-1,)   # "-1": the non-speech symbol list + the specials
    non_speech_tokens: Sequence[int] = NON_SPEECH_EN   # tokenizer.non_speech_tokens
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    seed: int = 0
    logit_bias: Optional[torch.Tensor] = None          # additive [rows] (tests: steer a random-init model); product: suppress_mask


@dataclass
class Result:
    tokens: List[int] = field(default_factory=list)
    avg_logprob: float = float("nan")
    no_speech_prob: float = float("nan")
    temperature: float = 0.0
    sum_logprob: float = float("nan")
    # whisper.decoding.DecodingTask.run: text = tokenizer.decode(tokens).strip() (timestamps dropped), compression_ratio = gzip ratio
    # of it.  Filled only when a tokenizer is supplied (the package is un-vendored: there is none offline).
    text: str = ""
    compression_ratio: float = float("nan")


def compression_ratio(text: str) -> float:
    """whisper.utils.compression_ratio (published algorithm of the un-vendored dependency): len(utf-8 bytes) / len(zlib of them)."""
    import zlib
    b = text.encode("utf-8")
    return len(b) / len(zlib.compress(b))


def suppress_list(opt: Options) -> List[int]:
    """DecodingTask._get_suppress_tokens."""
    sup = list(opt.suppress_tokens) if opt.suppress_tokens is not None else []
    if -1 in sup:
        sup = [t for t in sup if t >= 0] + list(opt.non_speech_tokens)
    sup += [TRANSCRIBE, TRANSLATE, SOT, SOT_PREV, SOT_LM, NO_SPEECH]
    return sorted(set(sup))


def apply_timestamp_rules(logits: torch.Tensor, tokens: torch.Tensor, sample_begin: int, max_initial_index: Optional[int]):
    """whisper.decoding.ApplyTimestampRules.apply, in place on logits [n, V]."""
    logits[:, NO_TIMESTAMPS] = -math.inf
    for k in range(tokens.shape[0]):
        seq = tokens[k, sample_begin:].tolist()
        last_was_ts = len(seq) >= 1 and seq[-1] >= TIMESTAMP_BEGIN
        penultimate_was_ts = len(seq) < 2 or seq[-2] >= TIMESTAMP_BEGIN
        if last_was_ts:
            if penultimate_was_ts:
                logits[k, TIMESTAMP_BEGIN:] = -math.inf
            else:
                logits[k, :EOT] = -math.inf
        stamps = [t for t in seq if t >= TIMESTAMP_BEGIN]
        if stamps:
            last = stamps[-1] if (last_was_ts and not penultimate_was_ts) else stamps[-1] + 1
            logits[k, TIMESTAMP_BEGIN:last] = -math.inf
    if tokens.shape[1] == sample_begin:
        logits[:, :TIMESTAMP_BEGIN] = -math.inf
        if max_initial_index is not None:
            logits[:, TIMESTAMP_BEGIN + max_initial_index + 1:] = -math.inf
    logprobs = F.log_softmax(logits.float(), dim=-1)
    for k in range(tokens.shape[0]):
        if logprobs[k, TIMESTAMP_BEGIN:].logsumexp(-1) > logprobs[k, :TIMESTAMP_BEGIN].max():
            logits[k, :TIMESTAMP_BEGIN] = -math.inf


def decode(sd, dims, mel: torch.Tensor, opt: Options) -> List[Result]:
    """DecodingTask.run for a batch of 30 s windows mel [n_audio, 80, 3000] (no prompt / prefix: the reference has
    prompt conditioning commented out, transcribe.py:297-302)."""
    xa = mo.encoder_forward(sd, dims, mel)
    n_audio = mel.shape[0]
    init = [SOT, NO_TIMESTAMPS] if opt.without_timestamps else [SOT]
    sample_begin, sot_index = len(init), 0
    sample_len = opt.sample_len or dims.n_text_ctx // 2
    n_group = opt.beam_size or opt.best_of or 1
    sup = suppress_list(opt) if opt.suppress_tokens is not None else []
    max_initial_index = None
    if not opt.without_timestamps and opt.max_initial_timestamp is not None:
        max_initial_index = round(opt.max_initial_timestamp / 0.02)
    gen = torch.Generator().manual_seed(opt.seed)

    xa_g = xa.repeat_interleave(n_group, dim=0)
    tokens = torch.tensor([init] * (n_audio * n_group), dtype=torch.long)
    sum_logprobs = torch.zeros(n_audio * n_group)
    no_speech = [float("nan")] * n_audio
    finished: List[dict] = [dict() for _ in range(n_audio)]
    max_candidates = round(opt.beam_size * (opt.patience or 1.0)) if opt.beam_size else None

    for i in range(sample_len):
        full = mo.decoder_forward(sd, dims, tokens, xa_g)  # cache-less re-forward (notebooks/ow_decoding.py:42-72 style)
        if i == 0:
            probs = full[:, sot_index].float().softmax(-1)
            no_speech = probs[::n_group, NO_SPEECH].tolist()
        logits = full[:, -1].clone()
        if opt.logit_bias is not None:
            logits = logits + opt.logit_bias
        if opt.suppress_blank and tokens.shape[1] == sample_begin:
            logits[:, [BLANK, EOT]] = -math.inf
        if sup:
            logits[:, sup] = -math.inf
        if not opt.without_timestamps:
            apply_timestamp_rules(logits, tokens, sample_begin, max_initial_index)
        if opt.beam_size:  # BeamSearchDecoder.update
            logprobs = F.log_softmax(logits.float(), dim=-1)
            nxt, src, fin = [], [], []
            for a in range(n_audio):
                scores, sources, done = {}, {}, {}
                for j in range(opt.beam_size):
                    idx = a * opt.beam_size + j
                    prefix = tokens[idx].tolist()
                    top = logprobs[idx].topk(opt.beam_size + 1)
                    for lp, t in zip(top.values.tolist(), top.indices.tolist()):
                        seq = tuple(prefix + [t])
                        scores[seq] = float(sum_logprobs[idx]) + lp
                        sources[seq] = idx
                saved = 0
                for seq in sorted(scores, key=scores.get, reverse=True):
                    if seq[-1] == EOT:
                        done[seq] = scores[seq]
                    else:
                        sum_logprobs_new = scores[seq]
                        nxt.append(list(seq))
                        src.append((sources[seq], sum_logprobs_new))
                        saved += 1
                        if saved == opt.beam_size:
                            break
                fin.append(done)
            tokens = torch.tensor(nxt, dtype=torch.long)
            sum_logprobs = torch.tensor([s for _, s in src])
            for prev, new in zip(finished, fin):
                for seq in sorted(new, key=new.get, reverse=True):
                    if len(prev) >= max_candidates:
                        break
                    prev[seq] = new[seq]
            completed = all(len(f) >= max_candidates for f in finished)
        else:  # GreedyDecoder.update
            if opt.temperature == 0:
                nxt = logits.argmax(-1)
            else:
                nxt = torch.multinomial(F.softmax(logits.float() / opt.temperature, -1), 1, generator=gen)[:, 0]
            logprobs = F.log_softmax(logits.float(), dim=-1)
            cur = logprobs[torch.arange(len(nxt)), nxt]
            alive = tokens[:, -1] != EOT
            sum_logprobs = sum_logprobs + cur * alive
            nxt = torch.where(alive, nxt, torch.full_like(nxt, EOT))
            tokens = torch.cat([tokens, nxt[:, None]], dim=-1)
            completed = bool((tokens[:, -1] == EOT).all())
        if completed or tokens.shape[-1] > dims.n_text_ctx:
            break

    # finalize + MaximumLikelihoodRanker
    out = []
    for a in range(n_audio):
        if opt.beam_size:
            f = dict(finished[a])
            if len(f) < opt.beam_size:  # BeamSearchDecoder.finalize: add the best unfinished beams, closed with eot
                order = sorted(range(opt.beam_size), key=lambda j: float(sum_logprobs[a * opt.beam_size + j]), reverse=True)
                for j in order:
                    f[tuple(tokens[a * opt.beam_size + j].tolist() + [EOT])] = float(sum_logprobs[a * opt.beam_size + j])
                    if len(f) >= opt.beam_size:
                        break
            cands = [(list(seq), lp) for seq, lp in f.items()]
        else:
            cands = [(tokens[a * n_group + j].tolist() + [EOT], float(sum_logprobs[a * n_group + j])) for j in range(n_group)]
        cut = []
        for seq, lp in cands:
            body = seq[sample_begin:]
            cut.append((body[:body.index(EOT)], lp))

        def score(c):
            length = len(c[0])
            pen = length if opt.length_penalty is None else ((5 + length) / 6) ** opt.length_penalty
            return c[1] / pen if pen else -math.inf
        best = max(cut, key=score)
        out.append(Result(tokens=best[0], sum_logprob=best[1], avg_logprob=best[1] / (len(best[0]) + 1), no_speech_prob=no_speech[a],
                          temperature=opt.temperature))
    return out


PUNCTUATION = "\"'“¿([{-\"'.。,，!！?？:：”)]}、"  # olmoasr/transcribe.py:188


def get_end(segments):
    """whisper.utils.get_end (third party, imported at olmoasr/transcribe.py:28): end of the last word of the last segment that has any,
    else the end of the last segment, else None."""
    for s in reversed(segments):
        for w in reversed(s["words"]):
            return w["end"]
    return segments[-1]["end"] if segments else None


def word_anomaly_score(word) -> float:  # olmoasr/transcribe.py:323-333
    probability = word.get("probability", 0.0)
    duration = word["end"] - word["start"]
    score = 0.0
    if probability < 0.15:
        score += 1.0
    if duration < 0.133:
        score += (0.133 - duration) * 15
    if duration > 2.0:
        score += duration - 2.0
    return score


def is_segment_anomaly(segment) -> bool:  # olmoasr/transcribe.py:335-342
    if segment is None or not segment["words"]:
        return False
    words = [w for w in segment["words"] if w["word"] not in PUNCTUATION]
    words = words[:8]
    score = sum(word_anomaly_score(w) for w in words)
    return score >= 3 or score + 0.01 >= len(words)


def next_words_segment(segments):  # olmoasr/transcribe.py:344-345
    return next((s for s in segments if s["words"]), None)


def transcribe(sd, dims, mel_padded: torch.Tensor, *, temperature=(0.0, 0.2, 0.4, 0.6, 0.8, 1.0), logprob_threshold: Optional[float] = -1.0,
               no_speech_threshold: Optional[float] = 0.6, clip_timestamps: Sequence[float] = (0.0,), tokenizer=None,
               compression_ratio_threshold: Optional[float] = 2.4, initial_prompt: Optional[str] = None, word_timestamps: bool = False,
               add_word_timestamps=None, hallucination_silence_threshold: Optional[float] = None,
               prepend_punctuations: str = "\"'“¿([{-", append_punctuations: str = "\"'.。,，!！?？:：”)]}、", **decode_kw) -> dict:
    """olmoasr/transcribe.py:147-517.  ``mel_padded`` = log_mel_spectrogram(audio, padding=N_SAMPLES) [80, content_frames + 3000]
    (:148).  With ``tokenizer`` (decode / encode; the reference's comes from the un-vendored whisper package, :167-172) the text steps
    run too: the compression-ratio fallback (:213-217, on the ``compression_ratio`` the decode result carries), segment / result text
    (:266-279, :519-523), the empty-text rule (:494-499), initial_prompt (:258-264).  Without one: token level, the compression test
    skipped.  ``word_timestamps`` (:409-486): ``add_word_timestamps`` is the callable the reference imports from whisper.timing (here a
    parameter: the pin scripts it), followed by the seek-to-last-word rule and the ``hallucination_silence_threshold`` skipping rules.
    Prompt conditioning is commented out in the reference (:297-302)."""
    content_frames = mel_padded.shape[-1] - N_FRAMES
    content_duration = float(content_frames * HOP_LENGTH / SAMPLE_RATE)
    last_speech_timestamp = 0.0
    seek_points = [round(ts * FRAMES_PER_SECOND) for ts in clip_timestamps] or [0]
    if len(seek_points) % 2 == 1:
        seek_points.append(content_frames)
    seek_clips = list(zip(seek_points[::2], seek_points[1::2]))
    temps = [temperature] if isinstance(temperature, (int, float)) else list(temperature)
    input_stride, time_precision = 2, 0.02

    def decode_with_fallback(segment):  # :193-233
        res = None
        for t in temps:
            kw = dict(decode_kw)
            if t > 0:
                kw.pop("beam_size", None)
                kw.pop("patience", None)
            else:
                kw.pop("best_of", None)
            res = decode(sd, dims, segment[None], Options(temperature=t, **kw))[0]
            needs_fallback = False
            if tokenizer is not None and compression_ratio_threshold is not None and res.compression_ratio > compression_ratio_threshold:
                needs_fallback = True  # too repetitive (:213-217)
            if logprob_threshold is not None and res.avg_logprob < logprob_threshold:
                needs_fallback = True  # (:218-222)
            if (no_speech_threshold is not None and res.no_speech_prob > no_speech_threshold and logprob_threshold is not None
                    and res.avg_logprob < logprob_threshold):
                needs_fallback = False
            if not needs_fallback:
                break
        return res

    all_tokens, all_segments, seeks = [], [], []
    prompt_tokens = list(tokenizer.encode(" " + initial_prompt.strip())) if (tokenizer is not None and initial_prompt is not None) else []
    all_tokens.extend(prompt_tokens)  # (:258-264)
    clip_idx, seek = 0, seek_clips[0][0]
    while clip_idx < len(seek_clips):
        clip_start, clip_end = seek_clips[clip_idx]
        if seek < clip_start:
            seek = clip_start
        if seek >= clip_end:
            clip_idx += 1
            if clip_idx < len(seek_clips):
                seek = seek_clips[clip_idx][0]
            continue
        seeks.append(seek)
        time_offset = seek * HOP_LENGTH / SAMPLE_RATE
        window_end_time = float((seek + N_FRAMES) * HOP_LENGTH / SAMPLE_RATE)
        segment_size = min(N_FRAMES, content_frames - seek, clip_end - seek)
        seg = mel_padded[:, seek:seek + segment_size]
        segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE
        seg = F.pad(seg, (0, N_FRAMES - seg.shape[-1]))  # pad_or_trim: literal zeros (:295)
        result = decode_with_fallback(seg)
        tokens = result.tokens
        if no_speech_threshold is not None:
            should_skip = result.no_speech_prob > no_speech_threshold
            if logprob_threshold is not None and result.avg_logprob > logprob_threshold:
                should_skip = False
            if should_skip:
                seek += segment_size
                continue
        previous_seek = seek
        current = []

        def new_segment(start, end, toks):
            seg = {"seek": seek, "start": start, "end": end, "tokens": list(toks), "temperature": result.temperature,
                   "avg_logprob": result.avg_logprob, "no_speech_prob": result.no_speech_prob}
            if tokenizer is not None:  # (:266-279)
                seg["text"] = tokenizer.decode([t for t in toks if t < EOT])
                seg["compression_ratio"] = result.compression_ratio
            return seg
        is_ts = [t >= TIMESTAMP_BEGIN for t in tokens]
        single_timestamp_ending = is_ts[-2:] == [False, True]
        consecutive = [i + 1 for i in range(len(tokens) - 1) if is_ts[i] and is_ts[i + 1]]
        if consecutive:
            slices = list(consecutive)
            if single_timestamp_ending:
                slices.append(len(tokens))
            last_slice = 0
            for cur in slices:
                sl = tokens[last_slice:cur]
                current.append(new_segment(time_offset + (sl[0] - TIMESTAMP_BEGIN) * time_precision,
                                           time_offset + (sl[-1] - TIMESTAMP_BEGIN) * time_precision, sl))
                last_slice = cur
            if single_timestamp_ending:
                seek += segment_size
            else:
                seek += (tokens[last_slice - 1] - TIMESTAMP_BEGIN) * input_stride
        else:
            duration = segment_duration
            stamps = [t for t in tokens if t >= TIMESTAMP_BEGIN]
            if stamps and stamps[-1] != TIMESTAMP_BEGIN:
                duration = (stamps[-1] - TIMESTAMP_BEGIN) * time_precision
            current.append(new_segment(time_offset, time_offset + duration, tokens))
            seek += segment_size
        if word_timestamps:  # (:409-486)
            add_word_timestamps(segments=current, model=None, tokenizer=tokenizer, mel=seg, num_frames=segment_size,
                                prepend_punctuations=prepend_punctuations, append_punctuations=append_punctuations,
                                last_speech_timestamp=last_speech_timestamp)
            if not single_timestamp_ending:
                last_word_end = get_end(current)
                if last_word_end is not None and last_word_end > time_offset:
                    seek = round(last_word_end * FRAMES_PER_SECOND)
            skip_window = False
            if hallucination_silence_threshold is not None:
                threshold = hallucination_silence_threshold
                if not single_timestamp_ending:
                    last_word_end = get_end(current)
                    if last_word_end is not None and last_word_end > time_offset:
                        remaining_duration = window_end_time - last_word_end
                        if remaining_duration > threshold:
                            seek = round(last_word_end * FRAMES_PER_SECOND)
                        else:
                            seek = previous_seek + segment_size
                first_segment = next_words_segment(current)
                if first_segment is not None and is_segment_anomaly(first_segment):
                    gap = first_segment["start"] - time_offset
                    if gap > threshold:
                        seek = previous_seek + round(gap * FRAMES_PER_SECOND)
                        skip_window = True
                if not skip_window:
                    hal_last_end = last_speech_timestamp
                    for si in range(len(current)):
                        segment = current[si]
                        if not segment["words"]:
                            continue
                        if is_segment_anomaly(segment):
                            next_segment = next_words_segment(current[si + 1:])
                            if next_segment is not None:
                                hal_next_start = next_segment["words"][0]["start"]
                            else:
                                hal_next_start = time_offset + segment_duration
                            silence_before = (segment["start"] - hal_last_end > threshold or segment["start"] < threshold
                                              or segment["start"] - time_offset < 2.0)
                            silence_after = (hal_next_start - segment["end"] > threshold or is_segment_anomaly(next_segment)
                                             or window_end_time - segment["end"] < 2.0)
                            if silence_before and silence_after:
                                seek = round(max(time_offset + 1, segment["start"]) * FRAMES_PER_SECOND)
                                if content_duration - segment["end"] < threshold:
                                    seek = content_frames
                                current[si:] = []
                                break
                        hal_last_end = segment["end"]
            if skip_window:
                continue  # (:441-443: the window is decoded again from after the leading silence)
            last_word_end = get_end(current)
            if last_word_end is not None:
                last_speech_timestamp = last_word_end
        for s in current:  # "instantaneous or does not contain text" (:494-499); without a tokenizer text == tokens below eot
            empty = s["text"].strip() == "" if tokenizer is not None else not any(t < EOT for t in s["tokens"])
            if s["start"] == s["end"] or empty:
                s["tokens"] = []
                s["words"] = []
                if tokenizer is not None:
                    s["text"] = ""
        base_id = len(all_segments)  # ids continue across windows (:501-508); evaluated before the list grows
        all_segments.extend({"id": base_id + i, **s} for i, s in enumerate(current))
        all_tokens.extend(t for s in current for t in s["tokens"])
    out = {"tokens": all_tokens[len(prompt_tokens):], "segments": all_segments, "seeks": seeks}
    if tokenizer is not None:
        out["text"] = tokenizer.decode(all_tokens[len(prompt_tokens):])  # (:519-523)
    return out
