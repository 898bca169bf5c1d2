"""Long-form transcription driver on the native engine: ``OLMoASR.transcribe`` (reference olmoasr/transcribe.py:47-523).

Same signature, defaults and control flow as the reference.  Text is a PLUG (``tokenizer=``, ``decoding.resolve_tokenizer``): the
reference takes its tokenizer from the un-vendored openai-whisper package (:167-172), which is absent offline.  With a tokenizer
(whisper's own is picked up automatically where the package is installed) every text-dependent step of the reference runs:
``text`` of each segment and of the result, the ``compression_ratio_threshold`` fallback (:213-217: "too repetitive" -> next
temperature), the "instantaneous or no text" rule on the decoded string (:494-499), ``initial_prompt`` (:258-264), ``verbose``
printing (:488-492).  Without one the same loop runs at TOKEN level: ``text`` is None, the compression-ratio test is skipped and a
segment counts as empty when it holds no token below eot.  ``word_timestamps`` (:409-424) runs ``olmoasr_amd.timing.add_word_timestamps``
-- cross-attention scores on request + DTW, the published algorithm of the ``whisper.timing`` function the reference imports -- and
``hallucination_silence_threshold`` (:426-486) the reference's own skipping rules on the words it returns; both need a tokenizer (words
are a property of the text).  With the reference's model class the option cannot work (its cross-attention returns ``qk = None``):

  * whole-file log-mel once with ``padding=N_SAMPLES`` (:148), ``content_frames = n_frames - 3000`` (:149)
  * ``clip_timestamps`` -> seek clips (:177-186); window = ``mel[:, seek : seek + segment_size]`` with
    ``segment_size = min(3000, content_frames - seek, clip_end - seek)``, zero-padded to 3000 frames (:292-295) -- the
    padding is literal 0.0, not the log-mel silence floor
  * ``decode_with_fallback`` (:193-233): temperatures in turn while ``avg_logprob < logprob_threshold`` (beam options at
    temperature 0 only, ``best_of`` above it); a window whose ``no_speech_prob > no_speech_threshold`` and low logprob is
    silence, not a failure
  * no-speech skip (:305-320), then the TIMESTAMP-DRIVEN SEEK (:348-408): consecutive timestamp-token pairs cut the window
    into segments and ``seek`` moves to the last closed timestamp (or by the whole window when the output ends on a single
    timestamp / has no pairs); segments that are instantaneous or hold no text tokens are cleared (:494-499)
  * prompt conditioning is commented out in the reference (:297-302) and absent here.

With ``without_timestamps=True`` the seek always advances by a full window, so windows are independent and are decoded
``batch_windows`` at a time (an MI355X-side batching the reference's one-window loop cannot do); with timestamps the loop is
sequential, as in the reference.
"""
import warnings
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .audio import FRAMES_PER_SECOND, HOP_LENGTH, N_FRAMES, N_SAMPLES, SAMPLE_RATE, log_mel_spectrogram, pad_or_trim
from .decoding import EOT, TIMESTAMP_BEGIN, DecodingOptions, DecodingResult, decode, resolve_tokenizer
from .timing import add_word_timestamps

PUNCTUATION = "\"'“¿([{-\"'.。,，!！?？:：”)]}、"  # (:188)


def get_end(segments: List[dict]) -> Optional[float]:
    """whisper.utils.get_end: the end of the last word of the last segment that has words, else the last segment's own end."""
    return next((w["end"] for s in reversed(segments) for w in reversed(s["words"])), segments[-1]["end"] if segments else None)


def word_anomaly_score(word: dict) -> float:
    """(:323-333) anomalous words are improbable, very short or very long"""
    probability, duration = word.get("probability", 0.0), word["end"] - word["start"]
    return (1.0 if probability < 0.15 else 0.0) + ((0.133 - duration) * 15 if duration < 0.133 else 0.0) + (duration - 2.0 if duration > 2.0 else 0.0)


def is_segment_anomaly(segment: Optional[dict]) -> bool:
    """(:335-342) over the first eight non-punctuation words"""
    if segment is None or not segment["words"]:
        return False
    words = [w for w in segment["words"] if w["word"] not in PUNCTUATION][:8]
    score = sum(word_anomaly_score(w) for w in words)
    return score >= 3 or score + 0.01 >= len(words)


def next_words_segment(segments: List[dict]) -> Optional[dict]:
    return next((s for s in segments if s["words"]), None)


def format_timestamp(seconds: float) -> str:
    """whisper.utils.format_timestamp with its defaults (mm:ss.mmm, hours only when needed) -- the verbose line of :488-492."""
    ms = round(seconds * 1000.0)
    h, ms = divmod(ms, 3_600_000)
    m, ms = divmod(ms, 60_000)
    sec, ms = divmod(ms, 1000)
    return (f"{h:02d}:" if h > 0 else "") + f"{m:02d}:{sec:02d}.{ms:03d}"


@torch.no_grad()
def transcribe(model, audio, *, verbose: Optional[bool] = None,
               temperature: Union[float, Tuple[float, ...]] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
               compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
               no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
               initial_prompt: Optional[str] = None, carry_initial_prompt: bool = False, word_timestamps: bool = False,
               prepend_punctuations: str = "\"'“¿([{-", append_punctuations: str = "\"'.。,，!！?？:：”)]}、",
               clip_timestamps: Union[str, Sequence[float]] = "0",
               hallucination_silence_threshold: Optional[float] = None, batch_windows: int = 16, tokenizer=None, **decode_options):
    if isinstance(audio, str):
        from .audio import load_audio
        audio = load_audio(audio)
    if hallucination_silence_threshold is not None and not word_timestamps:  # (the reference's CLI says the same, :575; the loop reads it under word_timestamps only, :409-428)
        warnings.warn("hallucination_silence_threshold is ignored without word_timestamps")
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.ascontiguousarray(audio))
    mel = log_mel_spectrogram(audio, model.dims.n_mels, padding=N_SAMPLES, device=model.device)  # [80, content + 3000]
    content_frames = mel.shape[-1] - N_FRAMES
    content_duration = float(content_frames * HOP_LENGTH / SAMPLE_RATE)
    if decode_options.get("language", None) is None:
        decode_options["language"] = "en"  # not model.is_multilingual (:152-154)
    if isinstance(clip_timestamps, str):
        clip_timestamps = [float(ts) for ts in (clip_timestamps.split(",") if clip_timestamps else [])]
    seek_points: List[int] = [round(ts * FRAMES_PER_SECOND) for ts in clip_timestamps]
    if len(seek_points) == 0:
        seek_points.append(0)
    if len(seek_points) % 2 == 1:
        seek_points.append(content_frames)
    seek_clips = list(zip(seek_points[::2], seek_points[1::2]))
    temperatures = [temperature] if isinstance(temperature, (int, float)) else list(temperature)
    tokenizer = resolve_tokenizer(model, tokenizer, decode_options["language"], decode_options.get("task", "transcribe"))  # (:167-172)
    if tokenizer is None and initial_prompt is not None:
        warnings.warn("initial_prompt needs a tokenizer (tokenizer=...): ignored")
    if word_timestamps and tokenizer is None:
        raise ValueError("word_timestamps=True needs a tokenizer (tokenizer=...): words are a property of the decoded text "
                         "(tokenizer.split_to_word_tokens); whisper's own is used where the package is installed")
    if word_timestamps and decode_options.get("task", "transcribe") == "translate":  # (:174-175)
        warnings.warn("Word-level timestamps on translations may not be reliable.")
    initial_prompt_tokens: List[int] = []
    if tokenizer is not None and initial_prompt is not None:  # (:258-264; the prompt conditioning itself is commented out in the reference)
        initial_prompt_tokens = list(tokenizer.encode(" " + initial_prompt.strip()))
    input_stride = N_FRAMES // model.dims.n_audio_ctx          # mel frames per output token: 2
    time_precision = input_stride * HOP_LENGTH / SAMPLE_RATE   # 0.02 s

    def decode_with_fallback(segments: torch.Tensor) -> List[DecodingResult]:
        """:193-233 for a batch of windows: each temperature is tried on the windows still failing."""
        results: List[Optional[DecodingResult]] = [None] * segments.shape[0]
        todo = list(range(segments.shape[0]))
        for t in temperatures:
            kwargs = {**decode_options}
            if t > 0:
                kwargs.pop("beam_size", None)   # sampling temperatures run without the beam (:201-204)
                kwargs.pop("patience", None)
            else:
                kwargs.pop("best_of", None)     # and the greedy/beam pass without best_of (:205-207)
            out = decode(model, segments[todo], DecodingOptions(**kwargs, temperature=t), tokenizer=tokenizer)
            still = []
            for j, r in zip(todo, out):
                results[j] = r
                needs_fallback = False
                if tokenizer is not None and compression_ratio_threshold is not None and r.compression_ratio > compression_ratio_threshold:
                    needs_fallback = True       # too repetitive (:213-217)
                if logprob_threshold is not None and r.avg_logprob < logprob_threshold:
                    needs_fallback = True       # average log probability is too low (:218-222)
                if (no_speech_threshold is not None and r.no_speech_prob > no_speech_threshold and logprob_threshold is not None
                        and r.avg_logprob < logprob_threshold):
                    needs_fallback = False      # a quiet window is accepted as it is (:223-229)
                if needs_fallback:
                    still.append(j)
            todo = still
            if not todo:
                break
        return results

    def window(seek: int, clip_end: int):
        segment_size = min(N_FRAMES, content_frames - seek, clip_end - seek)
        return pad_or_trim(mel[:, seek:seek + segment_size], N_FRAMES), segment_size

    all_tokens: List[int] = list(initial_prompt_tokens)
    all_segments: List[dict] = []
    independent = bool(decode_options.get("without_timestamps", False)) and not word_timestamps  # (word timing moves the seek)
    last_speech_timestamp = 0.0
    clip_idx = 0
    seek = seek_clips[clip_idx][0]
    pending: List[Tuple[int, int, DecodingResult]] = []  # (seek, segment_size, result) decoded ahead (independent windows only)
    while clip_idx < len(seek_clips):
        seek_clip_start, seek_clip_end = seek_clips[clip_idx]
        if seek < seek_clip_start:
            seek = seek_clip_start
        if seek >= seek_clip_end:
            clip_idx += 1
            if clip_idx < len(seek_clips):
                seek = seek_clips[clip_idx][0]
            continue
        if pending and pending[0][0] == seek:
            _, segment_size, result = pending.pop(0)
        else:
            pending = []
            ahead = [seek]
            if independent:  # without timestamps every window advances by segment_size: decode a batch of them at once
                while len(ahead) < batch_windows and ahead[-1] + N_FRAMES < seek_clip_end and ahead[-1] + N_FRAMES < content_frames:
                    ahead.append(ahead[-1] + N_FRAMES)
            wins = [window(s, seek_clip_end) for s in ahead]
            res = decode_with_fallback(torch.stack([w for w, _ in wins]))
            pending = [(s, sz, r) for s, (_, sz), r in zip(ahead, wins, res)]
            _, segment_size, result = pending.pop(0)
        time_offset = float(seek * HOP_LENGTH / SAMPLE_RATE)
        window_end_time = float((seek + N_FRAMES) * HOP_LENGTH / SAMPLE_RATE)
        segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE
        tokens = list(result.tokens)

        if no_speech_threshold is not None:  # windows judged silent are stepped over whole (:305-320)
            should_skip = result.no_speech_prob > no_speech_threshold
            if logprob_threshold is not None and result.avg_logprob > logprob_threshold:
                should_skip = False          # ... unless the decoder was confident about what it wrote
            if should_skip:
                seek += segment_size
                continue

        previous_seek = seek
        current_segments: List[dict] = []

        def new_segment(*, start: float, end: float, toks: List[int]):
            text = tokenizer.decode([t for t in toks if t < EOT]) if tokenizer is not None else None  # (:266-279)
            return {"seek": seek, "start": start, "end": end, "text": text, "tokens": list(toks), "temperature": result.temperature,
                    "avg_logprob": result.avg_logprob, "compression_ratio": result.compression_ratio,
                    "no_speech_prob": result.no_speech_prob}

        is_ts = [t >= TIMESTAMP_BEGIN for t in tokens]
        single_timestamp_ending = is_ts[-2:] == [False, True]
        consecutive = [i + 1 for i in range(len(tokens) - 1) if is_ts[i] and is_ts[i + 1]]
        if len(consecutive) > 0:  # <|t|><|t'|> pairs close segments (:348-386)
            slices = list(consecutive)
            if single_timestamp_ending:
                slices.append(len(tokens))
            last_slice = 0
            for current_slice in slices:
                sliced = tokens[last_slice:current_slice]
                current_segments.append(new_segment(start=time_offset + (sliced[0] - TIMESTAMP_BEGIN) * time_precision,
                                                    end=time_offset + (sliced[-1] - TIMESTAMP_BEGIN) * time_precision, toks=sliced))
                last_slice = current_slice
            if single_timestamp_ending:
                seek += segment_size  # output closed by a lone timestamp: the rest of the window holds nothing
            else:                     # open tail: resume from the last closed timestamp, the tail is decoded again
                seek += (tokens[last_slice - 1] - TIMESTAMP_BEGIN) * input_stride
        else:
            duration = segment_duration
            stamps = [t for t in tokens if t >= TIMESTAMP_BEGIN]
            if len(stamps) > 0 and stamps[-1] != TIMESTAMP_BEGIN:
                duration = (stamps[-1] - TIMESTAMP_BEGIN) * time_precision  # a lone timestamp bounds the segment (:388-399)
            current_segments.append(new_segment(start=time_offset, end=time_offset + duration, toks=tokens))
            seek += segment_size

        if word_timestamps:  # (:409-486)
            add_word_timestamps(segments=current_segments, model=model, tokenizer=tokenizer, mel=window(previous_seek, seek_clip_end)[0],
                                num_frames=segment_size, prepend_punctuations=prepend_punctuations, append_punctuations=append_punctuations,
                                last_speech_timestamp=last_speech_timestamp)
            if not single_timestamp_ending:  # resume right after the last word instead of at the last closed timestamp
                last_word_end = get_end(current_segments)
                if last_word_end is not None and last_word_end > time_offset:
                    seek = round(last_word_end * FRAMES_PER_SECOND)
            if hallucination_silence_threshold is not None:  # skip silence before possible hallucinations
                threshold = hallucination_silence_threshold
                if not single_timestamp_ending:
                    last_word_end = get_end(current_segments)
                    if last_word_end is not None and last_word_end > time_offset:
                        remaining_duration = window_end_time - last_word_end
                        seek = round(last_word_end * FRAMES_PER_SECOND) if remaining_duration > threshold else previous_seek + segment_size
                # if the first segment might be a hallucination, skip the leading silence
                first_segment = next_words_segment(current_segments)
                if first_segment is not None and is_segment_anomaly(first_segment):
                    gap = first_segment["start"] - time_offset
                    if gap > threshold:
                        seek = previous_seek + round(gap * FRAMES_PER_SECOND)
                        continue
                # skip silence before any possible hallucination that is surrounded by silence or more hallucinations
                hal_last_end = last_speech_timestamp
                for si in range(len(current_segments)):
                    segment = current_segments[si]
                    if not segment["words"]:
                        continue
                    if is_segment_anomaly(segment):
                        next_segment = next_words_segment(current_segments[si + 1:])
                        hal_next_start = next_segment["words"][0]["start"] if next_segment is not None else time_offset + segment_duration
                        silence_before = (segment["start"] - hal_last_end > threshold or segment["start"] < threshold
                                          or segment["start"] - time_offset < 2.0)
                        silence_after = (hal_next_start - segment["end"] > threshold or is_segment_anomaly(next_segment)
                                         or window_end_time - segment["end"] < 2.0)
                        if silence_before and silence_after:
                            seek = round(max(time_offset + 1, segment["start"]) * FRAMES_PER_SECOND)
                            if content_duration - segment["end"] < threshold:
                                seek = content_frames
                            current_segments[si:] = []
                            break
                    hal_last_end = segment["end"]
            last_word_end = get_end(current_segments)
            if last_word_end is not None:
                last_speech_timestamp = last_word_end

        if verbose and tokenizer is not None:  # (:488-492)
            for seg in current_segments:
                print(f"[{format_timestamp(seg['start'])} --> {format_timestamp(seg['end'])}] {seg['text']}")
        # instantaneous segments and segments without text keep their slot but lose their tokens (:494-499).  "Without text": the
        # decoded string is blank; at token level (no tokenizer) no token below eot
        for seg in current_segments:
            empty = seg["text"].strip() == "" if tokenizer is not None else not any(t < EOT for t in seg["tokens"])
            if seg["start"] == seg["end"] or empty:
                seg["tokens"] = []
                seg["words"] = []
                if tokenizer is not None:
                    seg["text"] = ""
        all_segments.extend({"id": i, **seg} for i, seg in enumerate(current_segments, start=len(all_segments)))
        all_tokens.extend(t for seg in current_segments for t in seg["tokens"])

    out_tokens = all_tokens[len(initial_prompt_tokens):]
    return {"text": tokenizer.decode(out_tokens) if tokenizer is not None else None, "tokens": out_tokens, "segments": all_segments,
            "language": decode_options["language"]}
