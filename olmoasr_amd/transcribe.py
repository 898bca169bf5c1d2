"""Long-form transcription driver on the native engine (greedy subset).

Mirrors the control flow of the reference's ``olmoasr/transcribe.py::transcribe`` (:47-523) for the configuration
BASELINE.json names (config 5: greedy, temperature 0): the whole waveform is converted to log-mel once with
``padding=N_SAMPLES`` (:148), then 30 s windows ``mel[:, seek:seek+3000]`` are padded/trimmed to 3000 frames (:293-295)
and decoded WITHOUT conditioning on previous text (the reference has prompt conditioning commented out, :297-302).
Without timestamp tokens the seek advances by a full window (:404-408, the no-timestamp branch).  ``temperature`` may be a
tuple: ``decode_with_fallback`` (:193-233) retries a window at the next temperature while its ``avg_logprob`` is below
``logprob_threshold`` (beam options apply at temperature 0 only, ``best_of`` above it), and keeps a silent window
(``no_speech_prob > no_speech_threshold`` with a low ``avg_logprob``) out of the token stream (:305-320).  The
compression-ratio test and word timestamps need text, i.e. the tokenizer of the un-vendored openai-whisper, and are out of
scope; results carry token ids, not text.
"""
from typing import Optional

import numpy as np
import torch

from .audio import HOP_LENGTH, N_FRAMES, N_SAMPLES, SAMPLE_RATE, log_mel_spectrogram, pad_or_trim
from .decoding import DecodingOptions, decode


@torch.no_grad()
def transcribe(model, audio, *, verbose: Optional[bool] = None, temperature=0.0, batch_windows: int = 16,
               no_speech_threshold: Optional[float] = None, logprob_threshold: Optional[float] = -1.0, **decode_options):
    temperatures = tuple(temperature) if isinstance(temperature, (tuple, list)) else (float(temperature),)
    if isinstance(audio, str):
        raise NotImplementedError("audio file decoding (ffmpeg) is out of scope: pass a waveform array/tensor")
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.ascontiguousarray(audio))
    mel = log_mel_spectrogram(audio, model.dims.n_mels, padding=N_SAMPLES, device=model.device)  # [80, n_frames + 3000]
    content_frames = mel.shape[-1] - N_FRAMES
    options = DecodingOptions(**{"without_timestamps": True, **decode_options})
    seeks = list(range(0, content_frames, N_FRAMES))
    all_tokens, segments = [], []
    for i in range(0, len(seeks), batch_windows):
        chunk = seeks[i:i + batch_windows]
        windows = torch.stack([pad_or_trim(mel[:, s:s + N_FRAMES], N_FRAMES) for s in chunk])  # windows are independent here
        results = [None] * len(chunk)
        todo = list(range(len(chunk)))
        for t in temperatures:  # decode_with_fallback, per window, batched over the windows still failing
            kw = dict(options.__dict__, temperature=t)
            if t > 0:
                kw.update(beam_size=None, patience=None)   # disable beam_size and patience when t > 0
            else:
                kw.update(best_of=None)                    # disable best_of when t == 0
            out = decode(model, windows[todo], DecodingOptions(**kw))
            still = []
            for j, r in zip(todo, out):
                results[j] = r
                needs_fallback = logprob_threshold is not None and r.avg_logprob < logprob_threshold
                if no_speech_threshold is not None and r.no_speech_prob == r.no_speech_prob and r.no_speech_prob > no_speech_threshold:
                    needs_fallback = False                 # silence
                if needs_fallback:
                    still.append(j)
            todo = still
            if not todo:
                break
        for s, r in zip(chunk, results):
            t0 = s * HOP_LENGTH / SAMPLE_RATE
            t1 = min(s + N_FRAMES, content_frames) * HOP_LENGTH / SAMPLE_RATE
            silent = (no_speech_threshold is not None and r.no_speech_prob == r.no_speech_prob and r.no_speech_prob > no_speech_threshold
                      and not (logprob_threshold is not None and r.avg_logprob > logprob_threshold))
            toks = [] if silent else r.tokens
            segments.append({"id": len(segments), "seek": s, "start": t0, "end": t1, "tokens": toks, "temperature": r.temperature,
                             "avg_logprob": r.avg_logprob, "no_speech_prob": r.no_speech_prob})
            all_tokens.extend(toks)
    return {"tokens": all_tokens, "segments": segments, "language": "en", "text": None}
