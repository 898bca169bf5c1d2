"""Long-form transcription driver on the native engine (greedy subset).

Mirrors the control flow of the reference's ``olmoasr/transcribe.py::transcribe`` (:47-523) for the configuration
BASELINE.json names (config 5: greedy, temperature 0): the whole waveform is converted to log-mel once with
``padding=N_SAMPLES`` (:148), then 30 s windows ``mel[:, seek:seek+3000]`` are padded/trimmed to 3000 frames (:293-295)
and decoded WITHOUT conditioning on previous text (the reference has prompt conditioning commented out, :297-302).
Without timestamp tokens the seek advances by a full window (:404-408, the no-timestamp branch).  Temperature fallback
(:193-233), beam search and word timestamps need the tokenizer/normalizer of the un-vendored openai-whisper and are out of
scope; results carry token ids, not text.
"""
from typing import Optional

import numpy as np
import torch

from .audio import HOP_LENGTH, N_FRAMES, N_SAMPLES, SAMPLE_RATE, log_mel_spectrogram, pad_or_trim
from .decoding import DecodingOptions, decode


@torch.no_grad()
def transcribe(model, audio, *, verbose: Optional[bool] = None, temperature: float = 0.0, batch_windows: int = 8,
               no_speech_threshold: Optional[float] = None, logprob_threshold: Optional[float] = -1.0, **decode_options):
    if temperature not in (0, 0.0, (0.0,), (0,)):
        raise NotImplementedError("only temperature 0 (greedy) is implemented on the native path")
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
        for s, r in zip(chunk, decode(model, windows, options)):
            t0 = s * HOP_LENGTH / SAMPLE_RATE
            t1 = min(s + N_FRAMES, content_frames) * HOP_LENGTH / SAMPLE_RATE
            segments.append({"id": len(segments), "seek": s, "start": t0, "end": t1, "tokens": r.tokens, "temperature": 0.0,
                             "avg_logprob": r.avg_logprob})
            all_tokens.extend(r.tokens)
    return {"tokens": all_tokens, "segments": segments, "language": "en", "text": None}
