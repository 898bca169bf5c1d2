"""Host-side mirror of ``whisper.audio`` as re-exported by the reference (olmoasr/__init__.py:21) and used at
scripts/training/train_timestamps.py:207-214 and olmoasr/transcribe.py:11-19,148: same names, arguments and constants.
``log_mel_spectrogram`` runs the HIP kernel (csrc/logmel.hip); there is no CPU implementation in the product."""
from typing import Optional, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import _native as N
from . import ops

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000 samples in a 30-second chunk
N_FRAMES = N_SAMPLES // HOP_LENGTH  # 3000 frames in a mel spectrogram input
N_SAMPLES_PER_TOKEN = HOP_LENGTH * 2
FRAMES_PER_SECOND = SAMPLE_RATE // HOP_LENGTH
TOKENS_PER_SECOND = SAMPLE_RATE // N_SAMPLES_PER_TOKEN


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    """Pad (zeros) or trim the audio array to ``length`` along ``axis`` (numpy arrays and tensors)."""
    if torch.is_tensor(array):
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            pad_widths = [(0, 0)] * array.ndim
            pad_widths[axis] = (0, length - array.shape[axis])
            array = F.pad(array, [p for sizes in pad_widths[::-1] for p in sizes])
        return array
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        pad_widths = [(0, 0)] * array.ndim
        pad_widths[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad_widths)
    return array


def mel_filters(device=None, n_mels: int = 80) -> torch.Tensor:
    """The slaney 80x201 filterbank (what whisper loads from assets/mel_filters.npz), computed by the library."""
    if n_mels != 80:
        raise N.NativeError("only n_mels=80 is supported (every OLMoASR variant, olmoasr/config/model_dims.py:28-89)")
    out = np.empty((80, 201), dtype=np.float32)
    N.check(N.lib().oasr_mel_filterbank(out.ctypes.data), "oasr_mel_filterbank")
    t = torch.from_numpy(out)
    return t.to(device) if device is not None else t


def log_mel_spectrogram(audio: Union[np.ndarray, torch.Tensor], n_mels: int = 80, padding: int = 0,
                        device: Optional[Union[str, torch.device]] = None) -> torch.Tensor:
    """float32 waveform in [-1,1] or int16 PCM, shape [n] or [B, n] -> log-mel [..., 80, n // 160] on the HIP device.
    (``str`` paths / ffmpeg decoding of the original are out of scope: SURVEY.md section 2, load_audio.)"""
    if isinstance(audio, str):
        raise N.NativeError("log_mel_spectrogram(path): audio file decoding (ffmpeg) is out of scope; pass samples")
    if n_mels != 80:
        raise N.NativeError("only n_mels=80 is supported")
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.ascontiguousarray(audio))
    if device is None:
        device = audio.device if audio.is_cuda else "cuda"
    audio = audio.to(device)
    if audio.dtype not in (torch.int16, torch.float32):
        audio = audio.float()
    if padding > 0:
        audio = F.pad(audio, (0, padding))
    squeeze = audio.dim() == 1
    if squeeze:
        audio = audio[None]
    lead = audio.shape[:-1]
    mel = ops.log_mel(audio.reshape(-1, audio.shape[-1]).contiguous())
    mel = mel.reshape(*lead, 80, mel.shape[-1])
    return mel[0] if squeeze else mel
