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


def load_audio(file: str, sr: int = SAMPLE_RATE) -> np.ndarray:
    """``whisper.audio.load_audio`` (re-exported olmoasr/__init__.py:21): a mono float32 waveform in [-1, 1] at ``sr`` Hz.
    The original shells out to the ffmpeg CLI; so does this when ffmpeg is on PATH (same decode: ``-ac 1 -ar sr -f s16le``).  Without
    ffmpeg, the formats that need no codec are read directly: RIFF/WAVE PCM (8/16/24/32-bit integer or 32/64-bit float) and ``.npy``
    int16 clips (the reference's training format, train_timestamps.py:196); channels are averaged, other sample rates are resampled
    with a polyphase filter (scipy.signal.resample_poly)."""
    import shutil
    if file.endswith(".npy"):
        arr = np.load(file)
        return (arr.astype(np.float32) / 32768.0) if arr.dtype == np.int16 else arr.astype(np.float32)
    if shutil.which("ffmpeg"):
        import subprocess
        cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", file, "-f", "s16le", "-ac", "1", "-acodec", "pcm_s16le", "-ar", str(sr), "-"]
        try:
            out = subprocess.run(cmd, capture_output=True, check=True).stdout
        except subprocess.CalledProcessError as e:
            raise RuntimeError(f"Failed to load audio: {e.stderr.decode()}") from e
        return np.frombuffer(out, np.int16).flatten().astype(np.float32) / 32768.0
    from scipy.io import wavfile
    try:
        rate, data = wavfile.read(file)
    except Exception as e:
        raise RuntimeError(f"Failed to load audio: {file!r} is not a PCM WAVE file and ffmpeg is not installed ({e})") from e
    if data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        x = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    else:
        x = data.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if rate != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(rate), int(sr))
        x = resample_poly(x, sr // g, rate // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


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


def log_mel_spectrogram(audio: Union[str, np.ndarray, torch.Tensor], n_mels: int = 80, padding: int = 0,
                        device: Optional[Union[str, torch.device]] = None) -> torch.Tensor:
    """float32 waveform in [-1,1] or int16 PCM, shape [n] or [B, n] (or a file path, see ``load_audio``) -> log-mel
    [..., 80, n // 160] on the HIP device."""
    if isinstance(audio, str):
        audio = load_audio(audio)
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
