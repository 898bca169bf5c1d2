"""olmoasr_amd -- MI355X-native (gfx950 HIP) implementation of the OLMoASR training hot path.

Mirrors the reference's Python surface for this path (olmoasr/__init__.py:17-21, olmoasr/model.py); all
arithmetic runs in liboasr.so (olmoasr_amd/csrc, C ABI in include/oasr.h).  There is no CPU fallback."""
__version__ = "0.1.0"

from .config.model_dims import VARIANT_TO_DIMS, ModelDimensions  # noqa: E402,F401


def __getattr__(name):
    """Lazy re-exports mirroring ``olmoasr/__init__.py:17-21`` (torch/HIP are only touched when used)."""
    if name in ("log_mel_spectrogram", "pad_or_trim", "load_audio", "N_SAMPLES", "N_FRAMES", "SAMPLE_RATE", "HOP_LENGTH", "N_FFT"):
        from . import audio
        return getattr(audio, name)
    if name == "OLMoASR":
        from .model import OLMoASR
        return OLMoASR
    if name in ("load_model", "gen_inf_ckpt", "MODEL2LINK"):
        from . import hub
        return getattr(hub, name)
    if name in ("model", "inf_model", "audio", "ddp", "ops", "synth", "decoding", "transcribe", "hub"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
