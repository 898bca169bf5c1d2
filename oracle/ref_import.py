"""Container-only loader for the UNMODIFIED reference model files.

TEST INFRASTRUCTURE. Never imported by the product path (olmoasr_amd/*).

The reference's ``olmoasr/model.py`` imports names from the third-party
``openai-whisper`` package at module load (model.py:9-12) and
``olmoasr/transcribe.py`` does the same (transcribe.py:11-33).  That package
is absent in this image, so we register empty stand-ins for exactly those
names, then expose ``/root/reference/olmoasr`` as a namespace-like package
named ``olmoasr_ref`` WITHOUT executing the reference's ``__init__`` (which
needs webvtt/jiwer).  Only used by ``oracle/gen_golden.py`` and by tests that
are skipped when /root/reference is not mounted (it never is on the GPU box).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("OLMOASR_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "olmoasr", "model.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Returns (model_module, inf_model_module, model_dims_module) of the reference."""
    if not available():
        raise RuntimeError("reference tree not mounted at %s" % REF_ROOT)
    if "olmoasr.model" in sys.modules and getattr(sys.modules["olmoasr"], "_is_ref_stub", False):
        return (sys.modules["olmoasr.model"], sys.modules["olmoasr.inf_model"],
                sys.modules["olmoasr.config.model_dims"])

    class _Any:  # placeholder for dataclass-typed names used only in annotations
        def __init__(self, *a, **k):
            pass

    def _nyi(*a, **k):
        raise NotImplementedError("openai-whisper is not installed; stub only")

    w = _stub("whisper")
    w.decoding = _stub("whisper.decoding", decode=_nyi, detect_language=_nyi,
                       DecodingOptions=_Any, DecodingResult=_Any)
    w.audio = _stub("whisper.audio", FRAMES_PER_SECOND=100, HOP_LENGTH=160, N_FRAMES=3000,
                    N_SAMPLES=480000, SAMPLE_RATE=16000, log_mel_spectrogram=_nyi,
                    pad_or_trim=_nyi)
    w.timing = _stub("whisper.timing", add_word_timestamps=_nyi)
    w.tokenizer = _stub("whisper.tokenizer", LANGUAGES={}, TO_LANGUAGE_CODE={},
                        get_tokenizer=_nyi, Tokenizer=_Any)
    w.utils = _stub("whisper.utils", exact_div=lambda a, b: a // b, format_timestamp=_nyi,
                    get_end=_nyi, get_writer=_nyi, make_safe=_nyi, optional_float=_nyi,
                    optional_int=_nyi, str2bool=_nyi)
    pkg = types.ModuleType("olmoasr")
    pkg.__path__ = [os.path.join(REF_ROOT, "olmoasr")]
    pkg._is_ref_stub = True
    sys.modules["olmoasr"] = pkg
    model = importlib.import_module("olmoasr.model")
    inf_model = importlib.import_module("olmoasr.inf_model")
    dims = importlib.import_module("olmoasr.config.model_dims")
    return model, inf_model, dims


def load_transcribe():
    """The UNMODIFIED ``/root/reference/olmoasr/transcribe.py`` as a module (its ``whisper.*`` imports are the stand-ins
    registered by ``load()``).  The caller patches the module globals it needs (``log_mel_spectrogram``, ``pad_or_trim``,
    ``get_tokenizer``, ``DecodingOptions``) -- see tests/test_oracle_transcribe_ref_cpu.py, the pin of
    ``oracle.decode_oracle.transcribe`` against the reference's own seek loop (transcribe.py:147-517)."""
    load()
    return importlib.import_module("olmoasr.transcribe")
