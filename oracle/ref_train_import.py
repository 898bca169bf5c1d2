"""Container-only loader for the UNMODIFIED ``/root/reference/scripts/training/train_timestamps.py``.

TEST INFRASTRUCTURE. Never imported by the product path (olmoasr_amd/*, scripts/training/*).

The reference's training script imports, at module load, packages this image does not have (``wandb``, ``jiwer``, ``fire``,
``zstandard``, ``whisper``, ``webvtt`` through ``olmoasr/utils.py``) and two sibling modules that pull in datasets / librosa /
torchaudio (``scripts.eval.eval``) -- none of which the INTEGER data-layout code of SURVEY 8 row a18 touches.  We register empty
stand-ins for exactly those names, then import the reference's own files:

  * ``olmoasr/utils.py`` (unmodified) -> ``convert_to_milliseconds`` and ``TranscriptReader`` are the reference's code; the only
    stand-in underneath is ``webvtt.from_string`` (third-party ``webvtt-py``, requirements.txt), restated here as the cue-block parser
    its documentation describes: blocks separated by blank lines, a ``start --> end`` timing line, the text lines joined by ``\\n``.
  * ``scripts/training/train_timestamps.py`` (unmodified) -> ``AudioTextDataset.preprocess_text`` / ``_process_empty_transcript`` /
    ``_process_non_empty_transcript`` / ``_build_timestamp_sequence`` / ``_convert_to_token_idx`` (:218-506), ``prepare_sched`` (:739-783).

Used by ``oracle/gen_token_layout_golden.py`` (writes tests/golden/token_layout_ref.json) and by the live half of
``tests/test_token_layout_ref_cpu.py`` (skipped where /root/reference is not mounted: it never is on the GPU box).
"""
import importlib
import importlib.util
import os
import re
import sys
import types

from oracle import ref_import

REF_ROOT = ref_import.REF_ROOT


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "scripts", "training", "train_timestamps.py"))


class _Caption:
    def __init__(self, start, end, text):
        self.start, self.end, self.text = start, end, text


_TIMING = re.compile(r"^\s*(\d{2,}:\d{2}:\d{2}\.\d{3}|\d{2}:\d{2}\.\d{3})\s+-->\s+(\d{2,}:\d{2}:\d{2}\.\d{3}|\d{2}:\d{2}\.\d{3})")


def webvtt_from_string(s: str):
    """Stand-in for ``webvtt.from_string`` (webvtt-py): the list of cues of a WebVTT document, each with ``.start`` / ``.end``
    (``HH:MM:SS.mmm`` strings, a short ``MM:SS.mmm`` form is widened) and ``.text`` (payload lines joined by a newline)."""
    def full(t):
        return t if t.count(":") == 2 else "00:" + t
    caps = []
    for block in re.split(r"\n\s*\n", s.replace("\r\n", "\n").strip()):
        lines = [ln for ln in block.split("\n") if ln.strip() != ""]
        for i, ln in enumerate(lines):
            m = _TIMING.match(ln)
            if m:
                caps.append(_Caption(full(m.group(1)), full(m.group(2)), "\n".join(lines[i + 1:])))
                break
    return caps


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Returns the reference's ``train_timestamps`` module (its ``olmoasr.utils`` is the reference's own file)."""
    if not available():
        raise RuntimeError("reference tree not mounted at %s" % REF_ROOT)
    if "oasr_ref_train_timestamps" in sys.modules:
        return sys.modules["oasr_ref_train_timestamps"]
    ref_import.load()  # whisper.* stand-ins + the reference's olmoasr package (without its __init__)

    class _Any:
        def __init__(self, *a, **k):
            pass

    def _nyi(*a, **k):
        raise NotImplementedError("stand-in only")

    w = sys.modules["whisper"]
    w.DecodingOptions = _Any
    w.normalizers = _stub("whisper.normalizers", EnglishTextNormalizer=_Any)
    _stub("wandb", init=_nyi, log=_nyi, Table=_Any, Artifact=_Any)
    _stub("jiwer", wer=_nyi)
    _stub("fire", Fire=_nyi)
    _stub("zstandard", ZstdDecompressor=_Any)
    _stub("webvtt", from_string=webvtt_from_string, read=_nyi)
    _stub("for_logging", TRAIN_TABLE_COLS=[], EVAL_TABLE_COLS=[])
    if "scripts" not in sys.modules:
        _stub("scripts").__path__ = []
    if "scripts.eval" not in sys.modules:
        _stub("scripts.eval").__path__ = []
    _stub("scripts.eval.eval", EvalDataset=_Any)
    # the reference's own olmoasr/utils.py (needs webvtt / jiwer / whisper.tokenizer.get_tokenizer by name only)
    utils = importlib.import_module("olmoasr.utils")
    sys.modules["olmoasr"].utils = utils
    spec = importlib.util.spec_from_file_location("oasr_ref_train_timestamps", os.path.join(REF_ROOT, "scripts", "training", "train_timestamps.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["oasr_ref_train_timestamps"] = mod
    spec.loader.exec_module(mod)
    return mod
