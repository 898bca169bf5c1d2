"""Transcript -> training token sequence: the INTEGER side of the reference's data path.

What ``AudioTextDataset.preprocess_text`` (scripts/training/train_timestamps.py:238-343) and its helpers ``_process_empty_transcript``
(:345-393), ``_process_non_empty_transcript`` (:395-460), ``_build_timestamp_sequence`` (:462-506) and ``_convert_to_token_idx``
(:218-236) do, as one table-driven function over a parsed transcript.  A sample has four possible layouts:

    no-timestamps   <sot> <notimestamps> text... <eot>
    timestamps      <sot> <|s0|> text0 <|e0|> <|s1|> text1 <|e1|> ... <|norm_end|> <eot>        (<|t|> = timestamp_begin + ms // 20)
    silence         <sot> <notimestamps> <nospeech> <eot>                                       (empty transcript, norm_end >= 30 s)
    empty + stamps  <sot> <|0|> <|norm_end|> <|norm_end|> <eot>                                 (empty transcript, norm_end < 30 s)

and which one is chosen depends on (transcript empty?, norm_end vs 30 s, ``only_no_ts_mode``, ``ts_mode``, a coin).  The coin is the
reference's ``np.random.rand() >= 0.5`` drawn from the SAME global numpy stream in the SAME order (one draw per decision, and the
reference's second, independent draw for an empty transcript's ``timestamp_mode`` flag), so a seeded worker produces the reference's
choices bit for bit.  ``tests/test_token_layout_ref_cpu.py`` pins this module, the oracle's restatement (``oracle/model_oracle.py::preprocess_text``) and the
synthetic generator's layout (``synth._layout``) against the reference's own file RUNNING (tests/golden/token_layout_ref.json).

The tokenizer is a plug with whisper's attribute names (``encode``, ``sot_sequence``, ``sot_sequence_including_notimestamps``,
``timestamp_begin``, ``no_speech``, ``eot``): the tiktoken vocabulary is not available offline.  The transcript reader handles the
WebVTT cue blocks the reference's shards carry in ``seg_content`` (``olmoasr.utils.TranscriptReader`` on ``webvtt``, :173-262).
"""
import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .synth import N_TEXT_CTX, _layout

THIRTY_S_MS = 30000
MS_PER_TIMESTAMP_TOKEN = 20

_CUE = re.compile(r"^\s*((?:\d{2,}:)?\d{2}:\d{2}\.\d{3})\s+-->\s+((?:\d{2,}:)?\d{2}:\d{2}\.\d{3})")


def to_ms(timestamp) -> int:
    """``HH:MM:SS.mmm`` (or an int already in ms) -> ms; olmoasr/utils.py:31-47 incl. its ValueError."""
    if not isinstance(timestamp, str):
        return int(timestamp)
    try:
        h, m, s, ms = map(float, timestamp.replace(".", ":").split(":"))
        return int(h * 3600000 + m * 60000 + s * 1000 + ms)
    except (ValueError, IndexError) as e:
        raise ValueError(f"Invalid timestamp format: {timestamp}") from e


def read_transcript(transcript_string: str, ext: str = "vtt") -> List[Tuple[str, str, str]]:
    """Cues of a transcript string as [(start, end, text)] in file order -- the items of the dict ``TranscriptReader.read()`` returns
    (olmoasr/utils.py:214-262; like a dict, a later cue with the same (start, end) replaces the text of the earlier one in place).
    Only WebVTT: the reference's own SRT branch cannot return (``_read_transcript_file`` fills the dict for "vtt" only and then reads
    start / end variables it never bound, :236-254), so its shards are WebVTT."""
    if ext != "vtt":
        raise ValueError(f"Unsupported file type: {ext}")
    cues: Dict[Tuple[str, str], str] = {}
    for block in re.split(r"\n\s*\n", transcript_string.replace("\r\n", "\n").strip()):
        lines = [ln for ln in block.split("\n") if ln.strip()]
        for i, ln in enumerate(lines):
            m = _CUE.match(ln)
            if m:
                a, b = (t if t.count(":") == 2 else "00:" + t for t in m.groups())
                cues[(a, b)] = "\n".join(lines[i + 1:])
                break
    return [(a, b, t) for (a, b), t in cues.items()]


def timestamp_token(timestamp, timestamp_begin: int) -> Optional[int]:
    """train_timestamps.py:218-236: None past 30 s."""
    ms = to_ms(timestamp)
    return None if ms > THIRTY_S_MS else timestamp_begin + ms // MS_PER_TIMESTAMP_TOKEN


def build_tokens(cues: Sequence[Tuple[str, str, str]], tokenizer, norm_end, ts_mode, only_no_ts_mode,
                 rand: Optional[Callable[[], float]] = None) -> Tuple[List[int], bool, object]:
    """(tokens [sot .. eot], timestamp_mode, norm_end as the reference hands it back) for one sample."""
    rand = rand or np.random.rand
    tb = tokenizer.timestamp_begin
    sot_nots = list(tokenizer.sot_sequence_including_notimestamps)
    if isinstance(norm_end, str):
        norm_end = to_ms(norm_end)
    end_tok = tb + min(norm_end, THIRTY_S_MS) // MS_PER_TIMESTAMP_TOKEN  # <|norm_end|>, clamped to <|30.00|> (:354-357, :499-502)

    if not cues:  # :283-291 + :345-393
        if norm_end >= THIRTY_S_MS:
            return sot_nots + [tokenizer.no_speech, tokenizer.eot], False, norm_end
        nothing = list(tokenizer.encode(""))
        if only_no_ts_mode is True:
            tokens = sot_nots + nothing + [tokenizer.eot]
        elif rand() >= 0.5:
            tokens = [tokenizer.sot_sequence[0], tb] + nothing + [end_tok, end_tok, tokenizer.eot]
        else:
            tokens = sot_nots + nothing + [tokenizer.eot]
        # (the flag is a second, independent draw in the reference: it can disagree with the layout chosen above)
        flag = only_no_ts_mode is False and rand() >= 0.5
        return tokens, bool(flag), norm_end

    cues = list(cues)
    if norm_end > THIRTY_S_MS:  # :406-412: drop the last cue, end at the previous one's end STRING, never timestamps
        if len(cues) > 1:
            cues.pop()
            norm_end = cues[-1][1]
        only_no_ts_mode = True
    text = [list(tokenizer.encode(" " + t.strip())) for _, _, t in cues]
    plain = sot_nots + [x for seg in text for x in seg] + [tokenizer.eot]
    if only_no_ts_mode is True or not (rand() >= 0.5) or ts_mode is not True:
        return plain, False, norm_end
    # :462-506 (norm_end is an int <= 30 s here)
    out = [tokenizer.sot_sequence[0]]
    for (a, b, _), seg in zip(cues, text):
        ta, tb_ = timestamp_token(a, tb), timestamp_token(b, tb)
        if ta is None or tb_ is None:
            return plain, False, norm_end  # a boundary past 30 s: fall back (:437-452)
        out += [ta] + seg + [tb_]
    return out + [end_tok, tokenizer.eot], True, norm_end


def preprocess_text(transcript_string: str, transcript_file: str, tokenizer, norm_end, ts_mode, only_no_ts_mode,
                    n_text_ctx: int = N_TEXT_CTX, rand: Optional[Callable[[], float]] = None):
    """``AudioTextDataset.preprocess_text`` without the [448, 448] float mask: returns (text_input i64 [448], text_y i64 [448],
    text_len, timestamp_mode, norm_end).  ``text_len`` = len(tokens) - 1 = the first -inf column of the reference's mask (:314-315)."""
    assert n_text_ctx == N_TEXT_CTX
    import torch
    cues = read_transcript(transcript_string, transcript_file.split(".")[-1])
    tokens, timestamp_mode, norm_end = build_tokens(cues, tokenizer, norm_end, ts_mode, only_no_ts_mode, rand)
    if len(tokens) - 1 > n_text_ctx:  # (the reference prints a warning, then np.pad raises on the negative width, :317-329)
        raise ValueError(f"{transcript_file}: {len(tokens) - 1} text tokens exceed n_text_ctx = {n_text_ctx}")
    text_input, text_y, text_len = _layout(torch.tensor(tokens, dtype=torch.long))
    return text_input, text_y, text_len, timestamp_mode, norm_end


def reference_text_fn(tokenizer, rand: Optional[Callable[[], float]] = None):
    """``text_fn`` for ``olmoasr_amd.data.AudioTextShards``: tokenises ``seg_content`` like the reference's dataset does."""
    def text_fn(sample: Dict):
        cues = read_transcript(sample["seg_content"], sample["subtitle_file"].split(".")[-1])
        return build_tokens(cues, tokenizer, sample["norm_end"], sample["ts_mode"], sample["only_no_ts_mode"], rand)
    return text_fn
