"""Writes tests/golden/token_layout_ref.json: the INTEGER token / mask layout of the reference's training samples (SURVEY.md section 8
row a18), produced by the reference's own ``scripts/training/train_timestamps.py`` RUNNING here (oracle/ref_train_import.py imports the
unmodified file): ``AudioTextDataset.preprocess_text`` and, through it, ``_process_empty_transcript`` / ``_process_non_empty_transcript``
/ ``_build_timestamp_sequence`` / ``_convert_to_token_idx`` (:218-506), and ``prepare_sched`` (:739-783).

TEST INFRASTRUCTURE (container only: /root/reference does not exist on the GPU box; the fixture travels instead).

    python -m oracle.gen_token_layout_golden

The tokenizer is scripted (whisper's is not installed): ``ScriptedTokenizer.encode`` maps each whitespace-separated word to
crc32(word) mod 50256; the special ids are the English-only vocabulary's.  The reference's coin (``np.random.rand() >= 0.5``) is made
reproducible by ``np.random.seed(case["seed"])`` right before the call; the cases cover both coin outcomes of every decision.
"""
import contextlib
import io
import json
import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "token_layout_ref.json")


class ScriptedTokenizer:
    """The attributes of ``whisper.tokenizer.Tokenizer`` the data path reads, English-only ids."""
    eot = 50256
    sot_sequence = (50257,)
    sot_sequence_including_notimestamps = (50257, 50362)
    no_speech = 50361
    timestamp_begin = 50363

    def encode(self, text: str):
        return [zlib.crc32(w.encode("utf-8")) % 50256 for w in text.split()]


def stamp(ms: int) -> str:
    return f"{ms // 3600000:02d}:{ms // 60000 % 60:02d}:{ms // 1000 % 60:02d}.{ms % 1000:03d}"


def vtt(cues) -> str:
    """cues: [(start_ms, end_ms, text)] -> a WebVTT document as the reference's shards carry it in ``seg_content``."""
    return "WEBVTT\n\n" + "".join(f"{stamp(a)} --> {stamp(b)}\n{t}\n\n" for a, b, t in cues)


def words(n: int, salt: int):
    return " ".join(f"w{salt}x{i}" for i in range(n))


def cases():
    out = []

    def add(name, cues, norm_end, ts_mode, only_no_ts_mode, seeds=(0, 1, 2, 3)):
        for seed in seeds:
            out.append({"name": f"{name}/seed{seed}", "seg_content": vtt(cues) if cues is not None else "WEBVTT\n\n", "subtitle_file": "x/y.vtt",
                        "norm_end": norm_end, "ts_mode": ts_mode, "only_no_ts_mode": only_no_ts_mode, "seed": seed})
    one = [(0, 4200, "hello there  world")]
    three = [(0, 2500, words(5, 1)), (2500, 9980, " " + words(9, 2) + " "), (12000, 21340, words(3, 3) + "\n" + words(2, 4))]
    # non-empty transcripts, the coin x ts_mode x only_no_ts_mode table (:414-458)
    add("one_cue_ts", one, 4200, True, False)
    add("one_cue_no_ts_flag", one, 4200, False, False, seeds=(0, 1))
    add("one_cue_only_no_ts", one, 4200, True, True, seeds=(0, 1))
    add("three_cues_ts", three, 21340, True, False)
    add("three_cues_norm_end_string", three, stamp(21345), True, False, seeds=(0, 1))
    add("three_cues_norm_end_30s", three, 30000, True, False, seeds=(0, 1))
    # a boundary past 30 s inside a <= 30 s sample: _build_timestamp_sequence returns None -> no-timestamp fallback (:437-452, :481-483)
    add("boundary_past_30s", [(0, 1000, "a b"), (1000, 30020, "c d e")], 29000, True, False)
    add("boundary_exactly_30s", [(0, 1000, "a b"), (1000, 30000, "c d e")], 30000, True, False, seeds=(0, 1))
    # > 30 s segments: the last cue is dropped, norm_end becomes the previous cue's END STRING, timestamps are off (:406-412)
    add("over_30s_three_cues", three + [(21340, 33000, "too long tail")], 33000, True, False, seeds=(0, 1))
    add("over_30s_single_cue", [(0, 31000, words(6, 5))], 31000, True, False, seeds=(0, 1))
    add("over_30s_norm_end_string", three + [(21340, 41000, "tail")], stamp(41000), True, False, seeds=(0,))
    # empty transcripts (:283-291, :345-393): two independent coins
    add("empty_short", None, 12340, True, False, seeds=(0, 1, 2, 3, 4, 5, 6, 7))
    add("empty_short_only_no_ts", None, 12340, True, True, seeds=(0, 1))
    add("empty_exactly_30s", None, 30000, True, False, seeds=(0, 1))
    add("empty_over_30s", None, 45000, False, False, seeds=(0,))
    add("empty_zero", None, 0, True, False, seeds=(0, 1, 2, 3))
    # length edge: exactly n_text_ctx input tokens (447 words + sot + notimestamps + eot = 450 tokens -> 449 > 448 would overflow; 446 fits exactly)
    add("fills_context", [(0, 29000, words(446, 6))], 29000, False, True, seeds=(0,))
    add("fills_context_ts", [(0, 29000, words(444, 7))], 29000, True, False, seeds=(0, 1, 2, 3))
    # duplicate cue keys: the reference's dict keeps the first position with the last text
    add("duplicate_cue", [(0, 1000, "first"), (1000, 2000, "mid"), (0, 1000, "again")], 2000, True, False, seeds=(0, 1, 2, 3))
    # overflow: the reference raises (np.pad with a negative width)
    add("overflows_context", [(0, 29000, words(460, 8))], 29000, False, True, seeds=(0,))
    return out


def run_reference(case, mod, tok):
    ds = mod.AudioTextDataset([], 448, 6)
    np.random.seed(case["seed"])
    try:
        with contextlib.redirect_stdout(io.StringIO()):  # (the reference prints the whole sample when it is too long)
            ti, ty, mask, ts_mode, norm_end, _ = ds.preprocess_text(case["seg_content"], case["subtitle_file"], tok, case["norm_end"],
                                                                   case["ts_mode"], case["only_no_ts_mode"])
    except ValueError as e:
        return {"raises": "ValueError", "message_has": "negative" if "negative" in str(e) else str(e)[:40]}
    neg = np.isneginf(mask.numpy())
    cols = np.flatnonzero(neg.all(axis=0))
    text_len = int(cols[0]) if len(cols) else 448
    # column-only mask: every row identical, zeros before text_len, -inf from text_len on (train_timestamps.py:314-315)
    assert (neg == neg[0]).all() and neg[0, text_len:].all() and not neg[0, :text_len].any() and (mask.numpy()[:, :text_len] == 0).all()
    return {"text_input": ti[:text_len].tolist(), "text_y": ty[:text_len].tolist(), "text_len": text_len,
            "pad_input_ok": bool((ti[text_len:] == 51864).all()), "pad_y_ok": bool((ty[text_len:] == 51864).all()),
            "timestamp_mode": bool(ts_mode), "norm_end": norm_end if isinstance(norm_end, str) else int(norm_end)}


def sched_cases(mod):
    import torch
    out = []
    for train_steps, world, bs, eff in ((1000, 8, 8, 256), (1000, 8, 32, 256), (1048576, 8, 16, 2048), (7, 1, 4, 4), (500, 2, 3, 100), (1, 1, 1, 1)):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sched, accum, warmup, steps = mod.prepare_sched(train_steps, world, bs, eff, opt)
        probe = sorted({0, 1, 2, int(warmup) - 1, int(warmup), int(warmup) + 1, train_steps // 2, train_steps - 1, train_steps, train_steps + 5} - {-1})
        lam = sched.lr_lambdas[0]
        out.append({"train_steps": train_steps, "world_size": world, "train_batch_size": bs, "eff_batch_size": eff,
                    "accumulation_steps": int(accum), "warmup_steps": float(warmup), "lr_factor": {str(s): float(lam(s)) for s in probe}})
    return out


def main():
    from oracle import ref_train_import
    mod = ref_train_import.load()
    tok = ScriptedTokenizer()
    cs = cases()
    for c in cs:
        c["expect"] = run_reference(c, mod, tok)
    idx = [{"timestamp": t, "token": mod.AudioTextDataset._convert_to_token_idx(t, 50363)}
           for t in ("00:00:00.000", "00:00:00.019", "00:00:00.020", "00:00:29.999", "00:00:30.000", "00:00:30.001", "00:01:00.000", "01:00:00.000")]
    doc = {"_meta": {"generator": "oracle/gen_token_layout_golden.py", "reference": "scripts/training/train_timestamps.py:218-506, 739-783 (run unmodified)",
                     "tokenizer": "ScriptedTokenizer: crc32(word) mod 50256 per whitespace-separated word; English-only special ids",
                     "coin": "np.random.seed(case.seed) immediately before preprocess_text"},
           "cases": cs, "convert_to_token_idx": idx, "prepare_sched": sched_cases(mod)}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
        f.write("\n")
    n_ts = sum(1 for c in cs if c["expect"].get("timestamp_mode"))
    print(f"{OUT}: {len(cs)} cases ({n_ts} in timestamp mode, {sum('raises' in c['expect'] for c in cs)} raising), "
          f"{len(doc['prepare_sched'])} scheduler cases, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
