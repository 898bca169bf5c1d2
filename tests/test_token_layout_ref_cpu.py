"""SURVEY.md section 8 row a18 -- the INTEGER token / mask layout of a training sample -- pinned against the reference RUNNING.

tests/golden/token_layout_ref.json was written by oracle/gen_token_layout_golden.py from the unmodified
/root/reference/scripts/training/train_timestamps.py (AudioTextDataset.preprocess_text and its helpers, :218-506; prepare_sched,
:739-783) on 54 scripted samples.  Bit-equality is asserted for
  * the oracle's restatement (oracle/model_oracle.py: preprocess_text on build_token_sequence / pad_sample, sched_rule),
  * the product (olmoasr_amd/text_layout.py; the teacher-forcing shift + 51864 padding shared with the synthetic generator,
    olmoasr_amd/synth.py::_layout; the train script's accumulation_steps / lr_lambda),
  * and, where /root/reference is mounted, the reference itself live (so the fixture cannot go stale silently).
"""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from oracle import model_oracle as mo
from oracle import ref_train_import
from oracle.gen_token_layout_golden import ScriptedTokenizer, run_reference, sched_cases
from olmoasr_amd import synth, text_layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = json.load(open(os.path.join(ROOT, "tests", "golden", "token_layout_ref.json")))
CASES = DOC["cases"]
TOK = ScriptedTokenizer()


def _expect_arrays(e):
    ti = torch.full((448,), 51864, dtype=torch.long)
    ty = torch.full((448,), 51864, dtype=torch.long)
    ti[:e["text_len"]] = torch.tensor(e["text_input"], dtype=torch.long)
    ty[:e["text_len"]] = torch.tensor(e["text_y"], dtype=torch.long)
    return ti, ty


def test_fixture_covers_every_branch_of_the_layout():
    ex = [c["expect"] for c in CASES]
    assert len(CASES) >= 40 and all(e.get("pad_input_ok", True) and e.get("pad_y_ok", True) for e in ex)
    first = {tuple(e["text_input"][:2]) for e in ex if "text_input" in e}
    assert (50257, 50362) in first and (50257, 50363) in first                                    # no-timestamp and timestamp layouts
    assert any(e.get("text_input", [])[2:3] == [50361] for e in ex)                               # <|nospeech|> for an empty >= 30 s segment
    assert any(e.get("timestamp_mode") for e in ex) and any(isinstance(e.get("norm_end"), str) for e in ex)  # > 30 s: norm_end becomes a string
    assert any("raises" in e for e in ex) and any(e.get("text_len") == 448 for e in ex)           # overflow raises; a sample that fills the context
    # the reference's two independent coins on an empty transcript: layout and flag disagree in some case
    emp = [e for c, e in zip(CASES, ex) if c["name"].startswith("empty_short/")]
    assert {(e["text_input"][1] == 50363, e["timestamp_mode"]) for e in emp} == {(True, True), (True, False), (False, True), (False, False)}


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_and_product_equal_the_reference(case):
    e = case["expect"]
    ext = case["subtitle_file"].split(".")[-1]
    cues = text_layout.read_transcript(case["seg_content"], ext)
    if "raises" in e:
        np.random.seed(case["seed"])
        with pytest.raises(ValueError):
            text_layout.preprocess_text(case["seg_content"], case["subtitle_file"], TOK, case["norm_end"], case["ts_mode"], case["only_no_ts_mode"])
        return
    ti_ref, ty_ref = _expect_arrays(e)
    # oracle
    np.random.seed(case["seed"])
    ti, ty, tl, ts, ne = mo.preprocess_text([((a, b), t) for a, b, t in cues], TOK.encode, case["norm_end"], case["ts_mode"], case["only_no_ts_mode"],
                                            np.random.rand)
    assert torch.equal(ti, ti_ref) and torch.equal(ty, ty_ref) and tl == e["text_len"] and bool(ts) == e["timestamp_mode"] and ne == e["norm_end"]
    # product
    np.random.seed(case["seed"])
    ti, ty, tl, ts, ne = text_layout.preprocess_text(case["seg_content"], case["subtitle_file"], TOK, case["norm_end"], case["ts_mode"],
                                                     case["only_no_ts_mode"])
    assert torch.equal(ti, ti_ref) and torch.equal(ty, ty_ref) and tl == e["text_len"] and ts is e["timestamp_mode"] and ne == e["norm_end"]
    assert type(ne) is type(e["norm_end"])
    # the synthetic generator's shift / pad on the same token list, and the span the loaders derive from it
    tokens = torch.tensor(e["text_input"] + e["text_y"][-1:], dtype=torch.long)
    si, sy, sl = synth._layout(tokens)
    assert torch.equal(si, ti_ref) and torch.equal(sy, ty_ref) and sl == e["text_len"]
    assert int(synth.supervised_span_host(sy[None], torch.tensor([sl]))[0]) == e["text_len"]
    # the loader plug built on it hands the same tokens to AudioTextShards
    np.random.seed(case["seed"])
    toks, ts2, ne2 = text_layout.reference_text_fn(TOK)(case)
    assert toks == tokens.tolist() and ts2 is e["timestamp_mode"] and ne2 == e["norm_end"]


def test_timestamp_token_rule():
    for row in DOC["convert_to_token_idx"]:
        assert text_layout.timestamp_token(row["timestamp"], 50363) == row["token"]
        assert mo.timestamp_token(mo.ms_of(row["timestamp"])) == row["token"]
    with pytest.raises(ValueError):
        text_layout.to_ms("12.5")


def _train_script():
    spec = importlib.util.spec_from_file_location("oasr_train_cli", os.path.join(ROOT, "scripts", "training", "train_timestamps.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_accumulation_and_schedule_rule():
    ts = _train_script()
    for c in DOC["prepare_sched"]:
        accum, warmup, factor = mo.sched_rule(c["train_steps"], c["world_size"], c["train_batch_size"], c["eff_batch_size"])
        assert accum == c["accumulation_steps"] and float(warmup) == c["warmup_steps"]
        assert ts.accumulation_steps(c["eff_batch_size"], c["world_size"], c["train_batch_size"]) == c["accumulation_steps"]
        for s, f in c["lr_factor"].items():
            assert factor(int(s)) == f and ts.lr_lambda(int(s), c["train_steps"]) == f  # exact: the same float operations


@pytest.mark.skipif(not ref_train_import.available(), reason="/root/reference is not mounted (GPU box): the committed fixture stands in")
def test_fixture_equals_the_reference_live():
    mod = ref_train_import.load()
    for c in CASES:
        assert run_reference(c, mod, TOK) == c["expect"], c["name"]
    assert sched_cases(mod) == DOC["prepare_sched"]
    for row in DOC["convert_to_token_idx"]:
        assert mod.AudioTextDataset._convert_to_token_idx(row["timestamp"], 50363) == row["token"]
