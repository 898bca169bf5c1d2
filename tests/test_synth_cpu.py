"""Integer work, bit-exact: the product's synthetic sample generator (olmoasr_amd/synth.py) against the oracle's
restatement of the reference's token layout (oracle/model_oracle.py: build_token_sequence / pad_sample following
scripts/training/train_timestamps.py:218-236,301-329,401-506), in both layouts (no-timestamp and timestamp mode)."""
import torch

from olmoasr_amd import synth
from oracle import model_oracle as mo


def _same(a, b):
    assert torch.equal(a[0], b[0]) and a[0].dtype == torch.int16
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[1].dtype == torch.int64
    assert int(a[3]) == int(b[3])


def test_synth_sample_equals_oracle_no_timestamps():
    for i in list(range(12)) + [1000, 99999]:
        _same(synth.synth_sample(i), mo.synthetic_sample(i))


def test_synth_sample_equals_oracle_timestamp_layout():
    seen_multi = False
    for i in list(range(24)) + [4242]:
        a, b = synth.synth_sample(i, timestamps=True), mo.synthetic_sample(i, timestamps=True)
        _same(a, b)
        ti, ty, L = a[1], a[2], int(a[3])
        toks = torch.cat([ti[:1], ty[:L]]).tolist()  # the unshifted sequence
        # structure of the reference's timestamp mode (train_timestamps.py:462-506)
        assert toks[0] == mo.SOT and toks[-1] == mo.EOT and mo.NO_TIMESTAMPS not in toks
        stamps = [t for t in toks if t >= mo.TIMESTAMP_BEGIN]
        assert len(stamps) % 2 == 1 and stamps[:-1] == sorted(stamps[:-1])  # (start, end) pairs + the final next-start stamp
        assert all(mo.TIMESTAMP_BEGIN <= t <= mo.TIMESTAMP_BEGIN + 1500 for t in stamps)  # <= 30 s at 20 ms per token
        assert toks[1] >= mo.TIMESTAMP_BEGIN and toks[-2] >= mo.TIMESTAMP_BEGIN and toks[-3] >= mo.TIMESTAMP_BEGIN
        assert (ti[L:] == mo.PAD_ID).all() and (ty[L:] == mo.PAD_ID).all() and ti[L - 1] != mo.PAD_ID
        seen_multi |= len(stamps) > 3
        # same text body as the no-timestamp layout of the same index
        plain = mo.synthetic_sample(i)
        body_plain = [t for t in plain[1][2:int(plain[3])].tolist()]
        assert [t for t in toks[1:-1] if t < mo.TIMESTAMP_BEGIN] == body_plain
    assert seen_multi


def test_token_sequence_rules():
    # a boundary past 30 s falls back to the no-timestamp layout (train_timestamps.py:437-452)
    toks, ts = mo.build_token_sequence([(0, 31000, [5, 6])], 31000, True)
    assert not ts and toks == [mo.SOT, mo.NO_TIMESTAMPS, 5, 6, mo.EOT]
    toks, ts = mo.build_token_sequence([(0, 1000, [5]), (1500, 29980, [6, 7])], 29980, True)
    assert ts and toks == [mo.SOT, 50363, 5, 50413, 50438, 6, 7, 50363 + 1499, 50363 + 1499, mo.EOT]
    assert mo.timestamp_token(30000) == 51863 and mo.timestamp_token(30020) is None  # <= n_vocab - 1 (model_dims.py:35)


def test_batch_helpers_agree():
    a = synth.synth_samples([3, 4], "cpu", timestamps=True)
    b = mo.synthetic_batch([3, 4], timestamps=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
