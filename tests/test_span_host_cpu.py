"""Host-side pieces of the supervised-span step that need no GPU: the span rule (OLMoASR.supervised_span) on the oracle's padded batches
(train_timestamps.py:238-343 layout via oracle.model_oracle.pad_sample), and the chunk-row table helper the GPU tests build their layouts with."""
import torch

from olmoasr_amd import _native as N
from olmoasr_amd import ops
from olmoasr_amd.model import OLMoASR
from oracle import model_oracle as mo

IGN = 51864


def test_supervised_span_bounds_every_position_that_can_carry_gradient():
    _, ti, ty, tl = mo.synthetic_batch(list(range(24)))
    span = OLMoASR.supervised_span(ty, tl)
    assert span.dtype == torch.int32 and span.device.type == "cpu" and span.shape == (24,)
    S = ty.shape[1]
    for b in range(24):
        s = int(span[b])
        assert int(tl[b]) <= s <= S
        assert bool((ty[b, s:] == IGN).all()), "a supervised target past the span"
        if s > int(tl[b]):  # the span was set by a target, not by the key mask: that target is the last supervised one
            assert int(ty[b, s - 1]) != IGN
    # hand-made corner cases: nothing supervised (span = text_len), everything supervised, text_len past the context
    t = torch.full((3, 128), IGN, dtype=torch.int64)
    t[1, :] = 7
    t[2, 5] = 9
    sp = OLMoASR.supervised_span(t, torch.tensor([10, 3, 400]))
    assert sp.tolist() == [10, 128, 128]


def test_chunk_rows_table_places_every_chunk_once_and_marks_the_rest():
    B, n = 3, 4
    order = [(1, 0), (0, 0), (2, 0), (0, 1), (2, 1), (1, 1), (0, 2), (1, 2), (2, 2), (0, 3), (1, 3), (2, 3)]
    tab = ops.chunk_rows_table(order, B, n)
    assert tab.shape == (B, N.ROWTAB) and tab.dtype == torch.int32
    rows = sorted(int(tab[b, c]) for b in range(B) for c in range(n))
    assert rows == [64 * i for i in range(B * n)]
    assert int(tab[1, 0]) == 0 and int(tab[0, 0]) == 64
    assert bool((tab[:, n:] == 0x3FFFFFFF).all())
    x = torch.arange(B * n * 64, dtype=torch.float32).reshape(B, n * 64, 1)
    xc = ops.to_chunked(x, tab)
    assert torch.equal(ops.from_chunked(xc, tab, B, n * 64), x)
    assert torch.equal(xc[:64, 0], x[1, :64, 0])
