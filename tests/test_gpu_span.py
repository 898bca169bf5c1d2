"""The supervised-span micro-step (oasr_train_fwd_bwd_span, olmoasr_amd/csrc/engine.hip): the decoder's token rows live in
64-position chunks with every chunk that can carry gradient first, and the decoder's backward runs on those rows only.

 * the tables it builds (chunk rows, spans, targets in row order) against a host restatement;
 * every attention kernel on chunked rows against the SAME kernel on the plain layout: bit-identical (only addresses change),
   and with a span the rows past it are neither read (poisoned with NaN here) nor written;
 * the whole step against the plain step (reference schedule: train_timestamps.py:1440-1454 over all 448 padded positions,
   :318-329, :1444): same loss, gradients equal up to fp32 summation order -- bf16 engine and fp32 validation engine."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
DEV = "cuda"
OOR = 0x3FFFFFFF


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def host_tables(span, S):
    """Restatement of build_span_tables_kernel: active chunks first, position-block-major, then the inactive ones."""
    B, nch = len(span), S // 64
    act = [(b, c) for c in range(nch) for b in range(B) if 64 * c < span[b]]
    ina = [(b, c) for c in range(nch) for b in range(B) if not 64 * c < span[b]]
    rows = [[OOR] * 16 for _ in range(B)]
    for i, (b, c) in enumerate(act + ina):
        rows[b][c] = 64 * i
    return rows, 64 * len(act)


@pytest.mark.parametrize("B,S,seed", [(1, 448, 0), (5, 448, 1), (128, 448, 2), (3, 64, 3), (7, 1024, 4)])
def test_span_tables(B, S, seed):
    from olmoasr_amd import _native as N
    g = torch.Generator().manual_seed(seed)
    span = torch.randint(0, S + 1, (B,), generator=g, dtype=torch.int32)
    span[0] = S if seed % 2 else 1
    if B > 2:
        span[1], span[2] = 64, min(65, S)
    targets = torch.randint(0, 50000, (B, S), generator=g, dtype=torch.int64)
    targets[torch.arange(S)[None, :] >= span[:, None]] = 51864
    tg = targets.to(DEV)
    rows = torch.full((B, 16), -1, dtype=torch.int32, device=DEV)
    span_d = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    tphys = torch.full((B * S,), -7, dtype=torch.int64, device=DEV)
    act = C.c_int64(0)
    N.check(N.lib().oasr_test_span_tables(C.c_void_p(span.data_ptr()), B, S, N.ptr(tg), N.ptr(rows), N.ptr(span_d), N.ptr(tphys),
                                          C.byref(act), N.stream_ptr()), "span tables")
    torch.cuda.synchronize()
    ref_rows, ref_act = host_tables(span.tolist(), S)
    assert act.value == ref_act
    assert rows.cpu().tolist() == ref_rows
    assert span_d.cpu().tolist() == [(int(s) + 63) // 64 * 64 for s in span]
    # every row is used exactly once, and the targets of the active rows are the samples' targets chunk by chunk
    flat = sorted(r for row in ref_rows for r in row if r != OOR)
    assert flat == list(range(0, B * S, 64))
    tp = tphys.cpu()
    for b in range(B):
        for c in range(S // 64):
            r = ref_rows[b][c]
            if 64 * c < int(span[b]):
                assert r < ref_act and torch.equal(tp[r:r + 64], targets[b, 64 * c:64 * c + 64])
            else:
                assert r >= ref_act and bool((tp[r:r + 64] == -7).all())  # not touched


def _cs_close(a, b):
    """fused bias-gradient column sums: the per-workgroup partial rows are identical, their fp32 atomic reduction order is not"""
    return bool(((a - b).abs() <= 1e-5 * b.abs() + 1e-5 * float(b.abs().max())).all())


def _placement(B, n_chunks, seed):
    """A scrambled chunk placement (any placement is legal for the kernels)."""
    from olmoasr_amd import ops
    g = torch.Generator().manual_seed(seed)
    pairs = [(b, c) for b in range(B) for c in range(n_chunks)]
    order = [pairs[i] for i in torch.randperm(len(pairs), generator=g).tolist()]
    return ops.chunk_rows_table(order, B, n_chunks)


@pytest.fixture(params=[1, 0], ids=["pingpong", "general"])
def attn_path(request):
    from olmoasr_amd import _native as N
    N.lib().oasr_attention_set_pingpong(request.param)
    yield request.param
    N.lib().oasr_attention_set_pingpong(1)


@pytest.mark.parametrize("kind", ["decoder-self", "cross"])
def test_attention_on_chunked_rows_is_bit_identical(kind, attn_path):
    """Same kernels, same arithmetic, different addresses: outputs must be torch.equal to the plain layout's."""
    from olmoasr_amd import ops
    if kind == "decoder-self" and attn_path == 0:
        pytest.skip("masked cases always run the general kernels")
    B, H, Tq = 5, 3, 448
    d = H * 64
    causal = kind == "decoder-self"
    Tk = Tq if causal else 1500
    kv_len = torch.tensor([7, 220, 448, 64, 129], dtype=torch.int32, device=DEV) if causal else None
    tab = _placement(B, Tq // 64, 11)
    tab_d = tab.to(DEV)
    if causal:
        qkv = rnd(B, Tq, 3 * d, seed=15)
        q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(3))
        qkv_c = ops.to_chunked(qkv, tab)
        qc, kc, vc = (qkv_c[:, i * d:(i + 1) * d].unflatten(1, (H, 64)) for i in range(3))
        k_rows = tab_d
    else:
        qb, kvb = rnd(B, Tq, d, seed=16), rnd(B, Tk, 2 * d, seed=17)
        q = qb.unflatten(2, (H, 64))
        k, v = (kvb[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(2))
        qc = ops.to_chunked(qb, tab).unflatten(1, (H, 64))
        kc, vc, k_rows = k, v, None
    o, lse, o_lo = ops.attention_fwd(q, k, v, kv_len, causal, want_o_lo=True)
    oc, lse_c, o_lo_c = ops.attention_fwd_rows(qc, kc, vc, B, H, Tq, Tk, tab_d, k_rows, kv_len, causal, want_o_lo=True)
    assert torch.equal(ops.from_chunked(oc, tab, B, Tq), o) and torch.equal(lse_c, lse)
    assert torch.equal(ops.from_chunked(o_lo_c, tab, B, Tq), o_lo)

    # ---- backward, no span: everything equal, including the fused bias-gradient column sums
    d_o = rnd(B, Tq, d, seed=18, scale=0.5)
    doc = ops.to_chunked(d_o, tab)
    cs = [torch.zeros(d, device=DEV) for _ in range(4)]
    dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, d_o, kv_len, causal, o_lo=o_lo, dq_colsum=cs[0], dv_colsum=cs[1])
    dqc, dkc, dvc = ops.attention_bwd_rows(qc, kc, vc, oc, lse, doc, B, H, Tq, Tk, tab_d, k_rows, None, kv_len, causal, o_lo=o_lo_c,
                                           dq_colsum=cs[2], dv_colsum=cs[3])
    if causal:  # dq | dk | dv share the fused qkv buffer's strides: compare through the same views
        for got, ref in ((dqc, dq), (dkc, dk), (dvc, dv)):
            assert torch.equal(ops.from_chunked(got.reshape(B * Tq, d), tab, B, Tq), ref.reshape(B, Tq, d))
    else:
        assert torch.equal(ops.from_chunked(dqc.reshape(B * Tq, d), tab, B, Tq), dq.reshape(B, Tq, d))
        assert torch.equal(dkc, dk) and torch.equal(dvc, dv)
    assert _cs_close(cs[2], cs[0]) and _cs_close(cs[3], cs[1])

    # ---- backward with a span: d_o is zero past it in the plain run; on chunked rows those rows hold NaN and must not be read,
    # and the gradient rows past the span must not be written
    span = torch.tensor([64, 256, 448, 64, 192], dtype=torch.int32)
    pos = torch.arange(Tq)[None, :, None]
    keep = (pos < span[:, None, None]).to(DEV)
    d_o0 = torch.where(keep, d_o, torch.zeros_like(d_o))
    doc0 = ops.to_chunked(torch.where(keep, d_o, torch.full_like(d_o, float("nan"))), tab)
    oc_p = ops.to_chunked(torch.where(keep, o, torch.full_like(o, float("nan"))), tab)
    olo_p = ops.to_chunked(torch.where(keep, o_lo, torch.full_like(o_lo, float("nan"))), tab)
    cs = [torch.zeros(d, device=DEV) for _ in range(4)]
    dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, d_o0, kv_len, causal, o_lo=o_lo, dq_colsum=cs[0], dv_colsum=cs[1])
    dqc, dkc, dvc = ops.attention_bwd_rows(qc, kc, vc, oc_p, lse, doc0, B, H, Tq, Tk, tab_d, k_rows, span.to(DEV), kv_len, causal,
                                           o_lo=olo_p, dq_colsum=cs[2], dv_colsum=cs[3], fill=float("nan"))
    keep_q = keep.expand(B, Tq, d)
    got_q = ops.from_chunked(dqc.reshape(B * Tq, d) if not causal else dqc.reshape(B * Tq, d), tab, B, Tq)
    ref_q = dq.reshape(B, Tq, d)
    assert torch.equal(got_q[keep_q], ref_q[keep_q]), "dq inside the span"
    assert bool(torch.isnan(got_q[~keep_q]).all()), "dq rows past the span must stay untouched"
    assert float(ref_q[~keep_q].abs().max() if bool((~keep_q).any()) else 0.0) == 0.0
    if causal:
        for got, ref, nm in ((dkc, dk, "dk"), (dvc, dv, "dv")):
            g2, r2 = ops.from_chunked(got.reshape(B * Tq, d), tab, B, Tq), ref.reshape(B, Tq, d)
            assert torch.equal(g2[keep_q], r2[keep_q]), nm
            assert bool(torch.isnan(g2[~keep_q]).all()), nm + " rows past the span must stay untouched"
            assert float(r2[~keep_q].abs().max()) == 0.0
    else:
        assert torch.equal(dkc, dk) and torch.equal(dvc, dv)
    assert _cs_close(cs[2], cs[0]) and _cs_close(cs[3], cs[1])


def _grads_by_tensor(net):
    return {n: p.grad.detach().clone() for n, p in net.named_parameters()}


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("variant,B,dtype", [("tiny", 6, "bfloat16"), ("tiny", 6, "float32"), ("base", 9, "bfloat16"), ("base", 3, "float32")])
def test_span_step_equals_plain_step(variant, B, dtype):
    """loss and every gradient of oasr_train_fwd_bwd_span == oasr_train_fwd_bwd (the reference schedule over all 448 positions)."""
    from olmoasr_amd import ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd.synth import synth_samples
    net = OLMoASR(VARIANT_TO_DIMS[variant], device=DEV, seed=0, compute_dtype=dtype)
    pcm, ti, ty, tl = synth_samples(list(range(70, 70 + B)), DEV)
    mel = ops.log_mel(pcm)
    net.zero_grad()
    loss0, _ = net.loss_and_backward(mel, ti, ty, tl, loss_scale=1024.0)
    torch.cuda.synchronize()
    g0 = _grads_by_tensor(net)
    f0 = net.flat_grads.clone()
    for mode in ("derived", "host", "loose", "forward"):
        net.zero_grad()
        if mode in ("derived", "forward"):  # "forward": the opt-in that also leaves the padded positions out of the decoder's forward
            span = True
        else:
            span = net.supervised_span(ty, tl)
            assert span.tolist() == tl.cpu().tolist()  # the synthetic targets are supervised exactly up to text_len
            if mode == "loose":  # any upper bound is legal: more rows than needed, same result
                span = (span + torch.tensor([0, 70, 500, 1, 64, 129, 13, 5, 300][:B], dtype=torch.int32)).clamp(max=448)
        loss1, lg = net.loss_and_backward(mel, ti, ty, tl, loss_scale=1024.0, span=span, span_forward=mode == "forward")
        torch.cuda.synchronize()
        assert lg is None
        tol_l, tol_g, tol_t = (1e-6, 1e-5, 1e-4)  # measured: loss equal to 7 digits, gradients 1e-7 (total) / 1e-6 (worst tensor), both engines
        assert abs(float(loss1) - float(loss0)) <= tol_l * abs(float(loss0)), (mode, float(loss1), float(loss0))
        worst = max((_rel(p.grad, g0[n]), n) for n, p in net.named_parameters())
        total = _rel(net.flat_grads, f0)
        print(f"   span step ({variant}, B={B}, {dtype}, {mode}): loss {float(loss1):.6f} vs {float(loss0):.6f}, grads rel-L2 {total:.2e}, worst tensor {worst[0]:.2e} {worst[1]}")
        assert total <= tol_g and worst[0] <= tol_t, (mode, total, worst)
    del net
    torch.cuda.empty_cache()


@pytest.mark.parametrize("variant,B,dtype", [("tiny", 6, "bfloat16"), ("base", 9, "bfloat16"), ("tiny", 4, "float32")])
def test_span_step_side_streams_change_nothing_but_the_summation_order(variant, B, dtype, monkeypatch):
    """The span step's side streams (csrc/engine.hip Runner::side_mode: weight gradients over the active rows, the cross-attention
    key|value projections and their gradients on lowest-priority streams beside the main chain) against the single-stream step:
    same loss (the forward's arithmetic does not change), gradients equal up to the order of the fp32 atomics --
    with and without per-segment events (the events force the key|value gradients' join at every block: the DDP path), twice in a
    row on the same workspace (a missing join would let step 2 overwrite what step 1's side work still reads)."""
    from olmoasr_amd import _native as N
    from olmoasr_amd import ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd.synth import synth_samples
    monkeypatch.setenv("OASR_TESTING_HOOKS", "1")
    net = OLMoASR(VARIANT_TO_DIMS[variant], device=DEV, seed=0, compute_dtype=dtype)
    pcm, ti, ty, tl = synth_samples(list(range(40, 40 + B)), DEV)
    mel = ops.log_mel(pcm)
    lib = N.lib()
    try:
        results = {}
        for mode in (0, 1, 7, 15):
            N.check(lib.oasr_span_set_side_streams(mode), "side streams")
            assert lib.oasr_span_side_streams() == mode
            for with_events in (False, True):
                ev = None
                if with_events:
                    ev = [torch.cuda.Event() for _ in net.grad_segments]
                    for e in ev:
                        e.record()
                for rep in range(2):
                    net.zero_grad()
                    loss, _ = net.loss_and_backward(mel, ti, ty, tl, loss_scale=1024.0, span=True, segment_events=ev)
                torch.cuda.synchronize()
                results[(mode, with_events)] = (float(loss), net.flat_grads.clone())
        l0, g0 = results[(0, False)]
        for key, (l, g) in results.items():
            rel = _rel(g, g0)
            print(f"   side streams {key} ({variant}, B={B}, {dtype}): loss {l:.7f} vs {l0:.7f}, grads rel-L2 {rel:.2e}")
            assert abs(l - l0) <= 1e-6 * abs(l0), (key, l, l0)
            assert rel <= 1e-6, (key, rel)  # measured 4-6e-8: the spread between two runs of the single-stream step itself
            assert bool(torch.isfinite(g).all())
    finally:
        lib.oasr_span_set_side_streams(-1)
    del net
    torch.cuda.empty_cache()


def test_span_step_rejects_bad_spans_and_falls_back():
    from olmoasr_amd import _native as N
    from olmoasr_amd import ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd.synth import synth_samples
    net = OLMoASR(VARIANT_TO_DIMS["tiny"], device=DEV, seed=0)
    pcm, ti, ty, tl = synth_samples([1, 2], DEV)
    mel = ops.log_mel(pcm)
    net.zero_grad()
    with pytest.raises(N.NativeError):
        net.loss_and_backward(mel, ti, ty, tl, span=[449, 10])
    with pytest.raises(N.NativeError):
        net.loss_and_backward(mel, ti, ty, tl, span=[0, 0])
    with pytest.raises(ValueError):
        net.loss_and_backward(mel, ti, ty, tl, span=True, return_logits=True)
    del net
    torch.cuda.empty_cache()


def test_unfinalized_log_mel_through_the_span_step_is_bit_identical():
    """oasr_log_mel_raw + mel_clip_max: whisper's last two lines (floor at the clip maximum - 8, (x + 4) / 4) applied inside the encoder's
    time-major transpose instead of a second pass over the tensor -- same fp32 operations, so the finalized tensor, the loss and the
    gradients' bits do not change."""
    from olmoasr_amd import ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd.synth import synth_samples
    pcm, ti, ty, tl = synth_samples([3, 4, 5], DEV)
    pcm[1, 200000:] = 0  # a long silent tail: many values at the floor
    mel = ops.log_mel(pcm)
    raw, cm = ops.log_mel(pcm, finalize=False)
    assert torch.equal((torch.maximum(raw, (cm - 8.0)[:, None, None]) + 4.0) * 0.25, mel)
    assert torch.equal(cm, raw.amax((1, 2)))
    for dtype in ("bfloat16", "float32"):
        net = OLMoASR(VARIANT_TO_DIMS["tiny"], device=DEV, seed=0, compute_dtype=dtype)
        net.zero_grad()
        l0, _ = net.loss_and_backward(mel, ti, ty, tl, span=True)
        # (split-K weight-gradient sums are fp32 atomics: compare the deterministic pieces -- loss -- exactly, gradients to 1e-6)
        g0 = net.flat_grads.clone()
        net.zero_grad()
        l1, _ = net.loss_and_backward(raw, ti, ty, tl, span=True, mel_clip_max=cm)
        torch.cuda.synchronize()
        assert float(l0) == float(l1), (dtype, float(l0), float(l1))
        assert _rel(net.flat_grads, g0) < 1e-6
        with pytest.raises(ValueError):
            net.loss_and_backward(raw, ti, ty, tl, mel_clip_max=cm)
        del net
    torch.cuda.empty_cache()
