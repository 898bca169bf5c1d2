"""Thin tensor-level wrappers over the unit operators of liboasr (used by the op-level parity tests and by the
host-side mirrors).  bf16 tensors are torch.bfloat16; everything runs on the current HIP stream."""
import ctypes as C

import torch

from . import _native as N

BF = torch.bfloat16


def _op(t, ld=None, rpb=0, bstride=0, lead=0, kvalid=0, trail_from=0):
    return N.Operand(t.data_ptr(), ld if ld is not None else t.stride(0), rpb, bstride, lead, kvalid, trail_from)


def gemm(A, B, M, N_, K, *, ta=False, tb=False, a_view=None, b_view=None, bias=None, act=0, pos=None, pos_period=0,
         dgelu_u=None, resid=None, out=None, out_pre=None, ldc=None, out_f32=None, beta=0.0, atomic=False, split_k=1,
         alpha=1.0, colsum=None, dgelu_deriv=False):
    """C[M,N] = epilogue(alpha * sum_k A(m,k) B(n,k)); see olmoasr_amd/csrc/kernels.h GemmArgs."""
    g = N.GemmArgs()
    g.A = a_view if a_view is not None else _op(A)
    g.B = b_view if b_view is not None else _op(B)
    g.M, g.N, g.K, g.ta, g.tb, g.alpha = M, N_, K, int(ta), int(tb), alpha
    g.bias = bias.data_ptr() if bias is not None else None
    g.act = act
    g.pos = pos.data_ptr() if pos is not None else None
    g.pos_period = pos_period
    g.dgelu_u = dgelu_u.data_ptr() if dgelu_u is not None else None
    g.ldu = dgelu_u.stride(0) if dgelu_u is not None else 0
    g.resid = resid.data_ptr() if resid is not None else None
    g.ldr = resid.stride(0) if resid is not None else 0
    g.out = out.data_ptr() if out is not None else None
    g.out_pre = out_pre.data_ptr() if out_pre is not None else None
    g.ldc = ldc if ldc is not None else (out.stride(0) if out is not None else (out_pre.stride(0) if out_pre is not None else 0))
    g.out_f32 = out_f32.data_ptr() if out_f32 is not None else None
    g.ldc32 = out_f32.stride(0) if out_f32 is not None else 0
    g.beta, g.atomic, g.split_k = beta, int(atomic), split_k
    g.colsum = colsum.data_ptr() if colsum is not None else None
    g.dgelu_deriv = int(dgelu_deriv)
    N.check(N.lib().oasr_gemm(C.byref(g), N.stream_ptr()), "oasr_gemm")


def layernorm_fwd(x, gamma, beta):
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    N.check(N.lib().oasr_layernorm_fwd(N.ptr(x), N.ptr(gamma), N.ptr(beta), N.ptr(y), N.ptr(mean), N.ptr(rstd), rows, d,
                                       N.stream_ptr()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None):
    rows, d = x.shape
    dx = torch.empty_like(x)
    dg = torch.zeros(d, device=x.device, dtype=torch.float32)
    db = torch.zeros_like(dg)
    N.check(N.lib().oasr_layernorm_bwd(N.ptr(dy), N.ptr(x), N.ptr(gamma), N.ptr(mean), N.ptr(rstd), N.ptr(dres), N.ptr(dx),
                                       N.ptr(dg), N.ptr(db), rows, d, N.stream_ptr()), "layernorm_bwd")
    return dx, dg, db


def _attn_args(q, k, v, o, lse, kv_len, causal):
    B, Tq, H, D = q.shape
    Tk = k.shape[1]
    assert D == 64 and q.stride(3) == 1 and q.stride(2) == 64
    a = N.AttnArgs()
    a.q, a.k, a.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    a.ldq, a.ldk, a.ldv = q.stride(1), k.stride(1), v.stride(1)
    a.bsq, a.bsk, a.bsv = q.stride(0), k.stride(0), v.stride(0)
    a.o, a.ldo, a.bso = o.data_ptr(), o.stride(1), o.stride(0)
    a.lse = lse.data_ptr()
    a.kv_len = kv_len.data_ptr() if kv_len is not None else None
    a.B, a.H, a.Tq, a.Tk, a.causal = B, H, Tq, Tk, int(causal)
    return a


def attention_fwd(q, k, v, kv_len=None, causal=False, want_o_lo=False):
    """q [B,Tq,H,64], k/v [B,Tk,H,64] (any token/batch strides) -> o [B,Tq,H*64], lse [B,H,Tq] (, o_lo = bf16 rounding residual of o)."""
    B, Tq, H, _ = q.shape
    o = torch.empty(B, Tq, H * 64, device=q.device, dtype=BF)
    lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
    a = _attn_args(q, k, v, o.view(B, Tq, H, 64), lse, kv_len, causal)
    o_lo = torch.empty(B, Tq, H * 64, device=q.device, dtype=torch.bfloat16) if want_o_lo else None
    a.o_lo = o_lo.data_ptr() if want_o_lo else None
    N.check(N.lib().oasr_attention_fwd(C.byref(a), N.stream_ptr()), "attention_fwd")
    return (o, lse, o_lo) if want_o_lo else (o, lse)


def attention_scores(q, k, kv_len=None, causal=False):
    """``qk`` of the reference's manual attention path (MultiHeadAttention.qkv_attention, olmoasr/model.py:347-442): q [B,Tq,H,64],
    k [B,Tk,H,64] (bf16 or fp32, any token/batch strides) -> fp32 [B, H, Tq, Tk] pre-softmax scaled scores, -inf where masked."""
    B, Tq, H, _ = q.shape
    Tk = k.shape[1]
    assert q.dtype == k.dtype and q.dtype in (BF, torch.float32)
    assert q.stride(3) == 1 and q.stride(2) == 64 and k.stride(3) == 1 and k.stride(2) == 64
    out = torch.empty(B, H, Tq, Tk, device=q.device, dtype=torch.float32)
    a = N.AttnArgs()
    a.q, a.k = q.data_ptr(), k.data_ptr()
    a.ldq, a.ldk, a.bsq, a.bsk = q.stride(1), k.stride(1), q.stride(0), k.stride(0)
    a.kv_len = kv_len.data_ptr() if kv_len is not None else None
    a.B, a.H, a.Tq, a.Tk, a.causal = B, H, Tq, Tk, int(causal)
    N.check(N.lib().oasr_attention_scores(C.byref(a), 0 if q.dtype == BF else 1, N.ptr(out), N.stream_ptr()), "attention_scores")
    return out


def attention_bwd(q, k, v, o, lse, d_o, kv_len=None, causal=False, o_lo=None, dq_colsum=None, dv_colsum=None, qtile_flags=None):
    """qtile_flags: optional int32 [B, H, ceil(Tq/64)] workspace -- all-zero 64-query tiles of d_o are recorded and skipped (bit-identical)."""
    B, Tq, H, _ = q.shape
    # gradients use the operands' own (possibly fused-qkv) strides
    dq, dk, dv = (torch.empty_strided(t.shape, t.stride(), device=t.device, dtype=t.dtype) for t in (q, k, v))
    delta = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
    a = _attn_args(q, k, v, o.view(B, Tq, H, 64), lse, kv_len, causal)
    a.d_o, a.delta = d_o.data_ptr(), delta.data_ptr()
    a.o_lo = o_lo.data_ptr() if o_lo is not None else None
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.dq_colsum = dq_colsum.data_ptr() if dq_colsum is not None else None
    a.dv_colsum = dv_colsum.data_ptr() if dv_colsum is not None else None
    a.qtile_flags = qtile_flags.data_ptr() if qtile_flags is not None else None
    scratch = None
    if dq_colsum is not None or dv_colsum is not None:
        Tk = k.shape[1]
        scratch = torch.empty(B * ((Tq + 127) // 128 + (Tk + 127) // 128) * H * 64, device=q.device, dtype=torch.float32)
        a.colsum_scratch = scratch.data_ptr()
    N.check(N.lib().oasr_attention_bwd(C.byref(a), N.stream_ptr()), "attention_bwd")
    return dq, dk, dv


# ---- chunked token rows (include/oasr.h: oasr_attn_args.q_rows / k_rows / q_span) -----------------------------------------------------
def chunk_rows_table(order, B, n_chunks):
    """Chunk-row table int32 [B, ROWTAB] (CPU) for a given placement: ``order`` lists the (b, chunk) pairs in the order their 64 rows
    appear in memory.  Entries past ``n_chunks`` hold the out-of-range sentinel the kernels expect."""
    tab = torch.full((B, N.ROWTAB), 0x3FFFFFFF, dtype=torch.int32)
    for i, (b, c) in enumerate(order):
        tab[b, c] = 64 * i
    assert int((tab[:, :n_chunks] == 0x3FFFFFFF).sum()) == 0, "every (b, chunk) needs a place"
    return tab


def to_chunked(x, tab):
    """x [B, T, ...] -> [B*T, ...] with the 64-position chunk c of sample b at rows tab[b, c] .. +63 (test helper)."""
    B, T = x.shape[:2]
    out = torch.empty((B * T,) + tuple(x.shape[2:]), device=x.device, dtype=x.dtype)
    for b in range(B):
        for c in range(T // 64):
            r = int(tab[b, c])
            out[r:r + 64] = x[b, 64 * c:64 * c + 64]
    return out


def from_chunked(xc, tab, B, T):
    out = torch.empty((B, T) + tuple(xc.shape[1:]), device=xc.device, dtype=xc.dtype)
    for b in range(B):
        for c in range(T // 64):
            r = int(tab[b, c])
            out[b, 64 * c:64 * c + 64] = xc[r:r + 64]
    return out


def _attn_args_rows(qc, kc, vc, oc, lse, B, H, Tq, Tk, q_rows, k_rows, kv_len, causal):
    """qc / oc: chunked [B*Tq, H, 64] views (any token stride); kc / vc: chunked [B*Tk, H, 64] when k_rows is given, else plain [B, Tk, H, 64]."""
    a = N.AttnArgs()
    a.q, a.k, a.v, a.o = qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), oc.data_ptr()
    a.ldq, a.ldo = qc.stride(0), oc.stride(0)
    if k_rows is not None:
        a.ldk, a.ldv = kc.stride(0), vc.stride(0)
        a.k_rows = k_rows.data_ptr()
    else:
        a.ldk, a.ldv, a.bsk, a.bsv = kc.stride(1), vc.stride(1), kc.stride(0), vc.stride(0)
    a.lse = lse.data_ptr()
    a.kv_len = kv_len.data_ptr() if kv_len is not None else None
    a.B, a.H, a.Tq, a.Tk, a.causal = B, H, Tq, Tk, int(causal)
    a.q_rows = q_rows.data_ptr()
    return a


def attention_fwd_rows(qc, kc, vc, B, H, Tq, Tk, q_rows, k_rows=None, kv_len=None, causal=False, want_o_lo=False):
    """attention_fwd on chunked token rows: returns oc [B*Tq, H*64] (chunked like qc), lse [B, H, Tq] (logical)(, o_lo)."""
    oc = torch.empty(B * Tq, H * 64, device=qc.device, dtype=BF)
    lse = torch.empty(B, H, Tq, device=qc.device, dtype=torch.float32)
    a = _attn_args_rows(qc, kc, vc, oc.view(B * Tq, H, 64), lse, B, H, Tq, Tk, q_rows, k_rows, kv_len, causal)
    o_lo = torch.empty_like(oc) if want_o_lo else None
    a.o_lo = o_lo.data_ptr() if want_o_lo else None
    N.check(N.lib().oasr_attention_fwd(C.byref(a), N.stream_ptr()), "attention_fwd(rows)")
    return (oc, lse, o_lo) if want_o_lo else (oc, lse)


def attention_bwd_rows(qc, kc, vc, oc, lse, doc, B, H, Tq, Tk, q_rows, k_rows=None, q_span=None, kv_len=None, causal=False, o_lo=None,
                       dq_colsum=None, dv_colsum=None, fill=None):
    """attention_bwd on chunked token rows.  Gradients are allocated with the operands' strides and pre-filled with ``fill`` (e.g. NaN)
    so that a test can see which rows the kernels left untouched."""
    def like(t):
        g = torch.empty_strided(t.shape, t.stride(), device=t.device, dtype=t.dtype)
        if fill is not None:
            g.fill_(fill)
        return g
    dq, dk, dv = like(qc), like(kc), like(vc)
    delta = torch.zeros(B, H, Tq, device=qc.device, dtype=torch.float32)
    a = _attn_args_rows(qc, kc, vc, oc.view(B * Tq, H, 64), lse, B, H, Tq, Tk, q_rows, k_rows, kv_len, causal)
    a.d_o, a.delta = doc.data_ptr(), delta.data_ptr()
    a.o_lo = o_lo.data_ptr() if o_lo is not None else None
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.q_span = q_span.data_ptr() if q_span is not None else None
    a.dq_colsum = dq_colsum.data_ptr() if dq_colsum is not None else None
    a.dv_colsum = dv_colsum.data_ptr() if dv_colsum is not None else None
    scratch = None
    if dq_colsum is not None or dv_colsum is not None:
        scratch = torch.empty(B * ((Tq + 127) // 128 + (Tk + 127) // 128) * H * 64, device=qc.device, dtype=torch.float32)
        a.colsum_scratch = scratch.data_ptr()
    N.check(N.lib().oasr_attention_bwd(C.byref(a), N.stream_ptr()), "attention_bwd(rows)")
    return dq, dk, dv


def cross_entropy_(logits, V, targets, ignore, gscale=1.0, write_grad=True):
    """In place on bf16 logits [rows, ld]: returns (mean loss over non-ignored rows, row_loss); logits become the gradient."""
    rows, ld = logits.shape
    nv = torch.zeros(1, device=logits.device, dtype=torch.int32)
    row_loss = torch.empty(rows, device=logits.device, dtype=torch.float32)
    loss = torch.zeros(1, device=logits.device, dtype=torch.float32)
    N.check(N.lib().oasr_cross_entropy(N.ptr(logits), logits.stride(0), V, N.ptr(targets), rows, ignore, gscale, N.ptr(nv),
                                       N.ptr(row_loss), N.ptr(loss), int(write_grad), N.stream_ptr()), "cross_entropy")
    return loss, row_loss


def cast_bf16(x):
    out = torch.empty(x.shape, device=x.device, dtype=BF)
    N.check(N.lib().oasr_cast_f32_bf16(N.ptr(x), N.ptr(out), x.numel(), N.stream_ptr()), "cast")
    return out


def log_mel(pcm, finalize: bool = True):
    """pcm int16 or float32 [B, n] on the GPU -> float32 [B, 80, n // 160].  ``finalize=False``: returns (mel_raw, clip_max [B]) --
    whisper's last two lines (floor at the clip maximum - 8, (x + 4) / 4) left to the consumer (``loss_and_backward(mel_clip_max=...)``)."""
    N.require_gpu(pcm, "pcm")
    assert pcm.dim() == 2 and pcm.is_contiguous()
    B, n = pcm.shape
    if pcm.dtype == torch.int16:
        dt = 1
    elif pcm.dtype == torch.float32:
        dt = 0
    else:
        raise N.NativeError(f"log_mel: unsupported dtype {pcm.dtype}")
    mel = torch.empty(B, 80, n // 160, device=pcm.device, dtype=torch.float32)
    ws = torch.empty(N.lib().oasr_log_mel_workspace_bytes(B), device=pcm.device, dtype=torch.uint8)
    if not finalize:
        cm = torch.empty(B, device=pcm.device, dtype=torch.float32)
        N.check(N.lib().oasr_log_mel_raw(N.ptr(pcm), dt, B, n, N.ptr(mel), N.ptr(cm), N.ptr(ws), N.stream_ptr()), "oasr_log_mel_raw")
        return mel, cm
    N.check(N.lib().oasr_log_mel(N.ptr(pcm), dt, B, n, N.ptr(mel), N.ptr(ws), N.stream_ptr()), "oasr_log_mel")
    return mel


def pick_tokens(logits, mask=None, mask2=None, want_logprob=True):
    """logits fp32 [rows, V] (row stride free) -> (argmax ids int64 [rows], log_softmax at the argmax fp32 [rows] | None);
    ``mask`` / ``mask2``: additive fp32 [V] (0 / -inf)."""
    N.require_gpu(logits, "logits")
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    rows, V = logits.shape
    tok = torch.empty(rows, device=logits.device, dtype=torch.int64)
    lp = torch.empty(rows, device=logits.device, dtype=torch.float32) if want_logprob else None
    for m in (mask, mask2):
        assert m is None or (m.dtype == torch.float32 and m.numel() == V and m.is_contiguous())
    N.check(N.lib().oasr_pick_tokens(N.ptr(logits), logits.stride(0), V, rows, N.ptr(mask), N.ptr(mask2), N.ptr(tok), N.ptr(lp),
                                     N.stream_ptr()), "oasr_pick_tokens")
    return tok, lp


def pick_tokens_ts(logits, history, n_history, *, timestamp_begin, eot, no_timestamps, max_initial_index=None, mask=None, mask2=None):
    """``pick_tokens`` with whisper's ApplyTimestampRules evaluated on the device: ``history`` int64 [rows, >= n_history] holds the
    tokens sampled so far (after the sot sequence).  Returns (ids int64 [rows], log_softmax of the pick over the surviving columns)."""
    N.require_gpu(logits, "logits")
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    rows, V = logits.shape
    assert n_history == 0 or (history.dtype == torch.int64 and history.dim() == 2 and history.shape[0] == rows and history.stride(1) == 1
                              and history.shape[1] >= n_history and history.device == logits.device)
    tok = torch.empty(rows, device=logits.device, dtype=torch.int64)
    lp = torch.empty(rows, device=logits.device, dtype=torch.float32)
    for m in (mask, mask2):
        assert m is None or (m.dtype == torch.float32 and m.numel() == V and m.is_contiguous())
    N.check(N.lib().oasr_pick_tokens_ts(N.ptr(logits), logits.stride(0), V, rows, N.ptr(mask), N.ptr(mask2),
                                        N.ptr(history) if n_history else None, history.stride(0) if n_history else 0, int(n_history),
                                        int(timestamp_begin), int(eot), int(no_timestamps),
                                        -1 if max_initial_index is None else int(max_initial_index), N.ptr(tok), N.ptr(lp), N.stream_ptr()),
            "oasr_pick_tokens_ts")
    return tok, lp


def _ts_args(logits, history, n_history, mask, mask2):
    N.require_gpu(logits, "logits")
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    rows, V = logits.shape
    if n_history is not None and n_history > 0:
        assert (history.dtype == torch.int64 and history.dim() == 2 and history.shape[0] == rows and history.stride(1) == 1
                and history.shape[1] >= n_history and history.device == logits.device)
    for m in (mask, mask2):
        assert m is None or (m.dtype == torch.float32 and m.numel() == V and m.is_contiguous())
    nh = -1 if n_history is None else int(n_history)
    return rows, V, (N.ptr(history) if nh > 0 else None), (history.stride(0) if nh > 0 else 0), nh


def topk_tokens(logits, k, *, history=None, n_history=None, timestamp_begin=50363, eot=50256, no_timestamps=50362, max_initial_index=None,
                mask=None, mask2=None):
    """The k best (log_softmax value, token) pairs per row of the filtered distribution (suppress masks; ``n_history`` not None: whisper's
    ApplyTimestampRules from ``history``, as ``pick_tokens_ts``) -- BeamSearchDecoder.update's ``logprobs.topk(beam_size + 1)`` in one
    kernel.  Returns (logprob f32 [rows, k] descending, ids int64 [rows, k])."""
    rows, V, hp, hld, nh = _ts_args(logits, history, n_history, mask, mask2)
    tok = torch.empty(rows, k, device=logits.device, dtype=torch.int64)
    lp = torch.empty(rows, k, device=logits.device, dtype=torch.float32)
    N.check(N.lib().oasr_topk_tokens(N.ptr(logits), logits.stride(0), V, rows, N.ptr(mask), N.ptr(mask2), hp, hld, nh, int(timestamp_begin),
                                     int(eot), int(no_timestamps), -1 if max_initial_index is None else int(max_initial_index), int(k),
                                     N.ptr(tok), N.ptr(lp), N.stream_ptr()), "oasr_topk_tokens")
    return lp, tok


def sample_tokens(logits, temperature, uniforms, *, history=None, n_history=None, timestamp_begin=50363, eot=50256, no_timestamps=50362,
                  max_initial_index=None, mask=None, mask2=None):
    """One draw per row from softmax(filtered logits / temperature) by inverse CDF on ``uniforms`` (f32 [rows] in [0, 1)); returns
    (ids int64 [rows], log_softmax of the draw at temperature 1) -- GreedyDecoder.update at temperature > 0."""
    rows, V, hp, hld, nh = _ts_args(logits, history, n_history, mask, mask2)
    assert uniforms.dtype == torch.float32 and uniforms.numel() == rows and uniforms.device == logits.device
    tok = torch.empty(rows, device=logits.device, dtype=torch.int64)
    lp = torch.empty(rows, device=logits.device, dtype=torch.float32)
    N.check(N.lib().oasr_sample_tokens(N.ptr(logits), logits.stride(0), V, rows, N.ptr(mask), N.ptr(mask2), hp, hld, nh, int(timestamp_begin),
                                       int(eot), int(no_timestamps), -1 if max_initial_index is None else int(max_initial_index),
                                       float(temperature), N.ptr(uniforms), N.ptr(tok), N.ptr(lp), N.stream_ptr()), "oasr_sample_tokens")
    return tok, lp
