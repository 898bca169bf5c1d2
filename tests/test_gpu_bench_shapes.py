"""Parity at the LAUNCH SHAPES of the benchmarked step (bench.py default: OLMoASR-medium, micro-batch 128 -- BASELINE configs[2]'s
per-GPU share): the shapes the driver's number is produced at had only been compared at B <= 4.

 * one micro-step at B = 128 (240 GiB workspace, offsets past 2^31 elements, attention grids of B*H = 2048, LayerNorm over 192,000
   rows, CE over 57,344 rows) against the SUM of 128 single-clip micro-steps on the same weights -- each of which IS pinned to the CPU
   oracle at B = 1 (tests/test_gpu_parity_sizes.py): loss and per-tensor gradients; plain step and supervised-span step;
 * the same comparison on the fp32 validation engine at the largest batch its workspace allows;
 * attention forward / backward at B = 128, H = 16, T = 1500 and 448 x 1500 against an fp32 torch reference on sampled (b, h);
 * LayerNorm at 192,000 x 1024 and cross-entropy at 57,344 x 51,968 against chunked fp32 references.
Reference: scripts/training/train_timestamps.py:1440-1454 at configs[2]'s per-GPU batch."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
PAD = 51864


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _batch_vs_singles(variant, B, dtype, tol_loss, tol_total, tol_tensor, span_modes):
    from olmoasr_amd import ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd.synth import synth_samples
    net = OLMoASR(VARIANT_TO_DIMS[variant], device=DEV, seed=0, compute_dtype=dtype)
    try:
        pcm, ti, ty, tl = synth_samples(list(range(B)), DEV)  # the samples bench.py's rank 0 trains on
        mel = ops.log_mel(pcm)
        nv = (ty != PAD).sum(1)
        n_tot = int(nv.sum())
        SC = 1024.0
        # ---- reference: the sum of B single-clip micro-steps.  CE normalises by the valid targets of ITS micro-batch, so clip b
        # enters the batch loss / gradient with weight n_b / n_total (loss_scale carries it into the gradient)
        net.zero_grad()
        ref_loss = 0.0
        one = torch.zeros(1, device=DEV)
        for b in range(B):
            w = float(nv[b]) / n_tot
            net.loss_and_backward(mel[b:b + 1], ti[b:b + 1], ty[b:b + 1], tl[b:b + 1], loss_scale=SC * w, loss_out=one)
            ref_loss += w * float(one)
        torch.cuda.synchronize()
        net._workspace = None
        torch.cuda.empty_cache()
        ref = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        ref_flat = net.flat_grads.clone()
        assert torch.isfinite(ref_flat).all() and float(ref_flat.abs().max()) > 0
        for mode in span_modes:
            net.zero_grad()
            loss, _ = net.loss_and_backward(mel, ti, ty, tl, loss_scale=SC, span=(True if mode != "plain" else None), span_forward=mode == "span-forward")
            torch.cuda.synchronize()
            total = _rel(net.flat_grads, ref_flat)
            worst = max((_rel(p.grad, ref[n]), n) for n, p in net.named_parameters())
            print(f"   {variant} {dtype} B={B} [{mode}] vs sum of {B} single-clip steps: loss {float(loss):.6f} vs {ref_loss:.6f}, "
                  f"grads rel-L2 {total:.2e}, worst tensor {worst[0]:.2e} ({worst[1]})")
            assert abs(float(loss) - ref_loss) <= tol_loss * abs(ref_loss), (mode, float(loss), ref_loss)
            assert total <= tol_total and worst[0] <= tol_tensor, (mode, total, worst)
    finally:
        net._workspace = None
        del net
        torch.cuda.empty_cache()


def test_medium_b128_step_equals_sum_of_single_clip_steps_bf16():
    # Measured (profiles/r04_span_step.txt): loss equal to the last printed digit, gradients rel-L2 1.3-1.8e-6, worst tensor 7e-6 -- only the
    # order of the fp32 atomic accumulation differs; every per-row result is bit-identical whatever the batch.  Bounds 50x above that.
    _batch_vs_singles("medium", 128, "bfloat16", tol_loss=2e-6, tol_total=1e-4, tol_tensor=4e-4, span_modes=("plain", "span", "span-forward"))


def test_medium_batch_step_equals_sum_of_single_clip_steps_fp32():
    # the fp32 validation engine keeps 2x the bytes per activation and runs on plain VALU kernels: B = 8 keeps the test in seconds
    _batch_vs_singles("medium", 8, "float32", tol_loss=1e-6, tol_total=1e-5, tol_tensor=1e-4, span_modes=("plain", "span", "span-forward"))  # measured 1.6e-6 / 3.2e-6


def _ref_attention_bh(q, k, v, causal, kv_len_b):
    """fp32 softmax attention of ONE (b, h): q [Tq, 64], k / v [Tk, 64] -> o [Tq, 64], lse [Tq]."""
    s = q.float() @ k.float().t() / 8.0
    Tq, Tk = s.shape
    if causal:
        s = s.masked_fill(torch.ones(Tq, Tk, device=s.device, dtype=torch.bool).triu(1), float("-inf"))
    if kv_len_b is not None:
        s = s.masked_fill(torch.arange(Tk, device=s.device)[None, :] >= kv_len_b, float("-inf"))
    return torch.softmax(s, -1) @ v.float(), torch.logsumexp(s, -1)


@pytest.mark.parametrize("kind", ["encoder", "cross", "decoder-self"])
def test_attention_at_bench_shapes(kind):
    """B = 128, H = 16 (grid 2048 x query blocks): forward and backward vs the fp32 reference on sampled (b, h) slices."""
    from olmoasr_amd import ops
    B, H = 128, 16
    d = H * 64
    Tq, Tk = (1500, 1500) if kind == "encoder" else ((448, 1500) if kind == "cross" else (448, 448))
    causal = kind == "decoder-self"
    g = torch.Generator(device=DEV).manual_seed(5)

    def rn(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=DEV) * scale).to(BF)
    if Tq == Tk:
        qkv = rn(B, Tq, 3 * d)
        q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(3))
    else:
        qb, kvb = rn(B, Tq, d), rn(B, Tk, 2 * d)
        q = qb.unflatten(2, (H, 64))
        k, v = (kvb[:, :, i * d:(i + 1) * d].unflatten(2, (H, 64)) for i in range(2))
    kv_len = None
    if causal:
        kv_len = torch.randint(8, 221, (B,), generator=g, device=DEV, dtype=torch.int32)
        kv_len[-1] = 448
    o, lse, o_lo = ops.attention_fwd(q, k, v, kv_len, causal, want_o_lo=True)
    d_o = rn(B, Tq, d, scale=0.5)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, d_o, kv_len, causal, o_lo=o_lo)
    torch.cuda.synchronize()
    for b, h in [(0, 0), (127, 15), (64, 7), (33, 12), (126, 1)]:
        qr, kr, vr = (t[b, :, h].detach().float().requires_grad_(True) for t in (q, k, v))
        ro, rlse = _ref_attention_bh(qr, kr, vr, causal, int(kv_len[b]) if kv_len is not None else None)
        got_o = o.view(B, Tq, H, 64)[b, :, h].float()
        err = (got_o - ro).abs()
        assert float(err.max()) <= 2e-3 + 1e-2 * float(ro.abs().mean()) + 1e-2 * float(ro.abs().max()), (kind, b, h, "o", float(err.max()))
        assert float((lse[b, h] - rlse).abs().max()) <= 2e-3 + 1e-3 * float(rlse.abs().max()), (kind, b, h, "lse")
        ro.backward(d_o.view(B, Tq, H, 64)[b, :, h].float())
        for nm, got, rg in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
            e = (got[b, :, h].float() - rg).abs()
            bad = e > 5e-3 * float(rg.abs().max()) + 1e-3 + 2e-2 * rg.abs()
            assert not bool(bad.any()), (kind, b, h, nm, float(e.max()), float(rg.abs().max()))


def test_layernorm_at_bench_shape():
    from olmoasr_amd import ops
    rows, d = 192000, 1024
    g = torch.Generator(device=DEV).manual_seed(7)
    x = (torch.randn(rows, d, generator=g, device=DEV) * 2.0 + 0.3).to(BF)
    gamma = 1 + 0.1 * torch.randn(d, generator=g, device=DEV)
    beta = 0.1 * torch.randn(d, generator=g, device=DEV)
    dy = torch.randn(rows, d, generator=g, device=DEV).to(BF)
    dres = torch.randn(rows, d, generator=g, device=DEV).to(BF)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    dx, dg, db = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres)
    torch.cuda.synchronize()
    dg_ref = torch.zeros(d, device=DEV, dtype=torch.float64)
    db_ref = torch.zeros(d, device=DEV, dtype=torch.float64)
    for r0 in range(0, rows, 16384):
        r1 = min(rows, r0 + 16384)
        xf = x[r0:r1].float().requires_grad_(True)
        gf, bf_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        ref = F.layer_norm(xf, (d,), gf, bf_, 1e-5)
        e = (y[r0:r1].float() - ref).abs()
        assert not bool((e > 1e-2 * ref.abs() + 1e-2 * float(ref.abs().mean())).any()), ("ln fwd", r0, float(e.max()))
        assert float((mean[r0:r1] - xf.mean(-1)).abs().max()) < 1e-4
        ref.backward(dy[r0:r1].float())
        want = xf.grad.to(BF).float() + dres[r0:r1].float()
        e = (dx[r0:r1].float() - want).abs()
        assert not bool((e > 1e-2 * want.abs() + 1e-2 * float(want.abs().mean())).any()), ("ln dx", r0, float(e.max()))
        dg_ref += gf.grad.double()
        db_ref += bf_.grad.double()
    assert _rel(dg, dg_ref) < 1e-3 and _rel(db, db_ref) < 1e-3, (_rel(dg, dg_ref), _rel(db, db_ref))


def test_cross_entropy_at_bench_shape():
    """[57344, 51968] bf16 logits (12 GB): loss and in-place gradient vs F.cross_entropy chunk by chunk, 3 of 4 rows ignored as in the step."""
    from olmoasr_amd import ops
    rows, V, Vp = 128 * 448, 51865, 51968
    g = torch.Generator(device=DEV).manual_seed(9)
    logits = torch.empty(rows, Vp, device=DEV, dtype=BF)
    for r0 in range(0, rows, 8192):
        logits[r0:r0 + 8192] = (torch.randn(8192, Vp, generator=g, device=DEV) * 2.0).to(BF)
    targets = torch.randint(0, V - 1, (rows,), generator=g, device=DEV)
    pos = torch.arange(rows, device=DEV) % 448
    targets[pos >= 112] = PAD
    keep = logits.clone()
    n_valid = int((targets != PAD).sum())
    loss, row_loss = ops.cross_entropy_(logits, V, targets, PAD, gscale=1.0, write_grad=True)
    torch.cuda.synchronize()
    tot = 0.0
    for r0 in range(0, rows, 4096):
        r1 = r0 + 4096
        lf = keep[r0:r1, :V].float().requires_grad_(True)
        l = F.cross_entropy(lf, targets[r0:r1], ignore_index=PAD, reduction="sum")
        l.backward()
        tot += float(l)
        want = lf.grad / n_valid
        got = logits[r0:r1, :V].float()
        e = (got - want).abs()
        assert not bool((e > 1e-2 * want.abs() + 2e-3 * float(want.abs().max())).any()), ("dlogits", r0, float(e.max()))
        assert float(logits[r0:r1, V:].float().abs().max()) == 0.0  # pad columns
    assert abs(float(loss) - tot / n_valid) <= 1e-5 * abs(tot / n_valid), (float(loss), tot / n_valid)
