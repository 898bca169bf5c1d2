"""Oracle parity of the training micro-step at BASELINE.json's model sizes (base = configs[1], small = configs[4]'s
dims, medium = configs[2], the benchmarked configuration) and of the GEMM kernels at the launch shapes bench.py runs.

The CPU oracle (oracle/model_oracle.py, pinned to the unmodified reference by tests/test_oracle_model.py) evaluates
the same seeded batch in fp32 and, as the error envelope of a bf16 evaluation, under the reference's autocast
rounding points ("bf16 mirror").  Checked per size, against the fp32 oracle (reference: olmoasr/model.py:856-887,
scripts/training/train_timestamps.py:1440-1454,1509-1512):
  * loss |delta| < 2e-2
  * logits: max |delta| <= 1.5 x the mirror's own max |delta| (what bf16 costs the reference itself), mean <= mirror's mean x 1.5
  * per-tensor gradient rel-L2 <= max(2 x mirror, 3 %), cosine > 1 - max(2 mirror^2, 1e-3), norm ratio within max(2 x mirror, 2 %), global <= 2 %
  * the fused clip + AdamW update of every weight against the oracle's AdamW on the ORACLE's gradients
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

# medium (the benchmarked configuration) is compared DIRECTLY with the CPU oracle at B = 4 -- a batch, not a single clip -- so the chain
# "medium B = 128 == sum of 128 single-clip HIP steps" (test_gpu_bench_shapes.py) + "single clip == oracle" has a batched link of its own
SIZES = [("base", 2), ("small", 2), ("medium", 4), ("large", 1)]  # large = BASELINE configs[3] (large-v2 shares its dims)


def bf16_ulp_report(native, mirror, valid, tag):
    """native (bf16 engine) vs the oracle's autocast MIRROR (same rounding points, CPU accumulation order), in bf16 ulps of the logit
    scale: ulp = 2^(floor(log2 max|logit|) - 7), the spacing of bf16 at the largest logit.  Returns (fraction within 2 ulp, max in ulp)."""
    import math
    scale = float(mirror[valid].abs().max())
    ulp = 2.0 ** (math.floor(math.log2(scale)) - 7)
    d = (native - mirror).abs()[valid] / ulp
    hist = [float((d <= k).float().mean()) for k in (0.5, 1, 2, 4)]
    print(f"   [{tag}] bf16 engine vs autocast mirror, valid logits: scale {scale:.2f} -> ulp {ulp:.5f}; within 0.5/1/2/4 ulp: "
          f"{hist[0]:.5f} {hist[1]:.5f} {hist[2]:.5f} {hist[3]:.5f}; max {float(d.max()):.2f} ulp; mean {float(d.mean()):.3f} ulp")
    return hist[2], float(d.max())


def _dims(mo_dims):
    from olmoasr_amd.config.model_dims import ModelDimensions
    return ModelDimensions(**{k: getattr(mo_dims, k) for k in ModelDimensions.__dataclass_fields__})


@pytest.fixture(scope="module", params=SIZES, ids=[s for s, _ in SIZES])
def case(request):
    import numpy as np
    from oracle import mel_oracle as me
    from oracle import model_oracle as mo
    variant, B = request.param
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dims = mo.VARIANTS[variant]
    sd = mo.init_state_dict(dims, seed=0)
    pcm, ti, ty, tl = mo.synthetic_batch(list(range(60, 60 + B)))
    mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
    loss, grads, logits = mo.loss_and_grads(sd, dims, mel, ti, ty, tl)
    loss_b, grads_b, logits_b = mo.loss_and_grads(sd, dims, mel, ti, ty, tl, autocast_bf16=True)
    return dict(variant=variant, B=B, dims=dims, sd=sd, mel=mel, ti=ti, ty=ty, tl=tl, loss=float(loss), grads=grads, logits=logits,
                loss_b=float(loss_b), grads_b=grads_b, logits_b=logits_b.float())


def test_step_vs_oracle_at_size(case):
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    c = case
    net = OLMoASR(_dims(c["dims"]), device=DEV, seed=0)
    res = net.load_state_dict(c["sd"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    net.zero_grad()
    loss, logits = net.loss_and_backward(c["mel"].to(DEV), c["ti"].to(DEV), c["ty"].to(DEV), c["tl"].to(DEV), return_logits=True)
    torch.cuda.synchronize()
    # ---- loss ----
    print(f"[{c['variant']} B={c['B']}] loss native {float(loss):.5f} oracle fp32 {c['loss']:.5f} oracle bf16-mirror {c['loss_b']:.5f}")
    assert abs(float(loss) - c["loss"]) < 2e-2
    # ---- logits on the positions the loss sees ----
    valid = torch.arange(448)[None, :] < c["tl"][:, None].long()
    lg = logits.cpu()
    err = (lg - c["logits"]).abs()[valid]
    env = (c["logits_b"] - c["logits"]).abs()[valid]
    print(f"   logits vs fp32 oracle: max {float(err.max()):.4f} mean {float(err.mean()):.5f} | mirror envelope max {float(env.max()):.4f} "
          f"mean {float(env.mean()):.5f} | scale {float(c['logits'].abs().max()):.2f}")
    assert float(err.max()) <= 1.5 * float(env.max())
    assert float(err.mean()) <= 1.5 * float(env.mean())
    top2 = c["logits"].topk(2, -1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 2 * float(env.max())) & valid
    assert (lg.argmax(-1)[safe] == c["logits"].argmax(-1)[safe]).all()
    # bf16 against bf16: the engine vs the oracle's autocast mirror differ by accumulation order and the rounding ties it flips -- a
    # bound a one-ulp-per-layer bug cannot hide in (the fp32 envelope above is ~3 ulp wide): >= 99.9 % of the valid logits within 2 ulp
    # of the logit scale, none beyond 4
    within2, worst_ulp = bf16_ulp_report(lg, c["logits_b"], valid, f"{c['variant']} B={c['B']}")
    assert within2 >= 0.999 and worst_ulp <= 4.0, (within2, worst_ulp)
    # ---- gradients ----
    rows = []
    num = den = 0.0
    for name, p in net.named_parameters():
        gn, gr, gb = p.grad.detach().cpu(), c["grads"][name], c["grads_b"][name].float()
        rn = float(gr.norm()) + 1e-20
        rel = float((gn - gr).norm()) / rn
        envg = float((gb - gr).norm()) / rn
        cos = float((gn * gr).sum() / (gn.norm() * gr.norm() + 1e-20))
        ratio = float(gn.norm()) / rn
        rows.append((rel, envg, cos, ratio, name))
        num += float((gn - gr).double().pow(2).sum())
        den += float(gr.double().pow(2).sum())
    rows.sort(reverse=True)
    print("   worst per-tensor grad rel-L2 (native, mirror, cosine, norm ratio):")
    for rel, envg, cos, ratio, name in rows[:6]:
        print(f"      {rel:.4f} {envg:.4f} {cos:.5f} {ratio:.4f} {name}")
    glob = (num / den) ** 0.5
    print(f"   global grad rel-L2 {glob:.4f}")
    for rel, envg, cos, ratio, name in rows:
        assert rel <= max(2.0 * envg, 0.03), (name, rel, envg)
        assert cos > 1.0 - max(2.0 * envg * envg, 1e-3), (name, cos, envg)  # cos ~ 1 - rel^2 / 2
        assert abs(ratio - 1.0) <= max(2.0 * envg, 0.02), (name, ratio, envg)
    assert glob <= 0.02
    # ---- fused unscale + clip + AdamW: the ORACLE's gradients are loaded into the arena, so both sides step from identical
    # inputs and weights, exp_avg and exp_avg_sq must agree to fp32 rounding -- two steps (bias corrections, moment decay) ----
    lr = 1e-3
    names = [n for n, _ in net.named_parameters()]
    params = {n: c["sd"][n].clone() for n in names}
    m = {n: torch.zeros_like(params[n]) for n in names}
    v = {n: torch.zeros_like(params[n]) for n in names}
    net.load_state_dict(c["sd"])
    m_dev, v_dev = net.init_optimizer_state()
    m_dev.zero_()
    v_dev.zero_()
    slices = {name: (off, numel, shape) for name, off, numel, shape in net._param_slices()}
    for step in (1, 2):
        for n, p in net.named_parameters():
            p.grad.copy_(c["grads"][n].to(DEV))
        grads = {n: c["grads"][n].clone() for n in names}
        total, coef = mo.clip_coef(grads, 1.0)
        for n in names:
            grads[n].mul_(coef)
        mo.adamw_step(params, grads, m, v, step=step, lr=lr)
        stats = net.optim_step(step=step, lr=lr)
        torch.cuda.synchronize()
        assert float(stats[1]) == 0.0
        assert abs(float(stats[0].sqrt()) - float(total)) / float(total) < 1e-4
        wp = wm = wv = 0.0
        for n, p in net.named_parameters():
            off, numel, shape = slices[n]
            wp = max(wp, float((p.detach().cpu() - params[n]).abs().max()))
            mm, vv = m_dev[off:off + numel].view(shape).cpu(), v_dev[off:off + numel].view(shape).cpu()
            wm = max(wm, float((mm - m[n]).abs().max() / (m[n].abs().max() + 1e-30)))
            wv = max(wv, float((vv - v[n]).abs().max() / (v[n].abs().max() + 1e-30)))
        print(f"   AdamW step {step} on the oracle's gradients: max |dw| {wp:.2e}, exp_avg rel {wm:.2e}, exp_avg_sq rel {wv:.2e}")
        assert wp <= 4e-6 and wm <= 1e-5 and wv <= 1e-5, (step, wp, wm, wv)
    del net
    torch.cuda.empty_cache()


# ---- GEMM kernels at the launch shapes of the benchmarked step (medium, micro-batch 128) ---------------------------
def _rnd(shape, scale, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(BF)


def _check_rows(out, ref_fn, M, rtol, atol_scale, name, chunk=16384):
    """Compare `out` [M, N] with ref_fn(r0, r1) -> fp32 [r1-r0, N], chunk by chunk (keeps the fp32 reference small)."""
    worst = 0.0
    for r0 in range(0, M, chunk):
        r1 = min(M, r0 + chunk)
        ref = ref_fn(r0, r1)
        mag = None
        if isinstance(ref, tuple):  # (reference, magnitude of the largest bf16-rounded intermediate): one rounding is 2^-8 of THAT
            ref, mag = ref
        got = out[r0:r1].float()
        atol = atol_scale * float(ref.abs().mean()) + 1e-6
        err = (got - ref).abs()
        bad = err > atol + rtol * (ref.abs() if mag is None else torch.maximum(ref.abs(), mag))
        assert not bool(bad.any()), f"{name}: rows [{r0},{r1}): {int(bad.sum())} off, max err {float(err.max()):.4g}"
        worst = max(worst, float(err.max()))
    return worst


ENC_M, DEC_M = 128 * 1500, 128 * 448
FWD_SHAPES = [(ENC_M, 3072, 1024, "qkv"), (ENC_M, 1024, 1024, "attn.out+resid"), (ENC_M, 4096, 1024, "mlp.0+gelu"),
              (ENC_M, 1024, 4096, "mlp.2+resid"), (DEC_M, 51968, 1024, "logits")]


@pytest.mark.parametrize("M,N,K,kind", FWD_SHAPES, ids=[s[3] for s in FWD_SHAPES])
def test_gemm_forward_at_bench_shapes(M, N, K, kind):
    from olmoasr_amd import ops
    import torch.nn.functional as F
    A, W = _rnd((M, K), 1.0, 1), _rnd((N, K), K ** -0.5, 2)
    Wf = W.float()
    bias = (torch.randn(N, generator=torch.Generator().manual_seed(3)) * 0.1).to(DEV) if kind != "logits" else None
    out = torch.empty(M, N, device=DEV, dtype=BF)
    if kind == "mlp.0+gelu":
        pre = torch.empty_like(out)
        ops.gemm(A, W, M, N, K, bias=bias, act=1, out=out, out_pre=pre)
        _check_rows(pre, lambda a, b: A[a:b].float() @ Wf.t() + bias, M, 1e-2, 1e-2, kind + " pre")
        _check_rows(out, lambda a, b: F.gelu(pre[a:b].float()), M, 1e-2, 1e-2, kind + " gelu(pre)")
    elif "resid" in kind:
        resid = _rnd((M, N), 1.0, 4)
        ops.gemm(A, W, M, N, K, bias=bias, resid=resid, out=out)
        def ref(a, b):
            pre = (A[a:b].float() @ Wf.t() + bias).to(BF).float()  # the autocast rounding point of the Linear's output
            r = resid[a:b].float()
            return pre + r, torch.maximum(pre.abs(), r.abs())       # a sum that cancels still carries its operands' rounding
        _check_rows(out, ref, M, 1e-2, 1e-2, kind)
    else:
        ops.gemm(A, W, M, N, K, bias=bias, out=out)
        _check_rows(out, lambda a, b: A[a:b].float() @ Wf.t() + (bias if bias is not None else 0.0), M, 1e-2, 1e-2, kind)


@pytest.mark.parametrize("M,N,K,dgelu", [(ENC_M, 1024, 4096, False), (ENC_M, 4096, 1024, True), (ENC_M, 1024, 3072, False),
                                          (DEC_M, 1024, 51968, False)], ids=["d(mlp.0)", "d(mlp.2)*gelu'", "d(qkv)", "d(logits)"])
def test_gemm_dgrad_at_bench_shapes(M, N, K, dgelu):
    """dx[M,N] = dy[M,K] . W[K,N]  (NN layout: the weight is read in its forward layout and transposed on the way in),
    optional GELU'(u) epilogue and the fused bias-gradient column sums."""
    from olmoasr_amd import ops
    dy, W = _rnd((M, K), 0.05, 5), _rnd((K, N), K ** -0.5, 6)
    Wf = W.float()
    out = torch.empty(M, N, device=DEV, dtype=BF)
    if dgelu:  # the engine's form: u holds GELU'(pre) saved by the act == 2 forward (values in [-0.13, 1.13])
        u = (_rnd((M, N), 1.0, 7).float().sigmoid() * 1.26 - 0.13).to(BF)
        cs = torch.zeros(N, device=DEV)
        ops.gemm(dy, W, M, N, K, tb=True, dgelu_u=u, dgelu_deriv=True, out=out, colsum=cs)
        _check_rows(out, lambda a, b: (dy[a:b].float() @ Wf).to(BF).float() * u[a:b].float(), M, 1e-2, 1e-2, "dgrad*gelu'")
        want = out.float().sum(0)
        assert float((cs - want).abs().max()) <= 1e-3 * float(want.abs().max()) + 1e-3
    else:
        ops.gemm(dy, W, M, N, K, tb=True, out=out)
        _check_rows(out, lambda a, b: dy[a:b].float() @ Wf, M, 1e-2, 1e-2, "dgrad")


@pytest.mark.parametrize("Mtok,N,K", [(ENC_M, 4096, 1024), (ENC_M, 1024, 4096), (ENC_M, 3072, 1024), (DEC_M, 51864, 1024)],
                         ids=["dW mlp.0", "dW mlp.2", "dW qkv", "dE logits"])
def test_gemm_wgrad_at_bench_shapes(Mtok, N, K):
    """dW[N,K] += dY[Mtok,N]^T . X[Mtok,K] (TN layout, split-K fp32 atomics) with the engine's own split heuristic range."""
    from olmoasr_amd import ops
    dY, X = _rnd((Mtok, N), 0.02, 8), _rnd((Mtok, K), 1.0, 9)
    for split in (1, 8):
        dW = torch.zeros(N, K, device=DEV)
        ops.gemm(dY, X, N, K, Mtok, ta=True, tb=True, out_f32=dW, atomic=True, split_k=split)
        ref = torch.zeros(N, K, device=DEV, dtype=torch.float64)
        for r0 in range(0, Mtok, 32768):
            ref += (dY[r0:r0 + 32768].float().t() @ X[r0:r0 + 32768].float()).double()
        ref = ref.float()
        err = (dW - ref).abs()
        tol = 1e-3 * ref.abs() + 2e-3 * float(ref.abs().mean())
        assert not bool((err > tol).any()), f"wgrad split {split}: max err {float(err.max()):.4g} (ref scale {float(ref.abs().max()):.3g})"


# ---- fp32 validation mode at the same sizes: BASELINE.json's "logits within 1e-3" criterion ---------------------------
def test_fp32_mode_vs_oracle_at_size(case):
    """compute_dtype="float32" (the reference's --precision float32, train_timestamps.py:2128) runs the SAME engine
    schedule on fp32 kernels: logits <= 1e-3 abs, loss <= 1e-4, every gradient tensor <= 1e-3 rel-L2 against the fp32
    CPU oracle -- at base, small and the benchmarked medium dims."""
    from olmoasr_amd.model import OLMoASR
    c = case
    net = OLMoASR(_dims(c["dims"]), device=DEV, seed=0, compute_dtype="float32")
    net.load_state_dict(c["sd"], strict=True)
    net.zero_grad()
    loss, logits = net.loss_and_backward(c["mel"].to(DEV), c["ti"].to(DEV), c["ty"].to(DEV), c["tl"].to(DEV), return_logits=True)
    torch.cuda.synchronize()
    valid = torch.arange(448)[None, :] < c["tl"][:, None].long()
    err = (logits.cpu() - c["logits"]).abs()[valid]
    print(f"[{c['variant']} fp32 mode] loss {float(loss):.6f} vs {c['loss']:.6f}; logits max |delta| {float(err.max()):.2e} mean {float(err.mean()):.2e}")
    assert abs(float(loss) - c["loss"]) < 1e-4
    assert float(err.max()) <= 1e-3
    worst = (0.0, "")
    for name, p in net.named_parameters():
        gr = c["grads"][name]
        rel = float((p.grad.detach().cpu() - gr).norm() / (gr.norm() + 1e-20))
        worst = max(worst, (rel, name))
    print(f"   worst gradient rel-L2 {worst[0]:.2e} ({worst[1]})")
    assert worst[0] <= 1e-3, worst
    del net
    torch.cuda.empty_cache()
