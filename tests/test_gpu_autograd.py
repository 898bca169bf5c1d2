"""OLMoASR.forward as a differentiable torch op: the reference's training lines (scripts/training/train_timestamps.py:1440-1454,
1509-1521) run UNCHANGED on the native model -- ``logits = model(mel, tokens, mask)``, ``F.cross_entropy(..., ignore_index=51864)``,
``scaler.scale(loss).backward()``, ``scaler.unscale_``, ``clip_grad_norm_``, ``scaler.step(optimizer)`` with torch's own AdamW --
and give the fused native step's gradients / updates."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
PAD = 51864


def _dims(mo_dims):
    from olmoasr_amd.config.model_dims import ModelDimensions
    return ModelDimensions(**{k: getattr(mo_dims, k) for k in ModelDimensions.__dataclass_fields__})


def _mask(tl):
    m = torch.zeros(tl.numel(), 448, 448)
    for b, n in enumerate(tl.tolist()):
        m[b, :, n:] = -float("inf")
    return m


def _ref_loss(net, c, scale=1.0, accum=1):
    logits = net(c["mel"].to(DEV), c["tokens"].to(DEV), _mask(c["text_len"]).to(DEV))
    assert logits.requires_grad and logits.dtype == torch.float32 and logits.shape == (2, 448, PAD + 1)
    loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), c["targets"].to(DEV).view(-1), ignore_index=PAD) / accum
    return loss, logits


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_autograd_backward_equals_the_fused_step(tiny_case, dtype):
    from olmoasr_amd.model import OLMoASR
    c = tiny_case
    net = OLMoASR(_dims(c["dims"]), device=DEV, seed=0, compute_dtype=dtype)
    net.load_state_dict(c["sd"])
    net.zero_grad()
    loss_f, _ = net.loss_and_backward(c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV), loss_scale=1024.0)
    fused = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad()
    loss, logits = _ref_loss(net, c)
    (loss * 1024.0).backward()
    assert abs(float(loss.detach()) - float(loss_f)) < 1e-4 * abs(float(loss_f)) + 1e-5
    # fp32: measured 3e-6.  bf16: the two paths meet at d(logits) -- torch's fp32 (softmax - onehot) * g rounded to bf16 by the bridge vs the
    # fused kernel's exp2(..) * (g / sum) rounded to bf16.  The bf16-quantised logits put whole groups of equal probabilities exactly
    # on bf16 rounding ties (e.g. 0.013275150 between 0.0132446 and 0.0133057), which the two formulas break differently for ~450 of
    # 5.4 M elements; one such ulp re-rounds the activations of every layer below: 2e-3 median / 6e-3 worst per-tensor rel-L2 at tiny,
    # the size of any other bf16 reordering (e.g. the two attention-backward kernel families, same test case: 6e-3).
    tol = 2e-5 if dtype == "float32" else 9e-3  # bf16: 1.5 x the measured worst tensor (6e-3)
    worst = 0.0
    for n, p in net.named_parameters():
        assert p.grad is not None
        rel = float((p.grad - fused[n]).norm() / (fused[n].norm() + 1e-20))
        worst = max(worst, rel)
        assert rel < tol, (n, rel)
    print(f"[{dtype}] autograd vs fused gradients: worst per-tensor rel-L2 {worst:.3g}")
    # gradient accumulation: a second forward/backward pair adds
    loss2, _ = _ref_loss(net, c)
    (loss2 * 1024.0).backward()
    for n, p in net.named_parameters():
        assert float((p.grad - 2 * fused[n]).norm() / (2 * fused[n].norm() + 1e-20)) < 2 * tol, n


def test_reference_training_lines_with_torch_optimizer_and_gradscaler(tiny_case):
    """train_timestamps.py:1440-1454 + 1509-1521 verbatim against the native fused step (optim_step): same losses, same weights."""
    from olmoasr_amd.model import OLMoASR
    c = tiny_case
    hp = dict(lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    ref = OLMoASR(_dims(c["dims"]), device=DEV, seed=0, compute_dtype="float32")
    nat = OLMoASR(_dims(c["dims"]), device=DEV, seed=0, compute_dtype="float32")
    ref.load_state_dict(c["sd"])
    nat.load_state_dict(c["sd"])
    optimizer = torch.optim.AdamW(ref.parameters(), **hp)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    nat.init_optimizer_state()
    losses = []
    for step in range(3):
        optimizer.zero_grad()  # set_to_none=True: the next forward re-attaches (zeroed) arena views
        loss, _ = _ref_loss(ref, c)
        scaler.scale(loss).backward()
        scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        scaler.step(optimizer)
        scaler.update()
        nat.zero_grad()
        ln, _ = nat.loss_and_backward(c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV), loss_scale=65536.0)
        nat.optim_step(step=step + 1, inv_loss_scale=1.0 / 65536.0, max_grad_norm=1.0, **hp)
        losses.append((float(loss.detach()), float(ln)))
        assert abs(losses[-1][0] - losses[-1][1]) < 2e-4, losses
    assert losses[-1][0] < losses[0][0]  # torch's optimizer really moved the engine's weights (compute copies refreshed)
    for (n, a), (_, b) in zip(ref.named_parameters(), nat.named_parameters()):
        assert torch.allclose(a, b, atol=2e-5, rtol=0), (n, float((a - b).abs().max()))


def test_one_forward_per_backward_is_enforced_and_eval_mode_has_no_graph(tiny_case):
    from olmoasr_amd.model import OLMoASR
    c = tiny_case
    net = OLMoASR(_dims(c["dims"]), device=DEV, seed=0)
    loss1, _ = _ref_loss(net, c)
    loss2, _ = _ref_loss(net, c)       # overwrites the saved activations of the first forward
    loss2.backward()
    with pytest.raises(RuntimeError, match="activations of this forward are gone"):
        loss1.backward()
    net.eval()
    assert not net(c["mel"].to(DEV), c["tokens"].to(DEV)).requires_grad
    net.train()
    with torch.no_grad():
        assert not net(c["mel"].to(DEV), c["tokens"].to(DEV)).requires_grad


def test_ddp_wrapper_runs_the_reference_loop_lines(tiny_case, tmp_path):
    """``model = DDP(model, device_ids=[local_rank])`` (train_timestamps.py:2330) with olmoasr_amd.ddp.DistributedDataParallel: one rank
    over RCCL (the collectives really run), the reference's lines, state_dict keys prefixed ``module.``; results equal the unwrapped
    model's (mean over one rank), with and without ``no_sync`` under gradient accumulation."""
    import torch.distributed as dist
    from olmoasr_amd import ddp
    from olmoasr_amd.model import OLMoASR
    c = tiny_case
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        plain = OLMoASR(_dims(c["dims"]), device=DEV, seed=0, compute_dtype="float32")
        plain.load_state_dict(c["sd"])
        base = OLMoASR(_dims(c["dims"]), device=DEV, seed=0, compute_dtype="float32")
        base.load_state_dict(c["sd"])
        model = ddp.DistributedDataParallel(base, device_ids=[0])
        assert all(k.startswith("module.") for k in model.state_dict())
        plain.zero_grad()
        for _ in range(2):
            loss, _ = _ref_loss(plain, c, accum=2)
            loss.backward()
        want = {n: p.grad.clone() for n, p in plain.named_parameters()}
        for use_no_sync in (False, True):
            model.zero_grad()
            if use_no_sync:
                with model.no_sync():
                    loss, _ = _ref_loss(model, c, accum=2)
                    loss.backward()
            else:
                loss, _ = _ref_loss(model, c, accum=2)
                loss.backward()
            loss, _ = _ref_loss(model, c, accum=2)
            loss.backward()
            torch.cuda.synchronize()
            for n, p in base.named_parameters():
                assert torch.allclose(p.grad, want[n], atol=1e-6, rtol=1e-5), (use_no_sync, n)
    finally:
        if created:
            dist.destroy_process_group()


def test_autograd_path_with_a_short_decoder_context(tiny_case):
    """model(mel, tokens[:, :S]) for S < 448 (the reference's forward takes any S <= n_text_ctx, model.py:716-732): same loss and
    gradients as the fused step over the same span (``text_ctx=S``), which in turn equal the full padded context (tests elsewhere)."""
    from olmoasr_amd.model import OLMoASR
    c = tiny_case
    S = 224
    assert int(c["text_len"].max()) <= S
    net = OLMoASR(_dims(c["dims"]), device=DEV, seed=0, compute_dtype="float32")
    net.load_state_dict(c["sd"])
    net.zero_grad()
    lf, _ = net.loss_and_backward(c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV), text_ctx=S)
    fused = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad()
    logits = net(c["mel"].to(DEV), c["tokens"].to(DEV)[:, :S], c["text_len"].to(DEV))
    assert logits.shape == (2, S, PAD + 1)
    loss = F.cross_entropy(logits.view(-1, PAD + 1), c["targets"].to(DEV)[:, :S].reshape(-1), ignore_index=PAD)
    loss.backward()
    assert abs(float(loss.detach()) - float(lf)) < 1e-5
    for n, p in net.named_parameters():
        assert float((p.grad - fused[n]).norm() / (fused[n].norm() + 1e-20)) < 2e-5, n
