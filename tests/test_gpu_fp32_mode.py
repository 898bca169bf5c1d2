"""fp32 validation mode (compute_dtype="float32" == the reference's --precision float32, train_timestamps.py:2128,
2220-2224): the same engine schedule on fp32 kernels (olmoasr_amd/csrc/fp32ref.hip), held to BASELINE.json's
"logits within 1e-3" against the fp32 CPU oracle, which tests/test_oracle_model.py pins to the unmodified reference.
Covers forward, loss, every gradient tensor, the optimizer step, gradient accumulation, the KV-cached decode step
(against the ORACLE's full-prefix decoder, not against the engine itself) and greedy token ids."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dims(mo_dims):
    from olmoasr_amd.config.model_dims import ModelDimensions
    return ModelDimensions(**{k: getattr(mo_dims, k) for k in ModelDimensions.__dataclass_fields__})


@pytest.fixture(scope="module")
def net32(tiny_case):
    from olmoasr_amd.model import OLMoASR
    net = OLMoASR(_dims(tiny_case["dims"]), device=DEV, seed=0, compute_dtype="float32")
    net.load_state_dict(tiny_case["sd"], strict=True)
    return net


@pytest.fixture(scope="module")
def oracle(tiny_case):
    from oracle import model_oracle as mo
    c = tiny_case
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    loss, grads, logits = mo.loss_and_grads(c["sd"], c["dims"], c["mel"], c["tokens"], c["targets"], c["text_len"])
    return dict(loss=float(loss), grads=grads, logits=logits)


def test_fp32_forward_logits_within_1e3(net32, oracle, tiny_case, golden_dir):
    from oracle import model_oracle as mo
    c = tiny_case
    pm = mo.build_padding_mask(c["text_len"])
    logits = net32(c["mel"].to(DEV), c["tokens"].to(DEV), pm.to(DEV)).detach().cpu()  # (training mode: the logits carry a grad_fn, as the reference's do)
    assert logits.shape == (2, 448, 51865) and logits.dtype == torch.float32
    valid = torch.arange(448)[None, :] < c["text_len"][:, None].long()
    err = (logits - oracle["logits"]).abs()
    print(f"fp32 mode: logits max |delta| on valid rows {float(err[valid].max()):.2e}, on all rows {float(err.max()):.2e} (scale {float(oracle['logits'].abs().max()):.1f})")
    assert float(err.max()) <= 1e-3  # padded query rows included: they are computed like any other row
    # and against the fixtures generated from the UNMODIFIED reference modules (oracle/gen_golden.py)
    g = np.load(os.path.join(golden_dir, "ref_tiny_b2.npz"))
    POS = [0, 1, 2, 7, 50, 100, 219, 447]
    assert np.abs(logits[:, POS, :512].numpy() - g["fp32_head"]).max() <= 1e-3
    assert (logits.argmax(-1)[valid] == oracle["logits"].argmax(-1)[valid]).float().mean() > 0.999
    # xa and logits(tokens, xa) agree with forward()
    xa = net32.embed_audio(c["mel"].to(DEV))
    assert xa.dtype == torch.float32
    xa_ref = mo.encoder_forward(c["sd"], c["dims"], c["mel"])
    assert float((xa.cpu() - xa_ref).abs().max()) <= 1e-3
    l2 = net32.logits(c["tokens"].to(DEV), xa, c["text_len"].to(DEV))
    assert torch.equal(l2.cpu(), logits)


def test_fp32_loss_grads_and_step_within_1e3(net32, oracle, tiny_case):
    from oracle import model_oracle as mo
    c = tiny_case
    net = net32
    args = (c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV))
    net.zero_grad()
    loss, _ = net.loss_and_backward(*args)
    torch.cuda.synchronize()
    assert abs(float(loss) - oracle["loss"]) < 1e-4
    rows = []
    for name, p in net.named_parameters():
        gr = oracle["grads"][name]
        gn = p.grad.detach().cpu()
        rows.append((float((gn - gr).norm() / (gr.norm() + 1e-20)), float((gn - gr).abs().max()), name))
    rows.sort(reverse=True)
    print("fp32 mode, worst gradient rel-L2 / max abs:", [(f"{r:.1e}", f"{a:.1e}", n) for r, a, n in rows[:4]])
    assert rows[0][0] <= 1e-3, rows[0]
    # two half-weight micro-steps accumulate to the same gradients (grad accumulation, train_timestamps.py:1448-1454)
    g1 = net.flat_grads.clone()
    net.zero_grad()
    net.loss_and_backward(*args, loss_scale=1024.0, accumulation_steps=2)
    net.loss_and_backward(*args, loss_scale=1024.0, accumulation_steps=2)
    assert float((net.flat_grads / 1024.0 - g1).norm() / g1.norm()) < 1e-5
    # unscale + clip + AdamW vs the oracle's step on the ORACLE's gradients
    names = [n for n, _ in net.named_parameters()]
    params = {n: c["sd"][n].clone() for n in names}
    grads = {n: oracle["grads"][n].clone() for n in names}
    total, coef = mo.clip_coef(grads, 1.0)
    for n in names:
        grads[n].mul_(coef)
    m = {n: torch.zeros_like(params[n]) for n in names}
    v = {n: torch.zeros_like(params[n]) for n in names}
    mo.adamw_step(params, grads, m, v, step=1, lr=1e-3)
    net.init_optimizer_state()
    for t in net._opt_state:
        t.zero_()
    stats = net.optim_step(step=1, lr=1e-3, inv_loss_scale=1.0 / 1024.0)
    torch.cuda.synchronize()
    assert float(stats[1]) == 0.0 and abs(float(stats[0].sqrt()) / 1024.0 - float(total)) / float(total) < 1e-4
    bad = tot = 0
    for n, p in net.named_parameters():
        d = (p.detach().cpu() - params[n]).abs()
        bad += int((d > 1e-5).sum())  # first AdamW step ~ lr * sign(g): only sign flips of ~zero gradients can differ
        tot += d.numel()
    print(f"post-AdamW: {bad} of {tot} weights differ by more than 1e-5")
    assert bad <= 1e-4 * tot
    net.load_state_dict(c["sd"])


def test_fp32_kv_cached_decode_vs_oracle(tiny_case):
    """install_kv_cache_hooks semantics (olmoasr/model.py:925-964): one-token decoder steps over the engine-owned KV
    cache against the ORACLE's cache-less full-prefix decoder (notebooks/ow_decoding.py:42-72 style), per-step logits
    within 1e-3, and the greedy ids of the two loops identical."""
    from olmoasr_amd.decoding import DecodingOptions, decode
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    sd = mo.init_state_dict(dims, seed=11, train_vocab_rows=False)
    net = OLMoASR(_dims(dims), device=DEV, seed=0, inference=True, compute_dtype="float32")
    net.load_state_dict(sd)
    mel_cpu = tiny_case["mel"]
    mel = mel_cpu.to(DEV)
    xa = net.embed_audio(mel)
    xa_ref = mo.encoder_forward(sd, dims, mel_cpu)
    toks = tiny_case["tokens"][:, :9]
    ref = mo.decoder_forward(sd, dims, toks, xa_ref)  # [B, 9, V]
    st = net.kv_cache_begin(xa)
    for p in range(9):
        step = net.kv_cache_step(st, toks[:, p].to(DEV)).cpu()
        err = float((step - ref[:, p]).abs().max())
        assert err <= 1e-3, (p, err)
    # greedy loop: cached native vs cache-less oracle, every position (fp32 both sides: no margin gating needed unless a tie)
    want = mo.greedy_decode(sd, dims, mel_cpu, [50257, 50362], max_new=8)
    res = decode(net, mel, DecodingOptions(sample_len=8, use_kv_cache=True, without_timestamps=True, suppress_tokens=None, suppress_blank=False))
    for b, r in enumerate(res):
        got = r.tokens
        exp = [t for t in want[b, 2:].tolist()]
        if 50256 in exp:
            exp = exp[:exp.index(50256)]
        assert got == exp, (b, got, exp)
