"""Pins oracle/model_oracle.py against fixtures generated from the UNMODIFIED reference model
(tests/golden/ref_tiny_b2.npz, oracle/gen_golden.py) and, when /root/reference is mounted (build
container only), against the reference live."""
import os

import numpy as np
import pytest
import torch

from oracle import model_oracle as mo
from oracle import ref_import

POS = [0, 1, 2, 7, 50, 100, 219, 447]


@pytest.fixture(scope="module")
def ora(tiny_case):
    c = tiny_case
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    loss, grads, logits = mo.loss_and_grads(c["sd"], c["dims"], c["mel"], c["tokens"], c["targets"], c["text_len"])
    return dict(loss=loss, grads=grads, logits=logits)


def test_fp32_logits_loss_vs_golden(ora, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_tiny_b2.npz"))
    lg = ora["logits"]
    s = lg[:, POS, :]
    assert np.abs(s[..., :512].numpy() - g["fp32_head"]).max() < 5e-5
    assert np.abs(s[..., -64:].numpy() - g["fp32_tail"]).max() < 5e-5
    assert np.abs(torch.logsumexp(lg, -1).numpy() - g["fp32_lse"]).max() < 5e-5
    # argmax equality wherever the reference's top-2 margin exceeds the fp32 noise
    top2 = lg.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]).numpy() > 1e-3
    assert (lg.argmax(-1).numpy()[safe] == g["fp32_argmax"][safe]).all() and safe.mean() > 0.95
    assert abs(float(ora["loss"]) - float(g["fp32_loss"])) < 1e-5


def test_fp32_grads_vs_golden(ora, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_tiny_b2.npz"))
    names = [str(n) for n in g["param_names"]]
    assert set(names) == set(ora["grads"].keys())
    gn = np.array([ora["grads"][n].double().norm().item() for n in names])
    assert np.abs(gn - g["fp32_grad_norm"]).max() / g["fp32_grad_norm"].max() < 1e-5
    rel = np.abs(gn - g["fp32_grad_norm"]) / (g["fp32_grad_norm"] + 1e-12)
    assert rel.max() < 1e-3
    total, coef = mo.clip_coef(ora["grads"], 1.0)
    # torch's clip_grad_norm_ accumulates in fp32 (foreach norm of norms); the oracle sums squares in fp64
    assert abs(total.item() - float(g["fp32_total_norm"])) / float(g["fp32_total_norm"]) < 2e-4


def test_adamw_step_vs_golden(ora, tiny_case, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_tiny_b2.npz"))
    names = [str(n) for n in g["param_names"]]
    params = {n: tiny_case["sd"][n].clone() for n in names}
    grads = {n: ora["grads"][n].clone() for n in names}
    _, coef = mo.clip_coef(grads, 1.0)
    for n in names:
        grads[n].mul_(coef)
    m = {n: torch.zeros_like(params[n]) for n in names}
    v = {n: torch.zeros_like(params[n]) for n in names}
    mo.adamw_step(params, grads, m, v, step=1, lr=1.5e-3)
    ps = np.array([params[n].double().sum().item() for n in names])
    pa = np.array([params[n].double().abs().sum().item() for n in names])
    assert (np.abs(pa - g["fp32_post_abs_sum"]) / (g["fp32_post_abs_sum"] + 1e-9)).max() < 1e-5
    assert (np.abs(ps - g["fp32_post_sum"]) / (g["fp32_post_abs_sum"] + 1e-9)).max() < 1e-5


def test_bf16_autocast_mirror_within_envelope(tiny_case, golden_dir):
    """Our bf16 mirror and the reference under CPU autocast(bf16) are two bf16 evaluations of the same
    graph; they agree to a couple of bf16 ulps of the logit scale (|logit| <= 8 -> ulp 0.03125)."""
    g = np.load(os.path.join(golden_dir, "ref_tiny_b2.npz"))
    c = tiny_case
    pm = mo.build_padding_mask(c["text_len"])
    lb = mo.forward(c["sd"], c["dims"], c["mel"], c["tokens"], pm, autocast_bf16=True)
    s = lb[:, POS, :]
    envelope = float(g["bf16_vs_fp32_maxabs"])
    assert 0.01 < envelope < 0.2
    assert np.abs(s[..., :512].numpy() - g["bf16_head"]).max() <= 1.5 * envelope


def test_schedule_helpers():
    assert mo.accumulation_steps(512, 8, 64) == 1 and mo.accumulation_steps(2048, 8, 32) == 8
    assert mo.lr_lambda(0, 1000) == 0.0 and mo.lr_lambda(1, 1000) == 0.5 and mo.lr_lambda(2, 1000) == 1.0
    assert mo.lr_lambda(1000, 1000) == 0.0
    m = mo.build_padding_mask([3], 8)[0]
    assert (m[:, :3] == 0).all() and torch.isinf(m[:, 3:]).all()


def test_synthetic_batch_layout():
    pcm, ti, ty, tl = mo.synthetic_batch([5])
    L = int(tl[0])
    assert pcm.dtype == torch.int16 and pcm.shape == (1, 480000)
    assert ti[0, 0] == 50257 and ti[0, 1] == 50362 and ty[0, L - 1] == 50256
    assert (ti[0, L:] == mo.PAD_ID).all() and (ty[0, L:] == mo.PAD_ID).all()
    assert torch.equal(ti[0, 1:L], ty[0, :L - 1])


def test_greedy_decode_runs_and_is_eot_sticky(tiny_case):
    c = tiny_case
    dims = mo.Dims(80, 1500, 384, 6, 1, 51864, 448, 384, 6, 1)  # 1-layer toy to keep it seconds
    sd = mo.init_state_dict(dims, seed=3)
    toks = mo.greedy_decode(sd, dims, c["mel"][:1], [50257, 50362], max_new=4)
    assert toks.shape[0] == 1 and toks.shape[1] <= 6 and (toks[:, :2] == torch.tensor([50257, 50362])).all()


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not mounted (GPU box)")
def test_oracle_vs_live_reference(ora, tiny_case):
    ref_model, _, ref_dims = ref_import.load()
    c = tiny_case
    net = ref_model.OLMoASR(ref_dims.VARIANT_TO_DIMS["tiny"])
    net.load_state_dict(c["sd"], strict=True)
    pm = mo.build_padding_mask(c["text_len"])
    with torch.no_grad():
        ref = net(c["mel"], c["tokens"], pm)
    assert (ref - ora["logits"]).abs().max().item() < 5e-5


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not mounted (build container only)")
def test_oracle_qkv_attention_equals_the_reference_manual_path_live():
    """row a8: ``qk`` as the reference's MultiHeadAttention returns it -- the manual path (a 2-D mask, model.py:316-327, 347-442) hands back
    the fp32 pre-softmax scores, the SDPA path (no mask / a 3-D mask) None.  oracle.model_oracle.qkv_attention is that path, bit for bit."""
    model, _, _ = ref_import.load()
    torch.manual_seed(0)
    m = model.MultiHeadAttention(128, 2)
    x, xa = torch.randn(2, 7, 128), torch.randn(2, 11, 128)
    mask = torch.full((9, 9), float("-inf")).triu_(1)[:7, :7]
    with torch.no_grad():
        out, qk = m(x, mask=mask)
        wv, qk_o = mo.qkv_attention(m.query(x), m.key(x), m.value(x), 2, mask)
        assert qk.dtype == torch.float32 and qk.shape == (2, 2, 7, 7) and torch.equal(qk, qk_o) and torch.equal(m.out(wv), out)
        assert m(x)[1] is None and m(x, xa)[1] is None and m(x, mask=torch.zeros(2, 7, 7))[1] is None  # SDPA path: no scores
        # cross-attention scores of a whole decoder, the tensor word-timestamp alignment needs, through the oracle's block walk
        dims = mo.Dims(80, 1500, 128, 2, 1, 51864, 448, 128, 2, 2)
        sd = mo.init_state_dict(dims, seed=3)
        toks = torch.tensor([[50257, 50362, 5, 6, 7, 50256]])
        xa2 = torch.randn(1, 1500, 128)
        got = mo.cross_attention_scores(sd, dims, toks, xa2, [1])
        assert set(got) == {1} and got[1].shape == (1, 2, 6, 1500) and torch.isfinite(got[1]).all()
