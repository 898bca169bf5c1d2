"""Token-id parity of decoding and the long-form driver against the CPU oracle restatement (oracle/decode_oracle.py on
oracle/model_oracle.py): greedy, timestamp rules, beam search (+patience), suppress lists, no_speech_prob, the
timestamp-driven seek of transcribe().  The engine runs in fp32 validation mode so that logits agree to ~1e-5 and ids are
comparable without margin gating (a mismatch is accepted only where the oracle's own top-2 gap is below 1e-3).
Reference: whisper.decoding as bound at olmoasr/model.py:966-968 (un-vendored: parity unpinned by the reference),
olmoasr/transcribe.py:147-517 (followed line by line on both sides)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def _dims(mo_dims):
    from olmoasr_amd.config.model_dims import ModelDimensions
    return ModelDimensions(**{k: getattr(mo_dims, k) for k in ModelDimensions.__dataclass_fields__})


@pytest.fixture(scope="module")
def pair(tiny_case):
    """2-layer d=384 inference-layout model on both sides, same weights; fp32 engine."""
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    sd = mo.init_state_dict(dims, seed=21, train_vocab_rows=False)
    # sharpen the head a little so candidate scores are well separated (kaiming init gives near-flat distributions)
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * 3.0
    net = OLMoASR(_dims(dims), device=DEV, seed=0, inference=True, compute_dtype="float32")
    net.load_state_dict(sd)
    return net, sd, dims, tiny_case["mel"]


def _same(got, want, what):
    assert len(got) == len(want)
    for b, (g, w) in enumerate(zip(got, want)):
        assert g.tokens == w.tokens, f"{what} row {b}: native {g.tokens} vs oracle {w.tokens}"
        assert abs(g.avg_logprob - w.avg_logprob) < 2e-4, (what, b, g.avg_logprob, w.avg_logprob)
        assert abs(g.no_speech_prob - w.no_speech_prob) < 1e-5 + 1e-3 * w.no_speech_prob


def test_greedy_without_timestamps_default_suppression(pair):
    from olmoasr_amd.decoding import DecodingOptions, decode
    from oracle import decode_oracle as do
    net, sd, dims, mel = pair
    want = do.decode(sd, dims, mel, do.Options(sample_len=8, without_timestamps=True))
    for cache in (True, False):
        got = decode(net, mel.to(DEV), DecodingOptions(sample_len=8, without_timestamps=True, use_kv_cache=cache))
        _same(got, want, f"greedy cache={cache}")
    # the first sampled token is never blank / eot, specials are never sampled (SuppressBlank / SuppressTokens)
    for r in want:
        assert r.tokens and r.tokens[0] not in (220, 50256) and not any(50257 <= t <= 50361 for t in r.tokens)
    # suppression off changes nothing but the masks
    want2 = do.decode(sd, dims, mel, do.Options(sample_len=5, without_timestamps=True, suppress_tokens=None, suppress_blank=False))
    got2 = decode(net, mel.to(DEV), DecodingOptions(sample_len=5, without_timestamps=True, suppress_tokens=None, suppress_blank=False))
    _same(got2, want2, "greedy unsuppressed")


def test_greedy_with_timestamp_rules(pair):
    from olmoasr_amd.decoding import TIMESTAMP_BEGIN, DecodingOptions, decode
    from oracle import decode_oracle as do
    net, sd, dims, mel = pair
    want = do.decode(sd, dims, mel, do.Options(sample_len=10))
    got = decode(net, mel.to(DEV), DecodingOptions(sample_len=10))
    _same(got, want, "timestamps")
    for r in got:
        assert TIMESTAMP_BEGIN <= r.tokens[0] <= TIMESTAMP_BEGIN + 50
        stamps = [t for t in r.tokens if t >= TIMESTAMP_BEGIN]
        assert stamps == sorted(stamps)


@pytest.mark.parametrize("beam,patience", [(3, None), (2, 2.0)])
def test_beam_search(pair, beam, patience):
    from olmoasr_amd.decoding import DecodingOptions, decode
    from oracle import decode_oracle as do
    net, sd, dims, mel = pair
    want = do.decode(sd, dims, mel, do.Options(sample_len=6, beam_size=beam, patience=patience))
    got = decode(net, mel.to(DEV), DecodingOptions(sample_len=6, beam_size=beam, patience=patience))
    _same(got, want, f"beam {beam} patience {patience}")
    want = do.decode(sd, dims, mel, do.Options(sample_len=5, beam_size=beam, patience=patience, without_timestamps=True, length_penalty=0.6))
    got = decode(net, mel.to(DEV), DecodingOptions(sample_len=5, beam_size=beam, patience=patience, without_timestamps=True, length_penalty=0.6))
    _same(got, want, "beam, no timestamps, length penalty")


def test_training_head_with_pad_class(tiny_case):
    """The training model's head has n_vocab + 1 rows (pad class, olmoasr/model.py:665-667); whisper's decode sees all of
    them, and so do both sides here."""
    from olmoasr_amd.decoding import DecodingOptions, decode
    from olmoasr_amd.model import OLMoASR
    from oracle import decode_oracle as do
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 384, 6, 1, 51864, 448, 384, 6, 1)
    sd = mo.init_state_dict(dims, seed=3)
    net = OLMoASR(_dims(dims), device=DEV, seed=0, compute_dtype="float32")
    net.load_state_dict(sd)
    mel = tiny_case["mel"]
    _same(decode(net, mel.to(DEV), DecodingOptions(sample_len=5, without_timestamps=True)),
          do.decode(sd, dims, mel, do.Options(sample_len=5, without_timestamps=True)), "training head")


def test_transcribe_timestamp_driven_seek(pair):
    """olmoasr/transcribe.py:281-517 on both sides: the window starts (seeks), segment boundaries and token streams agree;
    a clip shorter than 30 s is zero-padded (not silence-floor padded) and agrees too."""
    from olmoasr_amd import audio as A
    from oracle import decode_oracle as do
    from oracle import model_oracle as mo
    net, sd, dims, _ = pair
    pcm = torch.cat([mo.synthetic_sample(300 + i)[0] for i in range(2)])[: 41 * 16000]  # 41 s
    kw = dict(temperature=0.0, logprob_threshold=None, no_speech_threshold=0.6, sample_len=7)
    # a random-init model rarely closes a timestamp pair by itself: a constant bonus on the timestamp ids (the same additive
    # vector on both sides) makes it emit <|t|><|t'|> pairs, so the seek is driven by them
    bias = torch.zeros(51864)
    bias[do.TIMESTAMP_BEGIN:] = 25.0
    out = net.transcribe(pcm, suppress_mask=bias, **kw)
    mel_padded = A.log_mel_spectrogram(pcm, padding=A.N_SAMPLES, device=DEV).cpu()
    want = do.transcribe(sd, dims, mel_padded, logit_bias=bias, **kw)
    assert len(want["seeks"]) < 60
    print("seeks:", want["seeks"])
    assert [s["seek"] for s in out["segments"]] == [s["seek"] for s in want["segments"]]
    assert [s["tokens"] for s in out["segments"]] == [s["tokens"] for s in want["segments"]]
    for a, b in zip(out["segments"], want["segments"]):
        assert abs(a["start"] - b["start"]) < 1e-9 and abs(a["end"] - b["end"]) < 1e-9
    assert out["tokens"] == want["tokens"] and len(want["seeks"]) >= 2
    assert any(s > 0 and s % 3000 != 0 for s in want["seeks"]), "no timestamp-driven seek happened in this case"
    # without timestamps: every window advances by a full 3000 frames, batched decoding == the sequential oracle
    out2 = net.transcribe(pcm, without_timestamps=True, batch_windows=4, **kw)
    want2 = do.transcribe(sd, dims, mel_padded, without_timestamps=True, **kw)
    assert want2["seeks"] == [0, 3000] and [s["tokens"] for s in out2["segments"]] == [s["tokens"] for s in want2["segments"]]
    assert out2["segments"][-1]["end"] == 41.0
    # short clip (11 s) and clip_timestamps
    short = pcm[: 11 * 16000]
    o3 = net.transcribe(short, clip_timestamps="2,9", **kw)
    w3 = do.transcribe(sd, dims, A.log_mel_spectrogram(short, padding=A.N_SAMPLES, device=DEV).cpu(), clip_timestamps=(2.0, 9.0), **kw)
    assert [s["tokens"] for s in o3["segments"]] == [s["tokens"] for s in w3["segments"]] and w3["seeks"][0] == 200


def test_reference_style_kv_cache_driving(pair):
    """The call pattern of whisper's PyTorchInference on the reference surface: install_kv_cache_hooks(), the prompt in one
    decoder call, then tokens[:, -1:] per step, rearrange by indexing the cache entries -- against cache-less logits()."""
    net, sd, dims, mel = pair
    xa = net.encoder(mel.to(DEV))
    assert torch.equal(xa, net.embed_audio(mel.to(DEV)))
    toks = torch.tensor([[50257, 50363, 11, 12, 13], [50257, 50370, 21, 22, 23]], device=DEV)
    full = net.decoder(toks, xa)  # no cache: [B, 5, rows]
    assert torch.equal(full, net.logits(toks, xa))
    cache, hooks = net.install_kv_cache_hooks()
    assert cache == {} and len(hooks) >= 1
    first = net.decoder(toks[:, :2], xa, kv_cache=cache)
    assert first.shape == (2, 2, 51864) and len(cache) == 1
    assert next(iter(cache.values())).shape[1] == 2  # TextDecoder.forward's `offset`
    assert float((first - full[:, :2]).abs().max()) < 1e-4
    for p in range(2, 5):
        step = net.decoder(toks[:, p:p + 1], xa, kv_cache=cache)
        assert float((step[:, 0] - full[:, p]).abs().max()) < 1e-4
    # rearrange_kv_cache: swap the two sequences, then one more token
    for module, tensor in list(cache.items()):
        cache[module] = tensor[[1, 0]].detach()
    nxt = torch.tensor([[31], [41]], device=DEV)
    step = net.decoder(nxt, xa[[1, 0]], kv_cache=cache)
    want = net.logits(torch.cat([toks[[1, 0]], nxt], 1), xa[[1, 0]])[:, -1]
    assert float((step[:, 0] - want).abs().max()) < 1e-4
    for h in hooks:
        h.remove()
    with pytest.raises(ValueError):
        net.detect_language(mel.to(DEV))


def test_pick_tokens_kernel():
    from olmoasr_amd import ops
    g = torch.Generator().manual_seed(5)
    lg = (torch.randn(7, 51865, generator=g) * 3).to(DEV)
    m1 = torch.zeros(51865, device=DEV)
    m1[[50257, 50358, 220]] = -float("inf")
    m2 = torch.zeros(51865, device=DEV)
    m2[50256] = -float("inf")
    lg[3, 50256] = 100.0  # would win unless masked
    lg[4, 17] = lg[4, 40000] = 90.0  # tie: lowest index
    tok, lp = ops.pick_tokens(lg, m1, m2)
    ref = lg + m1 + m2
    assert torch.equal(tok, ref.argmax(-1)) and int(tok[4]) == 17 and int(tok[3]) != 50256
    want = torch.log_softmax(ref, -1).gather(1, tok[:, None])[:, 0]
    assert float((lp - want).abs().max()) < 1e-4
    tok2, none = ops.pick_tokens(lg[:, :51864].contiguous(), want_logprob=False)
    assert none is None and torch.equal(tok2, lg[:, :51864].argmax(-1))


def test_small_dims_two_windows_greedy_transcribe():
    """BASELINE config 5's model size (OLMoASR-small: 12 + 12 layers, d = 768, inference head) on two 30 s windows of the
    seeded generator: (i) the fp32 engine's greedy token stream equals the CPU oracle's transcribe() exactly, (ii) the bf16
    production engine agrees with it wherever the oracle's own top-2 margin clears the bf16 envelope (random-init weights:
    margins are small, so the gate matters; the count of compared positions is reported)."""
    from olmoasr_amd import audio as A
    from olmoasr_amd.model import OLMoASR
    from oracle import decode_oracle as do
    from oracle import model_oracle as mo
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dims = mo.VARIANTS["small"]
    sd = mo.init_state_dict(dims, seed=2, train_vocab_rows=False)
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * 3.0  # sharper head: larger margins
    pcm = torch.cat([mo.synthetic_sample(500 + i)[0] for i in range(2)])  # 60 s
    kw = dict(temperature=0.0, logprob_threshold=None, no_speech_threshold=None, sample_len=6, without_timestamps=True)
    mel_padded = A.log_mel_spectrogram(pcm, padding=A.N_SAMPLES, device=DEV).cpu()
    want = do.transcribe(sd, dims, mel_padded, **kw)
    assert want["seeks"] == [0, 3000] and len(want["tokens"]) == 12
    net32 = OLMoASR(_dims(dims), device=DEV, seed=0, inference=True, compute_dtype="float32")
    net32.load_state_dict(sd)
    got32 = net32.transcribe(pcm, **kw)
    assert [s["tokens"] for s in got32["segments"]] == [s["tokens"] for s in want["segments"]]
    assert all(abs(a["avg_logprob"] - b["avg_logprob"]) < 2e-4 for a, b in zip(got32["segments"], want["segments"]))
    del net32
    torch.cuda.empty_cache()
    net = OLMoASR(_dims(dims), device=DEV, seed=0, inference=True)  # bf16 production engine
    net.load_state_dict(sd)
    got = net.transcribe(pcm, **kw)
    checked = agreed = 0
    for w, (sg, sw) in enumerate(zip(got["segments"], want["segments"])):
        xa = mo.encoder_forward(sd, dims, torch.nn.functional.pad(mel_padded[:, w * 3000:(w + 1) * 3000], (0, 0))[None])
        prefix = [do.SOT, do.NO_TIMESTAMPS]
        for t_got, t_want in zip(sg["tokens"], sw["tokens"]):
            lg = mo.decoder_forward(sd, dims, torch.tensor([prefix]), xa)[0, -1]
            lg[do.suppress_list(do.Options())] = -float("inf")
            if len(prefix) == 2:
                lg[[do.BLANK, do.EOT]] = -float("inf")
            top2 = lg.topk(2).values
            if float(top2[0] - top2[1]) > 0.25:  # ~3x the bf16-vs-fp32 logit envelope at this scale
                checked += 1
                agreed += int(t_got == t_want)
            if t_got != t_want:
                break  # the streams diverge after a flipped near-tie: later positions are not comparable
            prefix.append(t_want)
    print(f"small dims, bf16 engine vs oracle: {agreed}/{checked} margin-gated positions agree")
    assert checked >= 4 and agreed == checked


def _small_long_form(seconds, seed=2):
    """OLMoASR-small dims (BASELINE config 5), random init with the sharpened head, and `seconds` of the seeded generator's audio."""
    from oracle import model_oracle as mo
    dims = mo.VARIANTS["small"]
    sd = mo.init_state_dict(dims, seed=seed, train_vocab_rows=False)
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * 3.0
    pcm = torch.cat([mo.synthetic_sample(700 + i)[0] for i in range((seconds + 29) // 30)])[: seconds * 16000]
    return dims, sd, pcm


def test_c5_long_form_fp32_engine_equals_oracle_on_sampled_windows_64_tokens():
    """BASELINE config 5's run shape -- OLMoASR-small, 600 s of audio = 20 windows, greedy, KV cache, windows batched -- with a parity
    sample INSIDE it: windows 0, 9 and 19 of the fp32 engine's output against the CPU oracle's greedy decode of the same windows
    for 64 tokens each (ids exact, avg_logprob to 2e-4).  The other 17 windows are held to the batch-invariance the driver relies
    on: decoding them 20 at a time equals decoding them 4 at a time."""
    from olmoasr_amd import audio as A
    from olmoasr_amd.model import OLMoASR
    from oracle import decode_oracle as do
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dims, sd, pcm = _small_long_form(600)
    kw = dict(temperature=0.0, logprob_threshold=None, no_speech_threshold=None, without_timestamps=True, sample_len=64)
    net = OLMoASR(_dims(dims), device=DEV, seed=0, inference=True, compute_dtype="float32")
    net.load_state_dict(sd)
    got = net.transcribe(pcm, batch_windows=20, **kw)
    assert [s["seek"] for s in got["segments"]] == [3000 * i for i in range(20)]
    assert all(len(s["tokens"]) == 64 or 50256 not in s["tokens"] for s in got["segments"])
    got4 = net.transcribe(pcm, batch_windows=4, **kw)
    assert [s["tokens"] for s in got4["segments"]] == [s["tokens"] for s in got["segments"]]
    mel_padded = A.log_mel_spectrogram(pcm, padding=A.N_SAMPLES, device=DEV).cpu()
    pick = [0, 9, 19]
    wins = torch.stack([mel_padded[:, 3000 * w:3000 * (w + 1)] for w in pick])
    want = do.decode(sd, dims, wins, do.Options(sample_len=64, without_timestamps=True))
    for w, r in zip(pick, want):
        seg = got["segments"][w]
        assert seg["tokens"] == r.tokens, f"window {w}: native {seg['tokens'][:12]}... vs oracle {r.tokens[:12]}..."
        assert len(r.tokens) >= 32 and abs(seg["avg_logprob"] - r.avg_logprob) < 2e-4
    print("C5 sample: windows", pick, "x", [len(r.tokens) for r in want], "tokens identical to the oracle")


def test_c5_small_dims_memorised_bf16_greedy_transcribe_is_exact():
    """north_star: "token ids bit-exact for greedy decode" -- UNGATED, on the production bf16 engine at BASELINE config 5's model
    size.  Random-init margins are below bf16 noise, so (as tests/test_gpu_model.py does for the tiny model) the model is first
    trained BY THIS ENGINE on three 30 s windows with 64-token transcripts until it is confident; then transcribe() of the 90 s
    file on the bf16 engine, on the fp32 engine and on the CPU oracle (same weights) must all return the memorised ids."""
    from olmoasr_amd import audio as A
    from olmoasr_amd.decoding import NON_SPEECH_TOKENS_EN
    from olmoasr_amd.model import OLMoASR
    from oracle import decode_oracle as do
    from oracle import model_oracle as mo
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dims = mo.VARIANTS["small"]
    pcm = torch.cat([mo.synthetic_sample(800 + i)[0] for i in range(3)])
    mel_padded = A.log_mel_spectrogram(pcm, padding=A.N_SAMPLES, device=DEV)
    mel = torch.stack([mel_padded[:, 3000 * w:3000 * (w + 1)] for w in range(3)]).contiguous()  # the windows transcribe() will cut
    g = torch.Generator().manual_seed(11)
    ok_ids = torch.tensor([t for t in range(1000, 20000) if t not in set(NON_SPEECH_TOKENS_EN)])
    Lb = 64
    body = ok_ids[torch.randint(0, len(ok_ids), (3, Lb), generator=g)]
    toks = torch.cat([torch.tensor([[do.SOT, do.NO_TIMESTAMPS]] * 3), body, torch.full((3, 1), do.EOT)], 1)
    L = toks.shape[1]
    ti = torch.full((3, 448), mo.PAD_ID, dtype=torch.long)
    ty = ti.clone()
    ti[:, :L - 1] = toks[:, :-1]
    ty[:, :L - 1] = toks[:, 1:]
    tl = torch.full((3,), L - 1, dtype=torch.int32)
    net = OLMoASR(_dims(dims), device=DEV, seed=3)  # bf16 production engine, training head
    args = (mel, ti.to(DEV), ty.to(DEV), tl.to(DEV))
    loss, margin = None, 0.0
    for step in range(1, 801):
        net.zero_grad()
        loss, logits = net.loss_and_backward(*args, loss_scale=65536.0, return_logits=step % 25 == 0)
        net.optim_step(step=step, lr=5e-4 * min(1.0, step / 20), inv_loss_scale=1.0 / 65536.0)
        if step % 25 == 0 and step >= 100:
            # "confident" = at EVERY supervised position the target's logit leads the runner-up by far more than the bf16-vs-fp32
            # envelope (~0.1-0.25 at this size) -- including the first token, which only the audio can decide
            top2 = logits[:, :L - 1].float().topk(2, -1)
            right = bool((top2.indices[..., 0] == ty[:, :L - 1].to(DEV)).all())
            margin = float((top2.values[..., 0] - top2.values[..., 1]).min()) if right else 0.0
            if margin > 4.0:
                break
    print(f"memorised after {step} steps, loss {float(loss):.4f}, smallest top-2 margin {margin:.2f}")
    assert margin > 4.0
    sd = {k: v.detach().cpu().float() for k, v in net.state_dict().items()}
    kw = dict(temperature=0.0, logprob_threshold=None, no_speech_threshold=None, without_timestamps=True, sample_len=80)
    got = net.transcribe(pcm, **kw)
    want = do.transcribe(sd, dims, mel_padded.cpu(), **kw)
    assert want["seeks"] == [0, 3000, 6000]
    assert [s["tokens"] for s in want["segments"]] == body.tolist(), "the oracle does not reproduce the memorised transcripts"
    assert [s["tokens"] for s in got["segments"]] == [s["tokens"] for s in want["segments"]]  # bf16 engine, every position, no gate
    del net
    torch.cuda.empty_cache()
    net32 = OLMoASR(_dims(dims), device=DEV, seed=0, compute_dtype="float32")
    net32.load_state_dict(sd)
    got32 = net32.transcribe(pcm, **kw)
    assert [s["tokens"] for s in got32["segments"]] == [s["tokens"] for s in want["segments"]]
    assert all(abs(a["avg_logprob"] - b["avg_logprob"]) < 2e-4 for a, b in zip(got32["segments"], want["segments"]))


@pytest.mark.parametrize("length", [0, 1, 2, 5, 40])
def test_pick_tokens_ts_kernel_equals_the_rules_then_argmax(length):
    """oasr_pick_tokens_ts (ApplyTimestampRules + argmax + log-softmax of the pick in one kernel, history read on the device)
    against the tensor form of the rules -- itself pinned to transformers' WhisperTimeStampLogitsProcessor on the CPU
    (tests/test_decoding_rules_cpu.py) -- followed by torch argmax / log_softmax: same ids, same log-probabilities, on random
    histories that reach every branch (pairs, open timestamps, repeats, text only), with and without the forced-timestamp rule,
    with the suppress masks, for the inference and the training head."""
    from olmoasr_amd import decoding as dec
    from olmoasr_amd import ops
    from test_decoding_rules_cpu import _random_history
    g = torch.Generator().manual_seed(40 + length)
    for rows_v in (51864, 51865):
        n, sample_begin = 24, 1
        tokens = _random_history(g, n, length, sample_begin).to(DEV)
        logits = (torch.randn(n, rows_v, generator=g) * 3.0).to(DEV)
        logits[: n // 2, dec.TIMESTAMP_BEGIN:] += 6.0  # half the rows: enough timestamp mass to force a timestamp
        base = torch.zeros(rows_v, device=DEV)
        base[dec.suppress_list(dec.DecodingOptions())] = -float("inf")
        first = torch.zeros(rows_v, device=DEV)
        first[[dec.BLANK, dec.EOT]] = -float("inf")
        for max_init in (None, 50):
            for fm in (None, first):
                ref = logits + base + (fm if fm is not None else 0.0)
                dec._timestamp_rules(ref, tokens, sample_begin, max_init)
                want_tok = ref.argmax(-1)
                want_lp = torch.log_softmax(ref.float(), -1).gather(1, want_tok[:, None])[:, 0]
                tok, lp = ops.pick_tokens_ts(logits, tokens[:, sample_begin:], length, timestamp_begin=dec.TIMESTAMP_BEGIN, eot=dec.EOT,
                                             no_timestamps=dec.NO_TIMESTAMPS, max_initial_index=max_init, mask=base, mask2=fm)
                assert torch.equal(tok, want_tok), (length, rows_v, max_init, (tok != want_tok).nonzero().flatten().tolist())
                assert float((lp - want_lp).abs().max()) < 2e-4


@pytest.mark.parametrize("length", [None, 0, 1, 2, 5, 40])
def test_topk_and_sample_kernels_equal_the_rules_then_torch(length):
    """oasr_topk_tokens / oasr_sample_tokens (round 4: beam search and temperature sampling select on the device like greedy does)
    against the tensor form of the filters (suppress masks; ``length`` not None: whisper's ApplyTimestampRules on a random history,
    pinned to transformers' processor on the CPU) followed by torch: log_softmax().topk(k) -- same ids, same values -- and the
    inverse CDF of softmax(filtered / T) in float64 at the same uniforms -- same draws, scored with the T = 1 log_softmax
    (whisper.decoding BeamSearchDecoder.update / GreedyDecoder.update).  ``length`` None = the without_timestamps mode (no rules)."""
    from olmoasr_amd import decoding as dec
    from olmoasr_amd import ops
    from test_decoding_rules_cpu import _random_history
    g = torch.Generator().manual_seed(140 + (length or 0))
    for rows_v in (51864, 51865):
        n, sample_begin = 24, 1
        tokens = _random_history(g, n, length or 0, sample_begin).to(DEV)
        logits = (torch.randn(n, rows_v, generator=g) * 3.0).to(DEV)
        logits[: n // 2, dec.TIMESTAMP_BEGIN:] += 6.0  # half the rows: enough timestamp mass to force a timestamp (when the rules are on)
        logits[3, 1000] = logits[3, 2000] = float(logits[3].max()) + 1.0  # a tie at the top: the lower id comes first
        base = torch.zeros(rows_v, device=DEV)
        base[dec.suppress_list(dec.DecodingOptions())] = -float("inf")
        first = torch.zeros(rows_v, device=DEV)
        first[[dec.BLANK, dec.EOT]] = -float("inf")
        for max_init in (None, 50):
            for fm in (None, first):
                ref = logits + base + (fm if fm is not None else 0.0)
                if length is not None:
                    dec._timestamp_rules(ref, tokens, sample_begin, max_init)
                sel = dict(history=tokens[:, sample_begin:], n_history=length, timestamp_begin=dec.TIMESTAMP_BEGIN, eot=dec.EOT,
                           no_timestamps=dec.NO_TIMESTAMPS, max_initial_index=max_init, mask=base, mask2=fm)
                lsm = torch.log_softmax(ref.double(), -1)
                for k in (1, 6):
                    want_lp, want_tok = lsm.topk(k, -1)
                    lp, tok = ops.topk_tokens(logits, k, **sel)
                    # (torch.topk does not define the order of equal values: compare values, and ids wherever the value is unique)
                    assert float((lp.double() - want_lp).abs().max()) < 2e-4, (length, rows_v, k)
                    uniq = torch.ones_like(tok, dtype=torch.bool)
                    if k > 1:
                        uniq[:, 1:] &= want_lp[:, 1:] != want_lp[:, :-1]
                        uniq[:, :-1] &= want_lp[:, 1:] != want_lp[:, :-1]
                    assert torch.equal(tok[uniq], want_tok[uniq]), (length, rows_v, k)
                    assert bool((lsm.gather(1, tok) - lp.double()).abs().max() < 2e-4)  # every returned id carries its own log-probability
                if length is None or max_init is None:
                    lp2, tok2 = ops.topk_tokens(logits, 2, **sel)
                    if not (length is not None and 3 < n // 2):  # (row 3 is a forced-timestamp row when the rules are on)
                        assert tok2[3].tolist() == [1000, 2000]
                for T in (0.4, 1.0):
                    u = torch.rand(n, generator=g).to(DEV)
                    u[0], u[1] = 0.0, 0.999999
                    tok, lp = ops.sample_tokens(logits, T, u, **sel)
                    p = torch.softmax(ref.double() / T, -1)
                    cdf = p.cumsum(-1)
                    want = torch.searchsorted(cdf, (u.double() * cdf[:, -1])[:, None]).clamp(max=rows_v - 1)[:, 0]
                    # a draw within float32 rounding of a CDF step may land on either side: accept the neighbouring SURVIVOR there
                    near = ((cdf.gather(1, want[:, None])[:, 0] - u.double() * cdf[:, -1]).abs() < 1e-5) | \
                           ((cdf.gather(1, tok[:, None])[:, 0] - p.gather(1, tok[:, None])[:, 0] - u.double() * cdf[:, -1]).abs() < 1e-5)
                    assert bool(((tok == want) | near).all()), (length, rows_v, T, (tok != want).nonzero().flatten().tolist())
                    assert bool((p.gather(1, tok[:, None]) > 0).all())  # never a masked token
                    assert float((lp.double() - lsm.gather(1, tok[:, None])[:, 0]).abs().max()) < 2e-4


def test_sampling_frequencies_follow_the_tempered_distribution():
    """20,000 rows of the same small distribution (everything but 6 tokens masked), independent uniforms: empirical frequencies of
    oasr_sample_tokens within 4 sigma of softmax(logits / T)."""
    from olmoasr_amd import ops
    V, n, T = 51864, 20000, 0.7
    keep = torch.tensor([5, 300, 301, 20000, 50256, 51000])
    vals = torch.tensor([1.0, 0.0, -1.0, 2.0, 0.5, -0.5])
    mask = torch.full((V,), -float("inf"))
    mask[keep] = 0.0
    row = torch.zeros(V)
    row[keep] = vals
    logits = row.to(DEV)[None].expand(n, V).contiguous()
    u = torch.rand(n, generator=torch.Generator().manual_seed(3)).to(DEV)
    tok, lp = ops.sample_tokens(logits, T, u, mask=mask.to(DEV))
    p = torch.softmax(vals.double() / T, -1)
    for j, t in enumerate(keep.tolist()):
        f = float((tok == t).float().mean())
        sigma = float((p[j] * (1 - p[j]) / n).sqrt())
        assert abs(f - float(p[j])) < 4 * sigma + 1e-4, (t, f, float(p[j]))
    assert int(torch.isin(tok.cpu(), keep).sum()) == n
    want_lp = torch.log_softmax(vals.double(), -1)
    for j, t in enumerate(keep.tolist()):
        sel = tok == t
        if bool(sel.any()):
            assert float((lp[sel].double() - want_lp[j]).abs().max()) < 1e-5
