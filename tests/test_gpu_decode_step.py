"""The KV-cached decoder step's engines -- separate LayerNorm kernels, LayerNorm folded into the projections' operand loads
(csrc/decode_proj.hip), and the ONE-launch engine of csrc/decode_xcd.hip (up to 4 sequences; on one XCD or spread over the chip) --
against each other (bit-identical) and against the cache-less decoder; the chip-wide one-launch engine for ONE sequence (csrc/decode_wide.hip,
the default at B = 1) against them within the fp32 rounding of its differently ordered K sums.
Reference semantics: TextDecoder.forward with the kv_cache hooks, olmoasr/model.py:786-817, 925-964."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dims(mo_dims):
    from olmoasr_amd.config.model_dims import ModelDimensions
    return ModelDimensions(**{k: getattr(mo_dims, k) for k in ModelDimensions.__dataclass_fields__})


def _tail(state):
    """the control words at the start of the cache's tail (include/oasr.h, OASR_KV_TAIL_BYTES)"""
    from olmoasr_amd import _native as N
    return state["cache"][-N.KV_TAIL_BYTES:].view(torch.int32)


def _steps(net, xa, toks, mode):
    """mode: 1 = LayerNorm folded into the projections; 0 = separate LayerNorm / logits-widening kernels; 2 / 3 / 4 = one launch for the
    whole decoder stack (one XCD / 32 / 64 workgroups spread), where the shape allows it (else the call falls back to the folded path);
    5 = the chip-wide one-launch engine (one sequence; more sequences: as 2); -1 = the library's default."""
    from olmoasr_amd import _native as N
    N.lib().oasr_decode_set_ln_fold(mode)
    try:
        st = net.kv_cache_begin(xa)
        out = [net.kv_cache_step(st, toks[:, p]) for p in range(toks.shape[1])]
        net.kv_cache_check(st)
        return torch.stack(out, 1)
    finally:
        N.lib().oasr_decode_set_ln_fold(-1)


@pytest.mark.parametrize("width,heads,layers,B,inference", [(384, 6, 4, 2, True), (384, 6, 2, 7, False), (768, 12, 2, 16, True),
                                                            (512, 8, 3, 32, True), (1024, 16, 1, 1, False), (768, 12, 2, 1, True), (384, 6, 3, 1, True), (1280, 20, 1, 1, True), (512, 8, 3, 3, True),
                                                            (1024, 16, 2, 4, False), (1280, 20, 1, 2, True)])
def test_step_engines_are_bit_identical(tiny_case, width, heads, layers, B, inference):
    """Same rounding points, same skinny-GEMM accumulation scheme: the two placements must agree to the last bit on every
    position; both must track the cache-less decoder to bf16 accumulation-order noise."""
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, width, heads, 1, 51864, 448, width, heads, layers)
    net = OLMoASR(_dims(dims), device=DEV, seed=5 + layers, inference=inference)
    mel = tiny_case["mel"].to(DEV)
    mel = mel.repeat((B + 1) // 2, 1, 1)[:B] * torch.linspace(1.0, 0.8, B, device=DEV)[:, None, None]
    xa = net.embed_audio(mel)
    g = torch.Generator().manual_seed(B)
    toks = torch.randint(0, 50000, (B, 9), generator=g).to(DEV)
    toks[:, 0] = 50257
    multi = _steps(net, xa, toks, 0)
    folded = _steps(net, xa, toks, 1)
    assert torch.isfinite(folded).all()
    assert torch.equal(folded, multi), float((folded - multi).abs().max())
    scale = float(multi.abs().max())
    full = net.logits(toks, xa)
    assert float((folded - full).abs().max()) < 0.08 + 0.02 * scale
    # a second window on the same buffers
    again = _steps(net, xa, toks, 1)
    assert torch.equal(again, folded)
    # the one-launch engine (B <= 4): every placement of its team, twice on the same buffers (the barrier epoch carries over)
    if B <= 4:
        for mode in (2, 3, 4, 2) + ((-1,) if B > 1 else ()):
            one = _steps(net, xa, toks, mode)
            assert torch.equal(one, multi), (mode, float((one - multi).abs().max()))
    # the chip-wide engine (one sequence; the default there): same rounding points, K sums in 512-element spans instead of the MFMA tiles' order
    # -> equal up to bf16 rounding flips of intermediate rows; deterministic, and twice on the same buffers (the flag epoch carries over)
    if B == 1:
        wide = _steps(net, xa, toks, 5)
        assert torch.isfinite(wide).all()
        assert float((wide - multi).abs().max()) < 0.08 + 0.02 * scale, float((wide - multi).abs().max())
        assert float((wide - full).abs().max()) < 0.08 + 0.02 * scale
        assert float((wide - multi).abs().mean()) < 0.01 + 0.002 * scale
        for mode in (-1, 5):
            assert torch.equal(_steps(net, xa, toks, mode), wide), mode


def test_one_launch_engine_over_a_long_window(tiny_case):
    """300 positions on a small-dims model, B = 2: the self-attention of the one-launch engine crosses its 256-key batch boundary, the ring
    wraps thousands of times, the barrier counter runs up -- logits bit-identical to the multi-launch step at every position."""
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 512, 8, 1, 51864, 448, 512, 8, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=3, inference=True)
    mel = tiny_case["mel"].to(DEV)
    xa = net.embed_audio(mel)
    toks = torch.randint(0, 50000, (2, 300), generator=torch.Generator().manual_seed(1)).to(DEV)
    toks[:, 0] = 50257
    multi = _steps(net, xa, toks, 1)
    one = _steps(net, xa, toks, 2)
    assert torch.isfinite(one).all()
    assert torch.equal(one, multi), float((one - multi).abs().max())


def test_chip_wide_engine_over_a_long_window(tiny_case):
    """ONE sequence, all 448 positions of the text context: the chip-wide engine's self-attention fills every one of its 7 key steps per lane
    group (72 groups: the step that holds row ``pos`` moves through all of them), the flag epochs run up to 448 x 16 -- logits track the
    multi-launch step at every position and the greedy choice agrees wherever the top-2 margin is above the bf16 envelope."""
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 512, 8, 1, 51864, 448, 512, 8, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=3, inference=True)
    xa = net.embed_audio(tiny_case["mel"][:1].to(DEV))
    toks = torch.randint(0, 50000, (1, 448), generator=torch.Generator().manual_seed(1)).to(DEV)
    toks[:, 0] = 50257
    multi = _steps(net, xa, toks, 1)
    wide = _steps(net, xa, toks, 5)
    assert torch.isfinite(wide).all()
    scale = float(multi.abs().max())
    env = 0.08 + 0.02 * scale
    assert float((wide - multi).abs().max()) < env, float((wide - multi).abs().max())
    top2 = multi.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * env
    assert bool((wide.argmax(-1) == multi.argmax(-1))[clear].all())


def test_decode_through_every_step_engine(tiny_case):
    from olmoasr_amd import _native as N
    from olmoasr_amd.decoding import DecodingOptions, decode
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=11, inference=True)
    mel = tiny_case["mel"].to(DEV)
    r_f = decode(net, mel, DecodingOptions(sample_len=6, use_kv_cache=True, without_timestamps=True))
    for mode in (1, 0):
        N.lib().oasr_decode_set_ln_fold(mode)
        try:
            r_m = decode(net, mel, DecodingOptions(sample_len=6, use_kv_cache=True, without_timestamps=True))
        finally:
            N.lib().oasr_decode_set_ln_fold(-1)
        for a, b in zip(r_f, r_m):
            assert a.tokens == b.tokens and abs(a.avg_logprob - b.avg_logprob) < 1e-6


def test_reindexed_cache_steps_on_the_one_launch_engine(tiny_case):
    """whisper's rearrange_kv_cache path (``cache[module][source_indices]``, model._EngineKV.__getitem__) builds a NEW cache buffer: its
    256-byte control tail (barrier counter, error flag, epoch base, XCC mask) must start zeroed like decode_begin leaves it -- a re-indexed
    one-sequence cache continues on the one-launch engine bit-identically to the original, and the window's check stays clean."""
    from olmoasr_amd import _native as N
    from olmoasr_amd.model import OLMoASR, _EngineKV
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 512, 8, 1, 51864, 448, 512, 8, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=4, inference=True)
    mel = tiny_case["mel"].to(DEV)
    xa = net.embed_audio(mel)  # two sequences
    toks = torch.randint(0, 50000, (2, 8), generator=torch.Generator().manual_seed(2)).to(DEV)
    toks[:, 0] = 50257
    N.lib().oasr_decode_set_ln_fold(2)
    try:
        st = net.kv_cache_begin(xa)
        for p in range(4):
            net.kv_cache_step(st, toks[:, p])
        # garbage where a fresh torch.empty could have it: the allocator hands this block out again for the re-indexed cache
        junk = torch.full((N.lib().oasr_kv_cache_bytes(net._ctx, 1),), 0xA5, dtype=torch.uint8, device=DEV)
        del junk
        one = _EngineKV(net, st)[[1]].state  # sequence 1 alone: B = 1
        assert one["B"] == 1 and int(one["cache"][-N.KV_TAIL_BYTES:].to(torch.int32).sum()) == 0
        ref = [net.kv_cache_step(st, toks[:, p])[1] for p in range(4, 8)]
        got = [net.kv_cache_step(one, toks[1:, p])[0] for p in range(4, 8)]
        assert net.kv_cache_check(one) is True and net.kv_cache_check(st) is True
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    finally:
        N.lib().oasr_decode_set_ln_fold(-1)


def test_poisoned_one_launch_engine_falls_back_to_the_multi_launch_engine(tiny_case):
    """A team of the one-launch engine that cannot be resident at once (shared / CU-masked device) poisons its barrier flag instead of
    hanging.  oasr_decode_check then returns OASR_ERETRY once, clears the flag and disables the one-launch engine for the CONTEXT;
    ``decode`` repeats the window and returns the tokens of the multi-launch engine."""
    import warnings
    from olmoasr_amd.decoding import DecodingOptions, decode
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 512, 8, 1, 51864, 448, 512, 8, 2)
    net = OLMoASR(_dims(dims), device=DEV, seed=9, inference=True)
    mel = tiny_case["mel"][:1].to(DEV)
    opts = DecodingOptions(sample_len=6, use_kv_cache=True, without_timestamps=True)
    clean = decode(net, mel, opts)
    xa = net.embed_audio(mel)
    st = net.kv_cache_begin(xa)
    net.kv_cache_step(st, torch.tensor([50257], device=DEV))
    _tail(st)[1] = 1  # what a timed-out team barrier / flag poll writes
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert net.kv_cache_check(st) is False
    assert w and "multi-launch" in str(w[0].message)
    assert int(_tail(st)[1]) == 0 and net.kv_cache_check(st) is True  # flag cleared, context switched
    again = decode(net, mel, opts)  # multi-launch engine now (the same arithmetic; the chip-wide engine's K sums are ordered differently)
    assert again[0].tokens == clean[0].tokens and abs(again[0].avg_logprob - clean[0].avg_logprob) < 2e-2
    # decode() itself repeats a window whose check asks for it
    net2 = OLMoASR(_dims(dims), device=DEV, seed=9, inference=True)
    calls = {"n": 0}
    real = net2.kv_cache_check

    def poison_first(state):
        calls["n"] += 1
        if calls["n"] == 1:
            _tail(state)[1] = 1
        return real(state)
    net2.kv_cache_check = poison_first
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = decode(net2, mel, opts)
    assert calls["n"] == 2 and r[0].tokens == clean[0].tokens


@pytest.mark.parametrize("width,heads,layers", [(768, 12, 3), (1024, 16, 2)])
def test_default_engine_for_one_sequence_is_the_one_launch_engine(tiny_case, width, heads, layers):
    """All step engines are bit-identical, so a default that silently falls back to the multi-launch kernels passes every parity test --
    and costs 10-30 % per token (it happened: a host-side guard rejected the product's own arena layout, whose decoder blocks are stored
    last-first, i.e. with a negative layer stride).  The one-launch engine leaves a trace: its team barrier counter in the cache's tail."""
    from olmoasr_amd import _native as N
    from olmoasr_amd.model import OLMoASR
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, width, heads, 1, 51864, 448, width, heads, layers)
    net = OLMoASR(_dims(dims), device=DEV, seed=2, inference=True)
    xa = net.embed_audio(tiny_case["mel"][:1].to(DEV))
    st = net.kv_cache_begin(xa)
    for p in range(3):
        net.kv_cache_step(st, torch.tensor([50257 + p], device=DEV))
    assert net.kv_cache_check(st) is True
    # the chip-wide engine's trace: its flag epoch (control word 4) = phases completed, every workgroup's flag at that epoch
    epoch = int(_tail(st)[4])
    assert epoch == 3 * 8 * layers, f"flag epoch {epoch}: the chip-wide one-launch engine did not run"
    for rep in range(8):  # (one replica of the flag array per XCC, 4 KB apart, the first 4 KB into the tail)
        flags = _tail(st)[1024 * (rep + 1):1024 * (rep + 1) + 256]
        assert int(flags.min()) == epoch and int(flags.max()) == epoch
    assert int(_tail(st)[0]) == 0
    # forced onto the one-XCD team: its barrier counter runs instead
    N.lib().oasr_decode_set_ln_fold(2)
    try:
        st2 = net.kv_cache_begin(xa)
        for p in range(3):
            net.kv_cache_step(st2, torch.tensor([50257 + p], device=DEV))
        assert net.kv_cache_check(st2) is True
    finally:
        N.lib().oasr_decode_set_ln_fold(-1)
    counter = int(_tail(st2)[0])
    assert counter >= 3 * 8 * layers and int(_tail(st2)[4]) == 0, f"team barrier counter {counter}: the one-XCD engine did not run"
    two = net.kv_cache_begin(xa.repeat(2, 1, 1))  # two sequences: the multi-launch kernels by default (they spread over the chip)
    net.kv_cache_step(two, torch.tensor([50257, 50257], device=DEV))
    assert net.kv_cache_check(two) is True and int(_tail(two)[0]) == 0 and int(_tail(two)[4]) == 0
