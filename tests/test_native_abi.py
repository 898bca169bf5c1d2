"""CPU-side checks of the C-ABI library: it loads, and exports every symbol include/oasr.h declares
(no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    from olmoasr_amd import _native
    if not os.path.isfile(_native.LIB_PATH):
        g.build()
    return _native


def test_library_exports_every_declared_symbol(native):
    lib = native.lib()
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):  # oasr.h (product ABI) + oasr_testing.h (hooks)
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        found = set(re.findall(r"\b(oasr_[a-z0-9_]+)\s*\(", hdr))
        assert found, h
        names |= found
    assert len(names) >= 40
    product = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "oasr.h")).read(), flags=re.S)
    assert "oasr_probe_" not in product and "oasr_profile_" not in product  # hooks stay out of the product header
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/oasr.h but not exported"
    assert set(native.EXPORTS) <= names | {"oasr_last_error"}
    # ABI generation: the header's constant, the library's answer and the binding's expectation agree (and so do the struct sizes the
    # binding passes by pointer) -- a stale library is refused by _native.lib() itself
    hdr_ver = int(re.search(r"#define\s+OASR_ABI_VERSION\s+(\d+)", product).group(1))
    assert lib.oasr_version() == hdr_ver == native.ABI_VERSION
    assert lib.oasr_sizeof_attn_args() == ctypes.sizeof(native.AttnArgs)
    assert int(re.search(r"#define\s+OASR_ROWTAB\s+(\d+)", product).group(1)) == native.ROWTAB


def test_param_table_matches_reference_state_dict(native):
    """The engine's parameter table must carry exactly the reference's state_dict names/shapes."""
    from oracle import model_oracle as mo
    lib = native.lib()
    for variant in ("tiny", "base"):
        dims = mo.VARIANTS[variant]
        cd = native.Dims(*[getattr(dims, f[0]) for f in native.Dims._fields_])
        ctx = lib.oasr_create(ctypes.byref(cd))
        assert ctx
        sd = mo.init_state_dict(dims, seed=0)
        n = lib.oasr_param_count(ctx)
        got = {}
        end = 0
        for i in range(n):
            name = ctypes.create_string_buffer(128)
            off, numel, ndim = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
            shape = (ctypes.c_int64 * 4)()
            assert lib.oasr_param_info(ctx, i, name, 128, ctypes.byref(off), ctypes.byref(numel), ctypes.byref(ndim), shape) == 0
            assert off.value == end and off.value % 8 == 0
            end += numel.value
            got[name.value.decode()] = tuple(shape[j] for j in range(ndim.value))
        assert end == lib.oasr_param_numel(ctx) and end % 8 == 0
        want = {k: tuple(v.shape) for k, v in sd.items() if k != "encoder.positional_embedding"}
        assert got == want
        # gradient segments tile the arena exactly once
        segs = []
        for i in range(lib.oasr_segment_count(ctx)):
            o, m = ctypes.c_int64(), ctypes.c_int64()
            assert lib.oasr_segment_info(ctx, i, ctypes.byref(o), ctypes.byref(m)) == 0
            segs.append((o.value, m.value))
        pos = 0
        for o, m in sorted(segs):
            assert o == pos
            pos += m
        assert pos == end
        assert lib.oasr_workspace_bytes(ctx, 2, 448, 1) > lib.oasr_workspace_bytes(ctx, 2, 448, 0) > 0
        lib.oasr_destroy(ctx)


def test_null_context_is_rejected_loudly(native):
    lib = native.lib()
    assert lib.oasr_create(None) is None
    assert b"null dims" in lib.oasr_last_error()
    with pytest.raises(native.NativeError):
        native.require_gpu(__import__("torch").zeros(1), "x")


def test_host_side_audio_mirror(native):
    """Host logic of the whisper.audio mirror (no GPU): constants, pad_or_trim on arrays/tensors, and the library's
    slaney filterbank against the oracle's (and thereby against transformers' WhisperFeatureExtractor)."""
    import numpy as np
    import torch
    import olmoasr_amd
    from olmoasr_amd import audio
    from oracle import mel_oracle as me
    assert (audio.SAMPLE_RATE, audio.N_FFT, audio.HOP_LENGTH, audio.N_SAMPLES, audio.N_FRAMES, audio.FRAMES_PER_SECOND) == \
        (16000, 400, 160, 480000, 3000, 100)
    x = np.arange(10, dtype=np.float32)
    for L in (4, 10, 16):
        assert np.array_equal(audio.pad_or_trim(x, L), me.pad_or_trim(x, L))
        assert torch.equal(audio.pad_or_trim(torch.from_numpy(x), L), torch.from_numpy(me.pad_or_trim(x, L)))
    y = np.ones((3, 5), np.float32)
    assert audio.pad_or_trim(y, 7, axis=0).shape == (7, 5)
    assert np.array_equal(audio.mel_filters().numpy(), me.mel_filters())
    assert olmoasr_amd.N_FRAMES == 3000 and olmoasr_amd.ModelDimensions is not None
    with pytest.raises(RuntimeError, match="Failed to load audio"):  # whisper.audio.load_audio's error for an unreadable file
        audio.log_mel_spectrogram("no_such_clip.wav")


def test_train_script_host_logic():
    """Accumulation rule, LR schedule and GradScaler bookkeeping of the torchrun entry vs the oracle's restatement of
    train_timestamps.py:764-781 and torch's GradScaler defaults."""
    import importlib.util
    from oracle import model_oracle as mo
    spec = importlib.util.spec_from_file_location("tt", os.path.join(ROOT, "scripts", "training", "train_timestamps.py"))
    tt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tt)
    for eff, w, b in ((512, 8, 64), (2048, 8, 32), (4096, 8, 64), (8, 1, 8), (7, 2, 8)):
        assert tt.accumulation_steps(eff, w, b) == mo.accumulation_steps(eff, w, b)
    for n in (10, 1000, 524288):
        for s in (0, 1, 2, 5, n // 2, n - 1, n):
            assert tt.lr_lambda(s, n) == mo.lr_lambda(s, n)
    sc = tt.GradScalerState()
    sc.update(True)
    assert sc.scale == 32768.0
    for _ in range(2000):
        sc.update(False)
    assert sc.scale == 65536.0 and sc.growth_tracker == 0
    a = tt.parse_args(["--model_variant", "medium", "--betas", "(0.9, 0.98)", "--eff_batch_size", "2048"])
    assert a.model_variant == "medium" and a.eff_batch_size == 2048


def test_module_forward_mask_mapping():
    """MultiHeadAttention.forward takes the reference's additive masks (causal [T,T], model.py:740; causal + key padding [B,T,T],
    train_timestamps.py:314-315 + model.py:741) and hands the kernels (causal, kv_len); anything else must be refused, not approximated."""
    import math

    import pytest
    import torch
    from olmoasr_amd import _native as N
    from olmoasr_amd.model import _mask_to_native
    T = 12
    causal = torch.full((T, T), -math.inf).triu_(1)
    assert _mask_to_native(None, 2, T, T) == (False, None)
    c, kv = _mask_to_native(causal, 2, T, T)
    assert c is True and kv is None
    pad = torch.zeros(2, T, T)
    pad[0, :, 5:] = -math.inf
    pad[1, :, 9:] = -math.inf
    c, kv = _mask_to_native(pad + causal, 2, T, T)
    assert c is True and kv.tolist() == [5, 9] and kv.dtype == torch.int32
    c, kv = _mask_to_native((pad + causal)[:, :7, :7], 2, 7, 7)  # a prefix of the context, as TextDecoder slices it
    assert kv.tolist() == [5, 7]
    hole = (pad + causal).clone()
    hole[0, 4, 1] = -math.inf
    c, kv = _mask_to_native(pad, 2, T, T)  # key padding without the triangle
    assert c is False and kv.tolist() == [5, 9]
    assert _mask_to_native(torch.zeros(T, T), 2, T, T) == (False, None)
    for bad in (hole, causal + 1.0, causal.t()):  # a hole, a finite bias, an anti-causal triangle
        with pytest.raises(N.NativeError):
            _mask_to_native(bad, 2, T, T)
    with pytest.raises(N.NativeError):
        _mask_to_native(causal, 2, T, T + 1)


def test_testing_hooks_are_inert_without_the_opt_in(native, monkeypatch):
    """The kernel-selection setters of include/oasr_testing.h live in the product library; without OASR_TESTING_HOOKS=1 in the
    process environment they must fail and change nothing (host logic only: no GPU needed)."""
    lib = native.lib()
    monkeypatch.delenv("OASR_TESTING_HOOKS", raising=False)
    for fn, args in ((lib.oasr_gemm_force_general, (1,)), (lib.oasr_attention_set_pingpong, (0,)), (lib.oasr_gemm_set_stagger, (1, 2)),
                     (lib.oasr_gemm_set_variant, (1,)), (lib.oasr_decode_set_ln_fold, (0,)), (lib.oasr_span_set_side_streams, (0,))):
        assert fn(*args) != 0
        assert b"OASR_TESTING_HOOKS" in lib.oasr_last_error()
    monkeypatch.setenv("OASR_TESTING_HOOKS", "1")
    assert lib.oasr_gemm_force_general(0) == 0 and lib.oasr_attention_set_pingpong(1) == 0 and lib.oasr_decode_set_ln_fold(-1) == 0
    assert lib.oasr_span_set_side_streams(-1) == 0 and 0 <= lib.oasr_span_side_streams() <= 15


def test_a_library_of_another_abi_generation_is_refused_before_any_symbol_is_declared(native, tmp_path, monkeypatch):
    """a stale liboasr.so (older ABI: no oasr_sizeof_attn_args, fewer entry points) must be refused by the VERSION check --
    not die with an AttributeError inside the declarations -- and must not stay cached as a half-declared handle."""
    import subprocess
    import pytest
    src = tmp_path / "stale.c"
    src.write_text("int oasr_version(void) { return 100; }\nconst char* oasr_last_error(void) { return \"\"; }\n")
    so = tmp_path / "liboasr_stale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    monkeypatch.setattr(native, "LIB_PATH", str(so))
    monkeypatch.setattr(native, "_lib", None)
    for _ in range(2):  # the second call must fail the same way (nothing cached)
        with pytest.raises(native.NativeError, match="ABI version 100"):
            native.lib()
        assert native._lib is None


@pytest.mark.parametrize("d,H,Te,M,L,team", [(1024, 16, 1500, 1, 3, 32), (768, 12, 1500, 2, 2, 32), (384, 6, 1500, 4, 2, 32), (1280, 20, 1500, 1, 2, 64),
                                              (512, 8, 700, 3, 2, 32)])
def test_one_launch_decode_step_block_stream_matches_the_consumer_loops(native, d, H, Te, M, L, team):
    """csrc/decode_xcd.hip: every streaming wave prefetches a STATIC sequence of ring blocks (weight tiles, cross-attention K/V) with a cursor
    that runs ahead of the phases that consume them.  The cursor's sequence (exported for this test) must be exactly what the consumer's nested
    loops walk: per layer qkv tiles, attn.out, cross q, cross K then V blocks of every (sequence, head, key segment) item, cross out, mlp.0,
    mlp.2 -- tile t of a phase belongs to workgroup t % team; together the workgroups cover every tile / item exactly once."""
    lib = native.lib()
    SEG = 768
    ns = -(-Te // SEG)
    seen = {}
    for wg in range(team):
        buf = (ctypes.c_int * (4 * 200000))()
        n = lib.oasr_xcd_plan_debug(d, H, Te, M, L, team, wg, buf, 200000)
        assert 0 < n <= 200000
        got = [tuple(buf[4 * i:4 * i + 4]) for i in range(n)]
        want = []
        kb_d, kb_4d = -(-(d // 64) // 4), -(-(d // 16) // 4)
        for layer in range(L):
            for seg, ntile, kb in ((0, 3 * d // 32, kb_d), (1, d // 32, kb_d), (2, d // 32, kb_d), (3, M * H * ns, None), (4, d // 32, kb_d),
                                   (5, 4 * d // 32, kb_d), (6, d // 32, kb_4d)):
                for idx, t in enumerate(range(wg, ntile, team)):
                    if seg == 3:
                        sg = t % ns
                        nkeys = min(SEG, Te - sg * SEG)
                        nb = 2 * (-(-(-(-nkeys // 32)) // 4))
                    else:
                        nb = kb
                    want += [(layer, seg, idx, sub) for sub in range(nb)]
                    seen[(layer, seg, t)] = seen.get((layer, seg, t), 0) + 1
        assert got == want, (wg, got[:8], want[:8])
    # coverage: every tile / item of every phase exactly once over the team
    for layer in range(L):
        for seg, ntile in ((0, 3 * d // 32), (1, d // 32), (2, d // 32), (3, M * H * ns), (4, d // 32), (5, 4 * d // 32), (6, d // 32)):
            assert all(seen.get((layer, seg, t)) == 1 for t in range(ntile)), (layer, seg)


def test_one_launch_decode_refuses_layouts_past_the_32_bit_buffer_offsets(native):
    """csrc/decode_xcd.hip addresses weights and cross K/V through raw buffer descriptors with 32-bit byte offsets: the host-side guard
    (decode_xcd_offsets_ok) accepts the real layouts -- OLMoASR-large's decoder ends at ~1.7 GB of bf16 shadow -- and sends anything
    that would reach 2 GiB to the multi-launch step instead of reading zeros."""
    lib = native.lib()

    def layer0(d):  # a decoder layer as the arena lays it out: LayerNorm / bias vectors between the matrices (element offsets)
        off, o = 0, []
        for n in (d, d, 3 * d * d, 0, d * d, d, d, d, d * d, d, d * d, d, d, d, 4 * d * d, 4 * d, 4 * d * d, d):
            o.append(off)
            off += n
        return (ctypes.c_int64 * 18)(*o), off + 2 * d * d + d  # (+ the cross key | value weights, which only decode_begin reads)
    for name, d, L in (("tiny", 384, 4), ("small", 768, 12), ("medium", 1024, 24), ("large", 1280, 32)):
        o, per_layer = layer0(d)
        kv = 3 * 448 * d + 1500 * 2 * d
        assert lib.oasr_xcd_offsets_ok_debug(o, per_layer, kv, d, 1500, L, 1) == 1, name
    o, per_layer = layer0(1280)
    assert per_layer * 2 * 32 < 2**31 < per_layer * 2 * 41  # (L = 41 still fits: the last layer ends before its cross key | value weights)
    assert lib.oasr_xcd_offsets_ok_debug(o, per_layer, 3 * 448 * 1280 + 1500 * 2 * 1280, 1280, 1500, 42, 1) == 0  # weights past 2 GiB
    assert lib.oasr_xcd_offsets_ok_debug(o, per_layer, 2**29, 1280, 1500, 4, 1) == 0                               # K/V layer stride past 2 GiB
    shifted = (ctypes.c_int64 * 18)(*[x + 2**30 for x in o])                                                       # decoder not at the arena start
    assert lib.oasr_xcd_offsets_ok_debug(shifted, per_layer, 3 * 448 * 1280 + 1500 * 2 * 1280, 1280, 1500, 32, 1) == 0
    # the PRODUCT's arena keeps the decoder blocks last-first (gradient-completion order): layer 0 sits highest and the layer stride is negative
    for name, d, L in (("small", 768, 12), ("medium", 1024, 24), ("large", 1280, 32)):
        o, per_layer = layer0(d)
        top = (ctypes.c_int64 * 18)(*[x + (L - 1) * per_layer for x in o])
        assert lib.oasr_xcd_offsets_ok_debug(top, -per_layer, 3 * 448 * d + 1500 * 2 * d, d, 1500, L, 1) == 1, name
        assert lib.oasr_xcd_offsets_ok_debug(o, -per_layer, 3 * 448 * d + 1500 * 2 * d, d, 1500, L, 1) == 0, name  # would walk below the arena


def test_chip_wide_step_engine_work_split(native):
    """csrc/decode_wide.hip deals a projection's rows out in runs per workgroup and a run's (row, 512-element K span) units to the workgroup's eight compute
    waves in contiguous runs: over all workgroups and waves the units must tile the [N x K] matrix exactly once, within the unrolled bound of the kernel
    instantiation -- for every projection shape of every model variant (no GPU needed: the host-side twin of the device function)."""
    import ctypes as C
    lib = native.lib()
    buf = (C.c_int * 64)()
    for name, d, H, L in (("tiny", 384, 6, 4), ("base", 512, 8, 6), ("small", 768, 12, 12), ("medium", 1024, 16, 24), ("large", 1280, 20, 32)):
        assert lib.oasr_wide_supports_debug(d, H, 1500, 448, L, 1, 256) == 1, name
        for N, K in ((3 * d, d), (d, d), (4 * d, d), (d, 4 * d)):
            seen = {}
            for wg in range(256):
                for wave in range(1, 9):
                    n = lib.oasr_wide_plan_debug(d, 256, N, K, wg, wave, buf, 32)
                    assert 0 <= n <= 8, (name, N, K, wg, wave, n)
                    for i in range(n):
                        key = (buf[2 * i], buf[2 * i + 1])
                        assert key not in seen, (name, N, K, key, seen[key], (wg, wave))
                        seen[key] = (wg, wave)
            spans = (K // 8 + 63) // 64
            assert len(seen) == N * spans and all(0 <= r < N and 0 <= j < spans for r, j in seen), (name, N, K, len(seen))
    # shapes the engine must decline: more than one sequence, a context past 448 positions, too few workgroups, a width whose rows do not fit a packet
    assert lib.oasr_wide_supports_debug(1024, 16, 1500, 448, 24, 2, 256) == 0
    assert lib.oasr_wide_supports_debug(1024, 16, 1500, 512, 24, 1, 256) == 0
    assert lib.oasr_wide_supports_debug(1024, 16, 1500, 448, 24, 1, 32) == 0
    assert lib.oasr_wide_supports_debug(2048, 32, 1500, 448, 24, 1, 256) == 0
    assert lib.oasr_wide_supports_debug(1024, 16, 2000, 448, 24, 1, 256) == 0  # (an audio context past 8 x 216 keys)
