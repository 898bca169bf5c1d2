"""Host-side mirror of the reference's ``olmoasr/model.py`` for the MI355X-native path.

Same class names, constructor arguments, attribute tree and state_dict keys as the reference
(``LayerNorm`` model.py:14, ``Linear`` :42, ``Conv1d`` :104, ``sinusoids`` :199, ``MultiHeadAttention`` :233,
``ResidualAttentionBlock`` :445, ``AudioEncoder`` :531, ``TextDecoder`` :626, ``OLMoASR`` :778), so checkpoints
round-trip (``load_state_dict`` / ``state_dict``; DDP's ``module.`` prefix is stripped by ``load_model``).

What differs is *where the numbers live and who computes*:
  * every parameter is a view into ONE flat fp32 arena laid out in gradient-ready order (see
    ``oasr_param_info``); ``.grad`` of every parameter is a view into a second flat arena.  That is what the
    fused AdamW and the bucketed RCCL reduction operate on.
  * ``OLMoASR.forward`` / ``embed_audio`` / ``logits`` call the C++ engine in ``liboasr.so`` (hand-written HIP
    kernels).  ``OLMoASR.loss_and_backward`` is the fused training micro-step (forward + CE + backward) that
    replaces ``logits = model(...); loss = F.cross_entropy(...); scaler.scale(loss).backward()``
    (train_timestamps.py:1440-1454) without materialising fp32 logits or a [B,448,448] mask.
There is no CPU fallback: constructing the model without a HIP device raises.
"""
import ctypes as C
import math
from typing import Optional

import numpy as np
import warnings

import torch
from torch import Tensor, nn

from . import _native as N
from .config.model_dims import ModelDimensions

PAD_ID = 51864


def sinusoids(length, channels, max_timescale=10000):
    """Returns sinusoids for positional embedding (reference model.py:199-230, same op sequence so the buffer is
    bit-identical)."""
    assert channels % 2 == 0
    log_timescale_increment = np.log(max_timescale) / (channels // 2 - 1)
    inv_timescales = torch.exp(-log_timescale_increment * torch.arange(channels // 2))
    scaled_time = torch.arange(length)[:, np.newaxis] * inv_timescales[np.newaxis, :]
    return torch.cat([torch.sin(scaled_time), torch.cos(scaled_time)], dim=1)


class _ParamModule(nn.Module):
    """Leaf holder: parameters are attached later as views of the flat arena.

    The engine runs whole stacks (``model.encoder(mel)``, ``model.decoder(tokens, xa[, kv_cache])``, ``OLMoASR.forward`` /
    ``loss_and_backward``): that is the hot path.  The per-module ``forward``s of the reference (``LayerNorm`` model.py:14-39,
    ``Linear`` :42-101, ``Conv1d`` :104-196, ``MultiHeadAttention`` :266-345, ``ResidualAttentionBlock`` :485-528) exist below as
    INFERENCE-ONLY compositions of the same native operators (bf16 MFMA GEMMs with fused bias / GELU / residual epilogues,
    LayerNorm and flash-attention kernels) for code that walks the module tree -- probing one block, feature extraction, a
    layer-wise comparison with a reference checkpoint.  They return tensors without a grad_fn; training goes through the engine."""

    def forward(self, *a, **k):  # pragma: no cover
        raise N.NativeError(f"{type(self).__name__}.forward: the native engine runs whole stacks -- call model.encoder(mel), "
                            "model.decoder(tokens, xa[, kv_cache]), OLMoASR.forward / embed_audio / logits / loss_and_backward")

    def _engine(self):
        owner = self.__dict__.get("_owner")
        owner = owner() if owner is not None else None
        if owner is None:
            raise N.NativeError(f"{type(self).__name__} is not attached to an OLMoASR model")
        return owner


def _rows_bf16(x: Tensor):
    """[..., d] activation -> contiguous bf16 [rows, d] on the device (the engine's activation format)."""
    N.require_gpu(x, "x")
    return x.detach().reshape(-1, x.shape[-1]).to(torch.bfloat16).contiguous()


def _mask_to_native(mask: Optional[Tensor], B: int, Tq: int, Tk: int):
    """The reference passes additive masks: the [n_ctx, n_ctx] causal triangle (eval, model.py:740) or, in training, causal +
    key-padding as [B, n_ctx, n_ctx] (train_timestamps.py:314-315).  The kernels take (causal, kv_len[b]); a mask that is not a
    combination of those two is refused rather than approximated."""
    if mask is None:
        return False, None
    if Tq != Tk:
        raise N.NativeError("MultiHeadAttention.forward: a mask needs self-attention (Tq == Tk)")
    m = mask[..., :Tq, :Tk]
    m = m if m.dim() == 3 else m.unsqueeze(0)
    fin = torch.isfinite(m)
    kv_len = fin[:, -1, :].sum(-1).to(torch.int32)  # the last query row sees every unpadded key
    ar = torch.arange(Tk, device=m.device)
    keys = ar[None, None, :] < kv_len[:, None, None]
    if torch.equal(fin, keys.expand_as(fin)):
        causal = False  # key padding only (or no masking at all)
    elif torch.equal(fin, ((ar[None, None, :] <= ar[None, :, None]) & keys).expand_as(fin)):
        causal = True
    else:
        causal = None
    if causal is None or bool((m[fin] != 0).any()):
        raise N.NativeError("MultiHeadAttention.forward: only key padding and / or the causal triangle (finite entries 0, masked entries "
                            "-inf) map onto the native attention kernels")
    if kv_len.shape[0] == 1 and B > 1:
        kv_len = kv_len.expand(B)
    full = bool((kv_len == Tk).all())
    return causal, None if full else kv_len.contiguous()


class _EngineKV:
    """What the engine-owned KV cache looks like from the reference's side of ``install_kv_cache_hooks``: the ONE value of
    the cache dict.  It quacks like the cached key tensor for the two things callers do with it -- ``.shape[1]`` = positions
    consumed so far (TextDecoder.forward's ``offset``, olmoasr/model.py:716) and ``tensor[source_indices].detach()`` (whisper's
    PyTorchInference.rearrange_kv_cache for beam search) -- while the data stay in the engine's layout."""

    def __init__(self, owner, state):
        self.owner, self.state = owner, state

    @property
    def shape(self):
        return (self.state["B"], self.state["pos"], self.owner.dims.n_text_state)

    def detach(self):
        return self

    def __getitem__(self, idx):
        m = self.owner
        idx = torch.as_tensor(idx, device=self.state["cache"].device, dtype=torch.long).reshape(-1)
        B, L, d = self.state["B"], m.dims.n_text_layer, m.dims.n_text_state
        n_self, n_cross = 3 * B * m.dims.n_text_ctx * d, B * m.dims.n_audio_ctx * 2 * d
        esz = 4 if m._act_dtype == torch.float32 else 2
        flat = self.state["cache"][: (n_self + n_cross) * L * esz].view(m._act_dtype).view(L, n_self + n_cross)
        parts = [flat[:, :n_self].view(L, B, -1).index_select(1, idx).reshape(L, -1),
                 flat[:, n_self:].view(L, B, -1).index_select(1, idx).reshape(L, -1)]
        Bn = idx.numel()
        lib = N.lib()
        cache = torch.empty(lib.oasr_kv_cache_bytes(m._ctx, Bn), dtype=torch.uint8, device=flat.device)
        cache[: (n_self + n_cross) // B * Bn * L * esz].view(m._act_dtype).copy_(torch.cat(parts, 1).reshape(-1))
        cache[-N.KV_TAIL_BYTES:].zero_()  # the one-launch step engines' control words (include/oasr.h, "Cached greedy decoding"): only decode_begin zeroes them
        ws = torch.empty(lib.oasr_decode_step_workspace_bytes(m._ctx, Bn), dtype=torch.uint8, device=flat.device)
        return _EngineKV(m, {"cache": cache, "ws": ws, "B": Bn, "pos": self.state["pos"]})


class _TrainStep(torch.autograd.Function):
    """OLMoASR.forward with a grad_fn: oasr_train_fwd keeps the saved activations in the model's workspace, backward hands
    d(loss)/d(logits) to oasr_train_bwd, which accumulates into the flat gradient arena (= every ``p.grad``).  Gradients do not
    travel through autograd's per-parameter AccumulateGrad nodes, so ``torch.nn.parallel.DistributedDataParallel``'s hooks never
    fire: data parallelism is ``olmoasr_amd.ddp.DistributedDataParallel`` (same constructor call; all-reduces the arena after every
    backward) or, on the fused path, ``olmoasr_amd.ddp.GradReducer``."""

    @staticmethod
    def forward(ctx, anchor, model, mel, tokens, text_len):
        N.require_gpu(mel, "mel")
        N.require_gpu(tokens, "tokens")
        B, S = tokens.shape
        assert mel.shape == (B, model.dims.n_mels, 2 * model.dims.n_audio_ctx), "incorrect audio shape"
        model._sync_for_autograd()
        mel = mel.float().contiguous()
        tokens = tokens.to(torch.int64).contiguous()
        text_len = (torch.full((B,), S, dtype=torch.int32, device=mel.device) if text_len is None else text_len.to(torch.int32).contiguous())
        ws = model._ws(B, S, 1)
        logits = torch.empty(B, S, model._n_rows, device=mel.device, dtype=torch.float32)
        with torch.cuda.device(mel.device):
            N.check(N.lib().oasr_train_fwd(model._ctx, N.ptr(mel), N.ptr(tokens), N.ptr(text_len), B, S, N.ptr(logits), N.ptr(ws), ws.numel(),
                                           N.stream_ptr()), "oasr_train_fwd")
        model._autograd_gen = getattr(model, "_autograd_gen", 0) + 1
        ctx.model, ctx.gen = model, model._autograd_gen
        ctx.save_for_backward(tokens, text_len)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if ctx.gen != model._autograd_gen:
            raise RuntimeError("OLMoASR.backward: the activations of this forward are gone -- the engine keeps ONE forward's activations "
                               "(in the model's workspace) and a later training-mode forward or a previous backward through this graph "
                               "used them; run forward and backward in pairs (gradient accumulation: forward/backward per micro-batch)")
        tokens, text_len = ctx.saved_tensors
        B, S = tokens.shape
        dlogits = dlogits.float().contiguous()
        ws = model._ws(B, S, 1)
        seg = getattr(model, "_autograd_segment_events", None)
        ev = None
        if seg is not None:
            ev = (C.c_void_p * len(seg))(*[e.cuda_event for e in seg])
        with torch.cuda.device(dlogits.device):
            N.check(N.lib().oasr_train_bwd(model._ctx, N.ptr(tokens), N.ptr(text_len), N.ptr(dlogits), B, S, ev, N.ptr(ws), ws.numel(),
                                           N.stream_ptr()), "oasr_train_bwd")
        model._autograd_gen += 1  # consumed
        post = getattr(model, "_autograd_post_backward", None)
        if post is not None:
            post()  # ddp.DistributedDataParallel: bucketed all-reduce of the arena, overlapped through the segment events
        return None, None, None, None, None


class _HookHandle:
    """RemovableHandle stand-in returned by install_kv_cache_hooks (there are no module hooks to remove)."""

    def remove(self):
        pass


class LayerNorm(_ParamModule):
    def __init__(self, n_state: int):
        super().__init__()
        self.normalized_shape = (n_state,)
        self.eps = 1e-5

    def forward(self, x: Tensor) -> Tensor:
        """fp32 statistics on the bf16-rounded input, result in ``x.dtype`` (reference model.py:37-39: ``super().forward(x.float()).type(x.dtype)``)."""
        from . import ops
        y, _, _ = ops.layernorm_fwd(_rows_bf16(x), self.weight.detach(), self.bias.detach())
        return y.view(x.shape).to(x.dtype)


class Linear(_ParamModule):
    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features, self.has_bias = in_features, out_features, bias

    def _apply(self, x: Tensor, act: int = 0, resid: Optional[Tensor] = None) -> Tensor:
        """bf16 MFMA GEMM, fp32 accumulation, bias (+ GELU) (+ residual) in the epilogue -- what autocast(bfloat16) makes of
        ``F.linear(x, self.weight.to(x.dtype), self.bias.to(x.dtype))`` (reference model.py:97-101)."""
        from . import ops
        a = _rows_bf16(x)
        w = ops.cast_bf16(self.weight.detach())
        out = torch.empty(a.shape[0], self.out_features, device=a.device, dtype=torch.bfloat16)
        ops.gemm(a, w, a.shape[0], self.out_features, self.in_features, bias=self.bias.detach() if self.has_bias else None, act=act,
                 resid=_rows_bf16(resid) if resid is not None else None, out=out)
        return out.view(*x.shape[:-1], self.out_features)

    def forward(self, x: Tensor) -> Tensor:
        return self._apply(x).to(x.dtype)


class Conv1d(_ParamModule):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size,), (stride,), (padding,)

    def forward(self, x: Tensor) -> Tensor:
        """[B, C_in, T] -> [B, C_out, T_out] (reference model.py:190-196): the windows are gathered by torch (index plumbing), the
        products run on the native GEMM.  (The engine's own stem fuses window gathering, GELU and the positional embedding.)"""
        from . import ops
        N.require_gpu(x, "x")
        B, Cin, T = x.shape
        k, st, pd = self.kernel_size[0], self.stride[0], self.padding[0]
        xp = torch.nn.functional.pad(x.detach().to(torch.bfloat16), (pd, pd))
        win = xp.unfold(2, k, st)                                    # [B, C_in, T_out, k]
        Tout = win.shape[2]
        a = win.permute(0, 2, 1, 3).reshape(B * Tout, Cin * k).contiguous()
        Kp = (Cin * k + 7) // 8 * 8                                   # operand rows are 16-byte multiples
        if Kp != Cin * k:
            a = torch.nn.functional.pad(a, (0, Kp - Cin * k))
        w = torch.zeros(self.out_channels, Kp, device=x.device, dtype=torch.bfloat16)
        w[:, :Cin * k] = self.weight.detach().reshape(self.out_channels, Cin * k).to(torch.bfloat16)
        out = torch.empty(B * Tout, self.out_channels, device=x.device, dtype=torch.bfloat16)
        ops.gemm(a, w, B * Tout, self.out_channels, Kp, bias=self.bias.detach(), out=out)
        return out.view(B, Tout, self.out_channels).permute(0, 2, 1).to(x.dtype)


class Embedding(_ParamModule):
    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: int):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.padding_idx = num_embeddings, embedding_dim, padding_idx


class MultiHeadAttention(_ParamModule):
    def __init__(self, n_state: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.query = Linear(n_state, n_state)
        self.key = Linear(n_state, n_state, bias=False)
        self.value = Linear(n_state, n_state)
        self.out = Linear(n_state, n_state)

    # ``qk`` on request for calls the reference answers with None (its SDPA branch, model.py:328-340): set on an instance -- or on the class --
    # by code that wants to look at the attention scores of mask-free calls, i.e. the cross-attention that word-level timestamp alignment
    # reads (olmoasr_amd/timing.py; whisper.timing.find_alignment hooks ``block.cross_attn`` and takes ``outs[-1]``).
    return_qk = False

    def forward(self, x: Tensor, xa: Optional[Tensor] = None, mask: Optional[Tensor] = None, kv_cache: Optional[dict] = None,
                verbose: bool = False):
        """(out, qk) of reference model.py:266-345, head_dim 64.  ``qk`` follows the reference: the fp32 pre-softmax score matrix
        [B, H, Tq, Tk] (``qkv_attention``, :347-442: (q * 64^-1/4) @ (k * 64^-1/4)^T + mask) when the call takes the reference's manual
        path -- a 2-D mask, i.e. the eval loop's causal decoder self-attention (:316-327) --, None on its SDPA path (3-D padding masks and
        mask-free calls) unless ``return_qk`` asks for it.  The attention output itself always comes from the flash kernels, which never
        form the matrix; ``qk`` is computed by its own kernel (csrc/scores.hip, ``oasr_attention_scores``) only when it is returned.
        ``kv_cache`` dicts belong to ``TextDecoder.forward`` (the engine owns the cache); a per-module cache is refused."""
        from . import ops
        if kv_cache is not None:
            raise N.NativeError("MultiHeadAttention.forward(kv_cache=...): the KV cache lives in the engine -- drive it through "
                                "model.decoder(tokens, xa, kv_cache=cache) with cache, hooks = model.install_kv_cache_hooks()")
        B, Tq, d = x.shape
        src = x if xa is None else xa
        Tk = src.shape[1]
        H = self.n_head
        assert d == H * 64, "the native attention kernels are built for head_dim 64"
        causal, kv_len = _mask_to_native(mask, B, Tq, Tk)
        q = self.query._apply(x).view(B, Tq, H, 64)
        k = self.key._apply(src).view(B, Tk, H, 64)
        v = self.value._apply(src).view(B, Tk, H, 64)
        o, _ = ops.attention_fwd(q, k, v, kv_len, causal)
        qk = None
        if (mask is not None and mask.dim() == 2) or self.return_qk:
            qk = ops.attention_scores(q, k, kv_len, causal)
        return self.out._apply(o).to(x.dtype), qk


class _Sequential(nn.Module):
    """Index-addressable container giving the reference's ``mlp.0`` / ``mlp.2`` state_dict keys."""

    def __init__(self, mods: dict):
        super().__init__()
        for k, v in mods.items():
            self.add_module(k, v)

    def __getitem__(self, i):
        return getattr(self, str(i))

    def forward(self, x: Tensor) -> Tensor:
        """Linear -> GELU -> Linear (reference model.py:480-482), GELU fused into the first GEMM's epilogue."""
        return self[2]._apply(self[0]._apply(x, act=1)).to(x.dtype)


class ResidualAttentionBlock(_ParamModule):
    def __init__(self, n_state: int, n_head: int, cross_attention: bool = False):
        super().__init__()
        self.attn = MultiHeadAttention(n_state, n_head)
        self.attn_ln = LayerNorm(n_state)
        self.cross_attn = MultiHeadAttention(n_state, n_head) if cross_attention else None
        self.cross_attn_ln = LayerNorm(n_state) if cross_attention else None
        self.mlp = _Sequential({"0": Linear(n_state, 4 * n_state), "2": Linear(4 * n_state, n_state)})
        self.mlp_ln = LayerNorm(n_state)

    def forward(self, x: Tensor, xa: Optional[Tensor] = None, mask: Optional[Tensor] = None, kv_cache: Optional[dict] = None,
                verbose: bool = False) -> Tensor:
        """Pre-LN residual block (reference model.py:485-528): x + attn(ln(x)); x + cross_attn(ln(x), xa); x + mlp(ln(x)), with the
        residual additions and the GELU in the GEMM epilogues, as in the engine."""
        a, _ = self.attn(self.attn_ln(x), mask=mask, kv_cache=kv_cache)
        x = (x.to(torch.bfloat16) + a.to(torch.bfloat16)).to(x.dtype)
        if self.cross_attn is not None:
            c, _ = self.cross_attn(self.cross_attn_ln(x), xa, kv_cache=kv_cache)
            x = (x.to(torch.bfloat16) + c.to(torch.bfloat16)).to(x.dtype)
        h = self.mlp[0]._apply(self.mlp_ln(x), act=1)
        return self.mlp[2]._apply(h, resid=x).to(x.dtype)


class AudioEncoder(_ParamModule):
    def __init__(self, n_mels: int, n_ctx: int, n_state: int, n_head: int, n_layer: int):
        super().__init__()
        self.conv1 = Conv1d(n_mels, n_state, kernel_size=3, padding=1)
        self.conv2 = Conv1d(n_state, n_state, kernel_size=3, stride=2, padding=1)
        self.register_buffer("positional_embedding", sinusoids(n_ctx, n_state))
        self.blocks = nn.ModuleList([ResidualAttentionBlock(n_state, n_head) for _ in range(n_layer)])
        self.ln_post = LayerNorm(n_state)

    def forward(self, x: Tensor, verbose: bool = False):
        """AudioEncoder.forward (olmoasr/model.py:571-623): mel [B, n_mels, 3000] -> [B, n_audio_ctx, n_state]."""
        return self._engine().embed_audio(x)


class TextDecoder(_ParamModule):
    def __init__(self, n_vocab: int, n_ctx: int, n_state: int, n_head: int, n_layer: int, pad_row: bool = True):
        super().__init__()
        self.token_embedding = Embedding(n_vocab + (1 if pad_row else 0), n_state, padding_idx=51864 if n_vocab == 51864 else 51865)
        self.blocks = nn.ModuleList([ResidualAttentionBlock(n_state, n_head, cross_attention=True) for _ in range(n_layer)])
        self.ln = LayerNorm(n_state)

    def forward(self, x: Tensor, xa: Tensor, kv_cache: Optional[dict] = None, padding_mask: Optional[Tensor] = None,
                verbose: bool = False):
        """TextDecoder.forward (olmoasr/model.py:688-775): fp32 logits [B, n_tokens, rows].  ``kv_cache`` is the dict of
        ``install_kv_cache_hooks``: empty on the first call (all prompt tokens are consumed), afterwards only the new
        tokens are passed (whisper's PyTorchInference.logits feeds ``tokens[:, -1:]``)."""
        m = self._engine()
        if kv_cache is None:
            return m.logits(x, xa, padding_mask)
        entry = next(iter(kv_cache.values()), None)
        if entry is None:
            entry = _EngineKV(m, m.kv_cache_begin(xa))
            kv_cache[self] = entry
        elif not isinstance(entry, _EngineKV):
            raise N.NativeError("kv_cache must be the dict returned by install_kv_cache_hooks() of this model")
        if entry.state["B"] != x.shape[0]:
            raise N.NativeError(f"kv_cache holds {entry.state['B']} sequences, got {x.shape[0]}")
        return torch.stack([m.kv_cache_step(entry.state, x[:, p]) for p in range(x.shape[1])], dim=1)


def _reference_init_(name: str, t: Tensor, gen: Optional[torch.Generator]):
    """Init distributions of the reference: kaiming_normal_(fan_in, relu) on every Linear/Conv1d weight, the token
    embedding (incl. the pad row) and the decoder positional embedding (model.py:81,171,258-264,665-675); biases
    torch's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)); LayerNorm ones/zeros."""
    if "_ln" in name or ".ln." in name or name.endswith("ln.weight") or name.endswith("ln.bias") or "ln_post" in name:
        t.fill_(1.0 if name.endswith("weight") else 0.0)
        return
    if name.endswith(".bias"):
        fan_in = _FAN_IN_OF_BIAS[name]
        bound = 1.0 / math.sqrt(fan_in)
        t.uniform_(-bound, bound, generator=gen)
        return
    fan_in = 1
    for s in t.shape[1:]:
        fan_in *= s
    t.normal_(0.0, math.sqrt(2.0 / fan_in), generator=gen)


_FAN_IN_OF_BIAS = {}


class OLMoASR(nn.Module):
    """MI355X-native ``olmoasr.model.OLMoASR`` (reference model.py:778-968)."""

    def __init__(self, dims: ModelDimensions, device=None, seed: Optional[int] = None, inference: bool = False,
                 compute_dtype="bfloat16"):
        """``inference=True`` gives the layout of the reference's ``olmoasr.inf_model.OLMoASR`` (token embedding with
        n_vocab rows, no pad row: inf_model.py:302), i.e. what ``load_model(..., inference=True)`` builds.

        ``compute_dtype`` is the reference's ``--precision`` (train_timestamps.py:2128): "bfloat16" = the production
        kernels (autocast(bfloat16) numerics), "float32" = the fp32 validation kernels on the same engine schedule (every
        activation, operand and accumulation in fp32) -- the mode in which logits match the fp32 reference to 1e-3."""
        super().__init__()
        lib = N.lib()
        self.inference = inference
        cdt = {"bfloat16": 0, "bf16": 0, torch.bfloat16: 0, "float32": 1, "fp32": 1, torch.float32: 1}.get(compute_dtype)
        if cdt is None:
            raise N.NativeError(f"compute_dtype {compute_dtype!r}: the native engine computes in 'bfloat16' or 'float32' "
                                "(the reference's float16 autocast has no MI355X-native counterpart here; see DESIGN.md section 4)")
        self.compute_dtype = "float32" if cdt else "bfloat16"
        self._act_dtype = torch.float32 if cdt else torch.bfloat16
        if device is None:
            device = "cuda"
        device = torch.device(device)
        if device.type != "cuda" or not torch.cuda.is_available():
            raise N.NativeError("olmoasr_amd.model.OLMoASR needs a HIP device (MI355X); there is no CPU fallback")
        self.dims = dims
        self.encoder = AudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer)
        self.decoder = TextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head, dims.n_text_layer,
                                   pad_row=not inference)
        cd = N.Dims(*[getattr(dims, f[0]) for f in N.Dims._fields_])
        self._n_rows = dims.n_vocab + (0 if inference else 1)
        self._ctx = lib.oasr_create_ex2(C.byref(cd), self._n_rows, cdt)
        if not self._ctx:
            raise N.NativeError("oasr_create: " + lib.oasr_last_error().decode())
        self._numel = lib.oasr_param_numel(self._ctx)
        self._table = []
        for i in range(lib.oasr_param_count(self._ctx)):
            name = C.create_string_buffer(128)
            off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int()
            shape = (C.c_int64 * 4)()
            N.check(lib.oasr_param_info(self._ctx, i, name, 128, C.byref(off), C.byref(numel), C.byref(ndim), shape), "param_info")
            self._table.append((name.value.decode(), off.value, numel.value, tuple(shape[j] for j in range(ndim.value))))
        self._segments = []
        for i in range(lib.oasr_segment_count(self._ctx)):
            o, m = C.c_int64(), C.c_int64()
            N.check(lib.oasr_segment_info(self._ctx, i, C.byref(o), C.byref(m)), "segment_info")
            self._segments.append((o.value, m.value))
        # ---- flat arenas -----------------------------------------------------------------------------------
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        flat = torch.empty(self._numel, dtype=torch.float32)
        d = dims.n_audio_state
        for name, off, numel, shape in self._table:
            if name.endswith(".bias"):
                if "conv1" in name:
                    _FAN_IN_OF_BIAS[name] = dims.n_mels * 3
                elif "conv2" in name:
                    _FAN_IN_OF_BIAS[name] = d * 3
                elif "mlp.2" in name:
                    _FAN_IN_OF_BIAS[name] = 4 * d
                else:
                    _FAN_IN_OF_BIAS[name] = d
            _reference_init_(name, flat[off:off + numel].view(shape), gen)
        self._flat = flat.to(device)
        self._gflat = None
        self._shadow = torch.zeros(lib.oasr_shadow_bytes(self._ctx), dtype=torch.uint8, device=device)
        self._workspace = None
        import weakref
        for sub in (self.encoder, self.decoder):  # stack-level forward()s call back into the engine (no module cycle)
            sub.__dict__["_owner"] = weakref.ref(self)
        self._attach_views()
        self.to(device)  # moves the sinusoid buffer; parameters are already there (see _apply)
        self._bind()
        self.refresh_shadow()

    # ---- arena plumbing --------------------------------------------------------------------------------------
    def _module_and_attr(self, name):
        parts = name.split(".")
        mod = self
        for p in parts[:-1]:
            mod = getattr(mod, p) if not p.isdigit() else mod[int(p)] if isinstance(mod, nn.ModuleList) else getattr(mod, p)
        return mod, parts[-1]

    def _attach_views(self):
        for name, off, numel, shape in self._table:
            mod, attr = self._module_and_attr(name)
            view = self._flat[off:off + numel].view(shape)
            if attr in mod._parameters and mod._parameters[attr] is not None:
                mod._parameters[attr].data = view
            else:
                mod.register_parameter(attr, nn.Parameter(view))
        if self._gflat is not None:
            for name, off, numel, shape in self._table:
                mod, attr = self._module_and_attr(name)
                mod._parameters[attr].grad = self._gflat[off:off + numel].view(shape)
        self._param_views = [(self._module_and_attr(name)[0]._parameters[self._module_and_attr(name)[1]], off, numel, shape)
                             for name, off, numel, shape in self._table]

    def _apply(self, fn, recurse=True):
        """.to()/.cuda()/.float(): move the flat arenas, then re-point every parameter view (nn.Module._apply would
        otherwise give each parameter private storage and break the arena)."""
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32 or new_flat.device.type != "cuda":
            raise N.NativeError("the native model keeps fp32 master weights on a HIP device")
        self._flat = new_flat.contiguous()
        if self._gflat is not None:
            self._gflat = fn(self._gflat).contiguous()
        self._shadow = self._shadow.to(self._flat.device)
        self._workspace = None
        if getattr(self, "_opt_state", None) is not None:  # optimizer arenas follow the parameters
            self._opt_state = tuple(fn(t).contiguous() for t in self._opt_state)
            self._opt_stats = self._opt_stats.to(self._flat.device)
            self._opt_scratch = self._opt_scratch.to(self._flat.device)
        for m in self.modules():  # buffers only
            for k, b in m._buffers.items():
                if b is not None:
                    m._buffers[k] = fn(b)
        self._attach_views()
        if hasattr(self, "_ctx"):
            self._bind()
        return self

    def _bind(self):
        lib = N.lib()
        pos = self.encoder.positional_embedding
        if pos.dtype != torch.float32 or not pos.is_contiguous():
            self.encoder._buffers["positional_embedding"] = pos.float().contiguous()
            pos = self.encoder.positional_embedding
        opt = getattr(self, "_opt_state", None)
        N.check(lib.oasr_bind(self._ctx, N.ptr(self._flat), N.ptr(self._gflat), N.ptr(opt[0]) if opt else None,
                              N.ptr(opt[1]) if opt else None, N.ptr(pos)), "oasr_bind")
        N.check(lib.oasr_bind_shadow(self._ctx, N.ptr(self._shadow)), "oasr_bind_shadow")

    def refresh_shadow(self):
        """Re-derive the bf16 compute copies after the fp32 parameters changed outside ``optim_step``."""
        with torch.cuda.device(self._flat.device):
            N.check(N.lib().oasr_refresh_shadow(self._ctx, N.stream_ptr()), "oasr_refresh_shadow")

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        res = super().load_state_dict(sd, strict=strict, assign=False)
        self._bind()
        self.refresh_shadow()
        return res

    def enable_grad_arena(self):
        if self._gflat is None:
            self._gflat = torch.zeros_like(self._flat)
            self._attach_views()
            self._bind()
        return self._gflat

    @property
    def flat_params(self) -> Tensor:
        return self._flat

    @property
    def flat_grads(self) -> Tensor:
        return self.enable_grad_arena()

    @property
    def grad_segments(self):
        """[(offset, numel)] arena ranges in the order their gradients become final during backward."""
        return list(self._segments)

    def _ws(self, B, S, mode):
        need = N.lib().oasr_workspace_bytes(self._ctx, B, S, mode)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
        return self._workspace

    # ---- reference API ---------------------------------------------------------------------------------------
    @staticmethod
    def _text_len_from_mask(padding_mask: Tensor) -> Tensor:
        """The reference's padding mask is column-only (train_timestamps.py:314-315): zeros with [:, len:] = -inf.
        Recover len[b] = number of finite columns in row 0."""
        return torch.isfinite(padding_mask[:, 0, :]).sum(-1).to(torch.int32)

    def _forward_impl(self, mel, tokens, text_len, want_logits=True, want_xa=False):
        N.require_gpu(mel, "mel")
        N.require_gpu(tokens, "tokens")
        B, S = tokens.shape
        assert mel.shape == (B, self.dims.n_mels, 2 * self.dims.n_audio_ctx), "incorrect audio shape"
        mel = mel.float().contiguous()
        tokens = tokens.to(torch.int64).contiguous()
        ws = self._ws(B, S, 0)
        logits = torch.empty(B, S, self._n_rows, device=mel.device, dtype=torch.float32) if want_logits else None
        xa = torch.empty(B, self.dims.n_audio_ctx, self.dims.n_audio_state, device=mel.device, dtype=self._act_dtype) if want_xa else None
        with torch.cuda.device(mel.device):
            N.check(N.lib().oasr_forward(self._ctx, N.ptr(mel), N.ptr(tokens), N.ptr(text_len), B, S, N.ptr(logits), N.ptr(xa),
                                         N.ptr(ws), ws.numel(), N.stream_ptr()), "oasr_forward")
        return logits, xa

    def forward(self, mel: Tensor, tokens: Tensor, padding_mask: Optional[Tensor] = None, verbose: bool = False) -> Tensor:
        """logits fp32 [B, S, n_vocab+1] (reference model.py:856-887).  ``padding_mask`` is the reference's
        [B,S,S] additive mask (or an int32 [B] text_len tensor).

        With autograd enabled on a module in training mode (``model.train()``, the nn.Module default) the logits carry a grad_fn, as
        the reference's do: the caller's own loss and ``.backward()`` (train_timestamps.py:1440-1454 unchanged, GradScaler
        included) run the engine's backward from d(loss)/d(logits) and ACCUMULATE into ``p.grad`` of every parameter (views of the
        flat gradient arena).  ``loss_and_backward`` is the fused form of the same step (cross-entropy inside, no fp32 logits)."""
        text_len = None
        if padding_mask is not None:
            text_len = padding_mask.to(torch.int32) if padding_mask.dim() == 1 else self._text_len_from_mask(padding_mask)
            text_len = text_len.to(mel.device).contiguous()
        if torch.is_grad_enabled() and self.training and not self.inference:
            return _TrainStep.apply(self._autograd_anchor(), self, mel, tokens, text_len)
        with torch.no_grad():
            return self._forward_impl(mel, tokens, text_len)[0]

    # ---- torch.autograd bridge -------------------------------------------------------------------------------------------
    def _autograd_anchor(self) -> Tensor:
        a = getattr(self, "_anchor", None)
        if a is None or a.device != self._flat.device:
            a = self._anchor = torch.zeros((), device=self._flat.device, requires_grad=True)
        return a

    def _sync_for_autograd(self):
        """What a torch training loop may have done to the parameters since the last engine call: ``optimizer.step()`` wrote the fp32
        masters in place (-> refresh the bf16 compute copies), ``zero_grad(set_to_none=True)`` dropped ``p.grad`` (-> those gradients
        are reset: zero their arena ranges and re-attach the views)."""
        self.enable_grad_arena()
        ver, dropped = 0, False
        for p, off, numel, shape in self._param_views:
            ver += p._version
            if p.grad is None:
                dropped = True
        if dropped:
            if all(p.grad is None for p, *_ in self._param_views):
                self.zero_grad()
            for p, off, numel, shape in self._param_views:
                if p.grad is None:
                    g = self._gflat[off:off + numel].view(shape)
                    g.zero_()
                    p.grad = g
        if ver != getattr(self, "_param_version", None):
            if getattr(self, "_param_version", None) is not None:
                self.refresh_shadow()
            self._param_version = ver

    @torch.no_grad()
    def embed_audio(self, mel: Tensor) -> Tensor:
        """AudioEncoder forward (reference model.py:815): [B, n_audio_ctx, n_audio_state] in the compute dtype."""
        N.require_gpu(mel, "mel")
        B = mel.shape[0]
        assert mel.shape[1:] == (self.dims.n_mels, 2 * self.dims.n_audio_ctx), "incorrect audio shape"
        mel = mel.float().contiguous()
        ws = self._ws(B, 1, 0)
        xa = torch.empty(B, self.dims.n_audio_ctx, self.dims.n_audio_state, device=mel.device, dtype=self._act_dtype)
        with torch.cuda.device(mel.device):
            N.check(N.lib().oasr_encode(self._ctx, N.ptr(mel), B, N.ptr(xa), N.ptr(ws), ws.numel(), N.stream_ptr()), "oasr_encode")
        return xa

    @torch.no_grad()
    def logits(self, tokens: Tensor, audio_features: Tensor, padding_mask: Tensor = None, last_only: bool = False):
        """TextDecoder forward on given audio features (reference model.py:818-854).  fp32 [B, S, rows], or [B, rows] for
        the last position only (``last_only`` -- the greedy decoding step)."""
        N.require_gpu(tokens, "tokens")
        N.require_gpu(audio_features, "audio_features")
        B, S = tokens.shape
        xa = audio_features.to(self._act_dtype).contiguous()
        assert xa.shape == (B, self.dims.n_audio_ctx, self.dims.n_audio_state)
        tokens = tokens.to(torch.int64).contiguous()
        text_len = None
        if padding_mask is not None:
            text_len = padding_mask.to(torch.int32) if padding_mask.dim() == 1 else self._text_len_from_mask(padding_mask)
            text_len = text_len.to(tokens.device).contiguous()
        ws = self._ws(B, S, 0)
        shape = (B, self._n_rows) if last_only else (B, S, self._n_rows)
        out = torch.empty(*shape, device=tokens.device, dtype=torch.float32)
        with torch.cuda.device(tokens.device):
            N.check(N.lib().oasr_decode_logits(self._ctx, N.ptr(tokens), N.ptr(xa), N.ptr(text_len), B, S, int(last_only), N.ptr(out),
                                               N.ptr(ws), ws.numel(), N.stream_ptr()), "oasr_decode_logits")
        return out

    # ---- cached decoding (the reference's install_kv_cache_hooks + one decoder step per token) ------------------------
    @torch.no_grad()
    def kv_cache_begin(self, audio_features: Tensor):
        """Allocates the KV cache for this batch of windows and fills the cross-attention K/V of every decoder layer."""
        N.require_gpu(audio_features, "audio_features")
        xa = audio_features.to(self._act_dtype).contiguous()
        B = xa.shape[0]
        lib = N.lib()
        cache = torch.empty(lib.oasr_kv_cache_bytes(self._ctx, B), dtype=torch.uint8, device=xa.device)
        ws = torch.empty(lib.oasr_decode_step_workspace_bytes(self._ctx, B), dtype=torch.uint8, device=xa.device)
        with torch.cuda.device(xa.device):
            N.check(lib.oasr_decode_begin(self._ctx, N.ptr(xa), B, N.ptr(cache), N.stream_ptr()), "oasr_decode_begin")
        return {"cache": cache, "ws": ws, "B": B, "pos": 0}

    @torch.no_grad()
    def kv_cache_step(self, state, tokens_last: Tensor) -> Tensor:
        """Feeds the token at position state['pos'] of every sequence; returns fp32 logits [B, rows] for the next one."""
        B = state["B"]
        tokens_last = tokens_last.to(torch.int64).contiguous()
        assert tokens_last.shape == (B,)
        out = torch.empty(B, self._n_rows, device=tokens_last.device, dtype=torch.float32)
        with torch.cuda.device(tokens_last.device):
            N.check(N.lib().oasr_decode_step(self._ctx, N.ptr(tokens_last), B, state["pos"], N.ptr(state["cache"]), N.ptr(out),
                                             N.ptr(state["ws"]), state["ws"].numel(), N.stream_ptr()), "oasr_decode_step")
        state["pos"] += 1
        return out

    @torch.no_grad()
    def kv_cache_reorder(self, state, source_indices) -> None:
        """whisper's ``PyTorchInference.rearrange_kv_cache`` (beam search: row j continues the sequence that was row
        source_indices[j]) on the engine-owned cache, IN PLACE: only the self-attention rows of the positions consumed so far
        move (``pos * 3d`` elements per layer and sequence); the cross-attention K/V stay where they are, so every source must
        belong to the same audio window as its destination -- beams never cross windows."""
        B, L, d, pos = state["B"], self.dims.n_text_layer, self.dims.n_text_state, state["pos"]
        idx = torch.as_tensor(source_indices, device=state["cache"].device, dtype=torch.long).reshape(-1)
        assert idx.numel() == B
        if pos == 0:
            return
        n_self, n_cross = 3 * B * self.dims.n_text_ctx * d, B * self.dims.n_audio_ctx * 2 * d
        esz = 4 if self._act_dtype == torch.float32 else 2
        if bool((idx == torch.arange(B, device=idx.device)).all()):
            return  # every beam continues itself: nothing moves
        flat = state["cache"][: (n_self + n_cross) * L * esz].view(self._act_dtype).view(L, n_self + n_cross)
        # layer by layer, and only the cached k | v columns (the q third of a row is the step's scratch slot): the gathered copy that
        # index_select materialises before the write-back is B * pos * 2d elements at a time, not L times that plus the q slots
        for layer in range(L):
            rows = flat[layer, :n_self].view(B, self.dims.n_text_ctx, 3 * d)[:, :pos, d:]
            rows.copy_(rows.index_select(0, idx))

    def kv_cache_check(self, state) -> bool:
        """Synchronises the stream (oasr_decode_check); call once per decoded window, before reading the tokens back.  False: the
        one-launch step engine could not keep its team resident (a shared or CU-masked device) and the context has switched to the
        multi-launch engine -- the window's tokens are void and the caller decodes it again (``decoding.decode`` does)."""
        with torch.cuda.device(state["cache"].device):
            rc = N.lib().oasr_decode_check(self._ctx, state["B"], N.ptr(state["cache"]), N.stream_ptr())
        if rc == N.ERETRY:
            msg = N.lib().oasr_last_error()
            warnings.warn(f"decode window repeated: {msg.decode() if msg else 'oasr_decode_check asked for a retry'}")
            return False
        N.check(rc, "oasr_decode_check")
        return True

    def install_kv_cache_hooks(self, cache: Optional[dict] = None):
        """olmoasr/model.py:925-964.  The reference hooks every key/value Linear and keeps their outputs in a dict; here the
        engine owns ONE cache buffer (self-attention K/V rows per position, cross-attention K/V once per window), so the
        returned dict starts empty and receives a single entry (an ``_EngineKV``) on the first
        ``model.decoder(tokens, xa, kv_cache=cache)`` call.  Usage is the reference's / whisper's:

            cache, hooks = model.install_kv_cache_hooks()
            logits = model.decoder(prompt_tokens, xa, kv_cache=cache)      # first pass: the whole prompt
            logits = model.decoder(tokens[:, -1:], xa, kv_cache=cache)     # then one token at a time
            for h in hooks: h.remove()
        """
        cache = {**cache} if cache is not None else {}
        return cache, [_HookHandle()]

    def detect_language(self, mel: Tensor, tokenizer=None):
        """whisper.decoding.detect_language as bound at olmoasr/model.py:966."""
        from .decoding import detect_language as _detect
        return _detect(self, mel, tokenizer)

    @torch.no_grad()
    def decode(self, mel: Tensor, options=None, **kwargs):
        """Greedy decoding (temperature 0) of a batch of 30 s windows -- see olmoasr_amd/decoding.py."""
        from .decoding import decode as _decode
        return _decode(self, mel, options, **kwargs)

    @torch.no_grad()
    def transcribe(self, audio, **kwargs):
        """Long-form sliding-window driver -- see olmoasr_amd/transcribe.py."""
        from .transcribe import transcribe as _transcribe
        return _transcribe(self, audio, **kwargs)

    # Word-level timestamp alignment (olmoasr_amd/timing.py) averages the cross-attention of these (layer, head) pairs.  None = whisper's
    # default when a checkpoint names none: every head of the upper half of the decoder (the reference's load_model would call
    # set_alignment_heads for a checkpoint that carries a mask, olmoasr/__init__.py:145, 163-164; none of the published ones does).
    alignment_heads = None

    def set_alignment_heads(self, dump) -> None:
        """``dump``: a bool [n_text_layer, n_text_head] tensor / array, or whisper's serialised form (base85 of a gzip'd bool array)."""
        if isinstance(dump, (bytes, str)):
            import base64
            import gzip
            raw = gzip.decompress(base64.b85decode(dump))
            mask = torch.from_numpy(np.frombuffer(raw, dtype=bool).copy())
        else:
            mask = torch.as_tensor(np.asarray(dump)).to(torch.bool)
        self.alignment_heads = mask.reshape(self.dims.n_text_layer, self.dims.n_text_head).to_sparse()

    @property
    def device(self):
        return self._flat.device

    @property
    def is_multilingual(self):
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self):
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    # ---- fused training micro-step ---------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        g = self.enable_grad_arena()
        with torch.cuda.device(g.device):
            N.check(N.lib().oasr_zero_grad(self._ctx, N.stream_ptr()), "oasr_zero_grad")

    def loss_and_backward(self, mel: Tensor, tokens: Tensor, targets: Tensor, text_len: Tensor, *, loss_scale: float = 1.0,
                          accumulation_steps: int = 1, loss_out: Optional[Tensor] = None, accumulate_loss: bool = False,
                          return_logits: bool = False, segment_events=None, text_ctx: Optional[int] = None, span=None,
                          span_forward: Optional[bool] = None, mel_clip_max: Optional[Tensor] = None):
        """forward + F.cross_entropy(ignore_index=51864)/accumulation_steps + backward of (loss * loss_scale)
        (train_timestamps.py:1440-1454).  Gradients accumulate into ``flat_grads``.  Returns (loss tensor [1], logits|None).

        ``span``: limit the decoder's BACKWARD to the positions that can carry gradient (``oasr_train_fwd_bwd_span``; the forward
        still covers all 448 positions, loss and gradients are those of the plain step up to fp32 summation order).  ``True``:
        derive it here from ``targets`` / ``text_len`` (one small device->host copy); a HOST int sequence / CPU tensor [B]: the
        caller's own bound (the data loader knows the token counts: every target at or past ``span[b]`` must be the ignore
        index and ``span[b] >= text_len[b]``); ``None`` / ``False``: the plain step.  Not combinable with ``return_logits`` / ``text_ctx``.
        ``mel_clip_max`` (with ``span``): ``mel`` is ``ops.log_mel(pcm, finalize=False)``'s un-finalized tensor and this is its per-clip
        maximum [B]; whisper's floor / scale lines are applied while the encoder transposes it (bit-identical input, one pass less).
        ``span_forward`` (with ``span``; default ``None`` = True): the decoder's forward leaves the positions past the span out as well --
        the reference computes their logits (it pads every sample to 448) and nothing reads them (this fused step never returns logits,
        train_timestamps.py:1440-1450 sees them only through ``ignore_index``); loss and gradients are unchanged.  ``False``: the forward
        covers all 448 positions like the reference's (round 4's step).

        ``text_ctx`` (opt-in, not in the reference): run the decoder over the first ``text_ctx`` positions only.  With
        ``text_ctx >= max(text_len)`` the loss and gradients equal the full-context ones (the rest is padding the
        reference computes and then ignores); logits are returned for those positions only."""
        for t, nm in ((mel, "mel"), (tokens, "tokens"), (targets, "targets"), (text_len, "text_len")):
            N.require_gpu(t, nm)
        self.enable_grad_arena()
        B, S = tokens.shape
        assert S == self.dims.n_text_ctx, "training feeds the full padded context (train_timestamps.py:318-329)"
        if text_ctx is not None:
            S = max(1, min(int(text_ctx), S))
            tokens, targets = tokens[:, :S], targets[:, :S]
        mel = mel.float().contiguous()
        tokens = tokens.to(torch.int64).contiguous()
        targets = targets.to(torch.int64).contiguous()
        text_len = text_len.to(torch.int32).contiguous()
        ws = self._ws(B, S, 1)
        if loss_out is None:
            loss_out = torch.zeros(1, device=mel.device, dtype=torch.float32)
        logits = torch.empty(B, S, self.dims.n_vocab + 1, device=mel.device, dtype=torch.float32) if return_logits else None
        ev = None
        if segment_events is not None:
            assert len(segment_events) == len(self._segments)
            handles = [e.cuda_event for e in segment_events]
            if not all(handles):
                raise N.NativeError("segment_events must be recorded-once torch.cuda.Event objects (null HIP event handle)")
            ev = (C.c_void_p * len(segment_events))(*handles)
        if span is not None and span is not False:
            if return_logits or text_ctx is not None:
                raise ValueError("span= cannot be combined with return_logits / text_ctx")
            span_h = self.supervised_span(targets, text_len) if span is True else torch.as_tensor(span, dtype=torch.int32, device="cpu")
            span_h = span_h.to(torch.int32).contiguous()
            assert span_h.numel() == B and not span_h.is_cuda
            if mel_clip_max is not None:
                N.require_gpu(mel_clip_max, "mel_clip_max")
                mel_clip_max = mel_clip_max.float().contiguous()
                assert mel_clip_max.numel() == B
            with torch.cuda.device(mel.device):
                N.check(N.lib().oasr_train_fwd_bwd_span(self._ctx, N.ptr(mel), N.ptr(tokens), N.ptr(targets), N.ptr(text_len),
                                                        C.c_void_p(span_h.data_ptr()), int(span_forward is None or bool(span_forward)), N.ptr(mel_clip_max), B,
                                                        float(loss_scale),
                                                        1.0 / accumulation_steps,
                                                        N.ptr(loss_out), int(accumulate_loss), ev, N.ptr(ws), ws.numel(), N.stream_ptr()),
                        "oasr_train_fwd_bwd_span")
            return loss_out, None
        if mel_clip_max is not None:
            raise ValueError("mel_clip_max needs span= (the un-finalized log-mel is consumed by oasr_train_fwd_bwd_span only)")
        with torch.cuda.device(mel.device):
            N.check(N.lib().oasr_train_fwd_bwd_s(self._ctx, N.ptr(mel), N.ptr(tokens), N.ptr(targets), N.ptr(text_len), B, S,
                                                 float(loss_scale), 1.0 / accumulation_steps, N.ptr(loss_out), int(accumulate_loss),
                                                 N.ptr(logits), ev, N.ptr(ws), ws.numel(), N.stream_ptr()), "oasr_train_fwd_bwd")
        return loss_out, logits

    @staticmethod
    def supervised_span(targets: Tensor, text_len: Tensor, ignore_index: int = 51864) -> Tensor:
        """HOST int32 [B]: per sample, one past the last decoder position that can carry gradient = max(text_len, index of the last
        target != ignore_index + 1) -- what ``loss_and_backward(span=...)`` / ``oasr_train_fwd_bwd_span`` take."""
        S = targets.shape[1]
        pos = torch.arange(1, S + 1, device=targets.device, dtype=torch.int32)
        last = ((targets != ignore_index).to(torch.int32) * pos).amax(dim=1)
        return torch.maximum(last, text_len.to(torch.int32).clamp(max=S)).cpu()

    def init_optimizer_state(self):
        if getattr(self, "_opt_state", None) is None:
            self._opt_state = (torch.zeros_like(self._flat), torch.zeros_like(self._flat))
            self._opt_stats = torch.zeros(2, device=self._flat.device, dtype=torch.float32)
            self._opt_scratch = torch.zeros(8192, device=self._flat.device, dtype=torch.uint8)
            self.enable_grad_arena()
            self._bind()
        return self._opt_state

    # ---- optimizer state in torch.optim.AdamW's own layout (checkpoint compatibility) ------------
    def _param_slices(self):
        """(name, offset, numel, shape) in ``named_parameters()`` order == the reference model's ``parameters()`` order, the
        index space of ``AdamW.state_dict()['state']``."""
        table = {name: (off, numel, shape) for name, off, numel, shape in self._table}
        return [(name,) + table[name] for name, _ in self.named_parameters()]

    def optimizer_state_dict(self, *, step: int, lr: float, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.1,
                             moments=None):
        """What ``torch.optim.AdamW(model.parameters(), ...).state_dict()`` holds after ``step`` steps
        (train_timestamps.py:727-733, saved at :949): the reference can ``optimizer.load_state_dict`` it.  ``moments``:
        full-length (exp_avg, exp_avg_sq) gathered from a sharded optimizer (olmoasr_amd/zero.py) instead of the model's own."""
        m, v = moments if moments is not None else self.init_optimizer_state()
        state = {}
        for i, (_, off, numel, shape) in enumerate(self._param_slices()):
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": m[off:off + numel].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": v[off:off + numel].view(shape).detach().cpu().clone()}
        group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(state)))}
        return {"state": state if step > 0 else {}, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd, into=None) -> int:
        """Inverse of ``optimizer_state_dict`` (also accepts a checkpoint written by the reference).  Returns the step count.
        ``into``: full-length (exp_avg, exp_avg_sq) buffers to fill instead of the model's own (sharded optimizer)."""
        m, v = into if into is not None else self.init_optimizer_state()
        m.zero_()
        v.zero_()
        step = 0
        slices = self._param_slices()
        for i, st in sd.get("state", {}).items():
            _, off, numel, shape = slices[int(i)]
            assert tuple(st["exp_avg"].shape) == tuple(shape), (slices[int(i)][0], st["exp_avg"].shape, shape)
            m[off:off + numel].copy_(st["exp_avg"].reshape(-1).to(m.device, torch.float32))
            v[off:off + numel].copy_(st["exp_avg_sq"].reshape(-1).to(v.device, torch.float32))
            step = max(step, int(float(st["step"])))
        return step

    def optim_step(self, *, step: int, lr: float, inv_loss_scale: float = 1.0, max_grad_norm: float = 1.0, betas=(0.9, 0.98),
                   eps: float = 1e-6, weight_decay: float = 0.1):
        """scaler.unscale_ + clip_grad_norm_ + AdamW.step (train_timestamps.py:1509-1512), fused, plus the bf16 shadow
        refresh.  Returns the device stats tensor [sum g^2 (scaled), found_inf]."""
        self.init_optimizer_state()
        with torch.cuda.device(self._flat.device):
            N.check(N.lib().oasr_optim_step(self._ctx, float(inv_loss_scale), float(max_grad_norm), float(lr), float(betas[0]),
                                            float(betas[1]), float(eps), float(weight_decay), int(step), N.ptr(self._opt_stats),
                                            N.ptr(self._opt_scratch), N.stream_ptr()), "oasr_optim_step")
        return self._opt_stats

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                N.lib().oasr_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass
