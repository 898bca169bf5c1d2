"""CPU oracle for the OLMoASR training model + loss + optimizer step.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (olmoasr_amd/*) never does and fails loudly without its HIP library.

A plain-PyTorch (CPU, fp32 by default) *functional* restatement of the reference algorithm, driven
by a state_dict with the reference's key names.  Every function cites the reference
lines it follows.  It is independent of /root/reference at run time (that tree does not exist on
the GPU box); oracle/gen_golden.py + tests/test_oracle_model.py pin it against the UNMODIFIED
reference modules imported in the build container (bit-for-bit on CPU fp32) and against committed
fixtures in tests/golden/.

`autocast_bf16=True` mirrors what `torch.autocast(..., dtype=torch.bfloat16)` does to the reference
(train_timestamps.py:1414): matmul-class ops (linear / conv1d / SDPA / the tied-logits matmul) take
bf16 inputs and produce bf16, LayerNorm / softmax-in-CE / GELU internals are fp32, the residual
stream is bf16 (SURVEY.md §2.2 "Residual stream dtype").
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

PAD_ID = 51864  # train_timestamps.py:318-329, model.py:665-667 (padding_idx of the V+1-row embedding)


@dataclass
class Dims:  # olmoasr/config/model_dims.py:4-25
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


def _v(d, h, l):
    return Dims(80, 1500, d, h, l, 51864, 448, d, h, l)


# olmoasr/config/model_dims.py:28-89
VARIANTS = {"tiny": _v(384, 6, 4), "base": _v(512, 8, 6), "small": _v(768, 12, 12),
            "medium": _v(1024, 16, 24), "large": _v(1280, 20, 32)}


def sinusoids(length, channels, max_timescale=10000):
    """model.py:199-230."""
    assert channels % 2 == 0
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def init_state_dict(dims: Dims, seed: int = 0, train_vocab_rows: bool = True):
    """Random state_dict with the reference's names, shapes and init distributions
    (kaiming_normal fan_in/relu on every Linear/Conv1d weight, the embedding incl. pad row and the
    decoder positional embedding -- model.py:81,171,258-264,665-675; biases torch default uniform;
    LayerNorm ones/zeros).  NOT the reference's RNG stream: parity tests always copy one state_dict
    into both sides, so only the distribution matters."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def kaiming(*shape):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(*shape, generator=g) * math.sqrt(2.0 / fan_in)

    def bias(n, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(n, generator=g) * 2 - 1) * b

    def ln(prefix, d):
        sd[prefix + ".weight"] = torch.ones(d)
        sd[prefix + ".bias"] = torch.zeros(d)

    def mha(prefix, d):
        for nm in ("query", "key", "value", "out"):
            sd[f"{prefix}.{nm}.weight"] = kaiming(d, d)
            if nm != "key":  # model.py:259 key has no bias
                sd[f"{prefix}.{nm}.bias"] = bias(d, d)

    def block(prefix, d, cross):
        mha(prefix + ".attn", d)
        ln(prefix + ".attn_ln", d)
        if cross:
            mha(prefix + ".cross_attn", d)
            ln(prefix + ".cross_attn_ln", d)
        sd[prefix + ".mlp.0.weight"] = kaiming(4 * d, d)
        sd[prefix + ".mlp.0.bias"] = bias(4 * d, d)
        sd[prefix + ".mlp.2.weight"] = kaiming(d, 4 * d)
        sd[prefix + ".mlp.2.bias"] = bias(d, 4 * d)
        ln(prefix + ".mlp_ln", d)

    d = dims.n_audio_state
    sd["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, d)
    sd["encoder.conv1.weight"] = kaiming(d, dims.n_mels, 3)
    sd["encoder.conv1.bias"] = bias(d, dims.n_mels * 3)
    sd["encoder.conv2.weight"] = kaiming(d, d, 3)
    sd["encoder.conv2.bias"] = bias(d, d * 3)
    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", d, False)
    ln("encoder.ln_post", d)
    dt = dims.n_text_state
    rows = dims.n_vocab + (1 if train_vocab_rows else 0)
    sd["decoder.token_embedding.weight"] = kaiming(rows, dt)
    sd["decoder.positional_embedding"] = kaiming(dims.n_text_ctx, dt)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", dt, True)
    ln("decoder.ln", dt)
    return sd


class _Cfg:
    def __init__(self, autocast_bf16):
        self.bf16 = autocast_bf16


def _mm_dtype(x, cfg):
    return x.to(torch.bfloat16) if cfg.bf16 else x


def linear(x, w, b, cfg):
    """model.py:97-101 (+ autocast: inputs cast to bf16, output bf16)."""
    if cfg.bf16:
        return F.linear(x.to(torch.bfloat16), w.to(torch.bfloat16), None if b is None else b.to(torch.bfloat16))
    return F.linear(x, w, b)


def layer_norm(x, w, b):
    """model.py:39 -- fp32 internals, cast back to the input dtype; eps = 1e-5."""
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5).to(x.dtype)


def gelu(x):
    """nn.GELU()/F.gelu default = exact erf form (model.py:481,592-593)."""
    return F.gelu(x)


def attention(q, k, v, n_head, mask, cfg):
    """model.py:317-340 SDPA path: softmax(q k^T / sqrt(64) + mask) v, heads split from the last dim.
    `mask` None, [S,S] or [B,S,S] additive float (3-D is unsqueezed to [B,1,S,S], model.py:323).
    Written out explicitly (not F.scaled_dot_product_attention) so the fp32 softmax is unambiguous;
    equals the manual path qkv_attention (model.py:347-442) whose scale is d_head^-0.25 on q and k."""
    B, N, D = q.shape
    M = k.shape[1]
    hd = D // n_head
    qh = q.view(B, N, n_head, hd).permute(0, 2, 1, 3)
    kh = k.view(B, M, n_head, hd).permute(0, 2, 1, 3)
    vh = v.view(B, M, n_head, hd).permute(0, 2, 1, 3)
    s = (qh.float() @ kh.float().transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if mask is not None:
        s = s + (mask if mask.dim() == 2 else mask.unsqueeze(1))
    p = torch.softmax(s, dim=-1)
    if cfg.bf16:
        o = (p.to(torch.bfloat16).float() @ vh.float()).to(torch.bfloat16)
    else:
        o = p @ vh
    return o.permute(0, 2, 1, 3).reshape(B, N, D)


def qkv_attention(q, k, v, n_head, mask=None):
    """MultiHeadAttention.qkv_attention, model.py:347-442 (inf_model.py:172-196): the manual path, returning (wv, qk) with
    qk = (q * d_head^-1/4) @ (k * d_head^-1/4)^T (+ mask), .float(), PRE-softmax -- the tensor ``forward`` hands back as its second
    output (:327, :345) and whisper.timing reads from the cross-attention."""
    B, N, D = q.shape
    M = k.shape[1]
    scale = (D // n_head) ** -0.25
    qh = q.view(B, N, n_head, -1).permute(0, 2, 1, 3) * scale
    kh = k.view(B, M, n_head, -1).permute(0, 2, 3, 1) * scale
    vh = v.view(B, M, n_head, -1).permute(0, 2, 1, 3)
    qk = qh @ kh
    if mask is not None:
        qk = qk + (mask[:N, :M] if mask.dim() == 2 else mask.unsqueeze(1))
    qk = qk.float()
    w = torch.softmax(qk, dim=-1).to(q.dtype)
    return (w @ vh).permute(0, 2, 1, 3).flatten(start_dim=2), qk.detach()


def cross_attention_scores(sd, dims: Dims, tokens, xa, layers, autocast_bf16=False):
    """{layer: qk [B, H, S, n_audio_ctx]} of the decoder's CROSS-attention for a teacher-forced token sequence: the tensors whisper.timing's
    forward hooks on ``block.cross_attn`` collect (``outs[-1]``) when the manual attention path is active -- a walk of TextDecoder.forward
    (model.py:688-775) block by block with qkv_attention() in place of the cross-attention's SDPA call."""
    cfg = _Cfg(autocast_bf16)
    S = tokens.shape[-1]
    x = (sd["decoder.token_embedding.weight"][tokens] + sd["decoder.positional_embedding"][:S]).to(xa.dtype)
    causal = torch.full((dims.n_text_ctx, dims.n_text_ctx), float("-inf")).triu_(1)[:S, :S]
    out = {}
    for i in range(dims.n_text_layer):
        pre = f"decoder.blocks.{i}"
        x = x + mha(sd, pre + ".attn", layer_norm(x, sd[pre + ".attn_ln.weight"], sd[pre + ".attn_ln.bias"]), None, causal, dims.n_text_head, cfg)
        h = layer_norm(x, sd[pre + ".cross_attn_ln.weight"], sd[pre + ".cross_attn_ln.bias"])
        q = linear(h, sd[pre + ".cross_attn.query.weight"], sd[pre + ".cross_attn.query.bias"], cfg)
        k = linear(xa, sd[pre + ".cross_attn.key.weight"], None, cfg)
        v = linear(xa, sd[pre + ".cross_attn.value.weight"], sd[pre + ".cross_attn.value.bias"], cfg)
        wv, qk = qkv_attention(q.float(), k.float(), v.float(), dims.n_text_head)
        if i in layers:
            out[i] = qk
        x = x + linear(wv.to(q.dtype), sd[pre + ".cross_attn.out.weight"], sd[pre + ".cross_attn.out.bias"], cfg)
        h = layer_norm(x, sd[pre + ".mlp_ln.weight"], sd[pre + ".mlp_ln.bias"])
        h = gelu(linear(h, sd[pre + ".mlp.0.weight"], sd[pre + ".mlp.0.bias"], cfg))
        x = x + linear(h, sd[pre + ".mlp.2.weight"], sd[pre + ".mlp.2.bias"], cfg)
    return out


def mha(sd, prefix, x, xa, mask, n_head, cfg):
    """model.py:266-345."""
    q = linear(x, sd[prefix + ".query.weight"], sd[prefix + ".query.bias"], cfg)
    src = x if xa is None else xa
    k = linear(src, sd[prefix + ".key.weight"], None, cfg)
    v = linear(src, sd[prefix + ".value.weight"], sd[prefix + ".value.bias"], cfg)
    wv = attention(q, k, v, n_head, mask, cfg)
    return linear(wv, sd[prefix + ".out.weight"], sd[prefix + ".out.bias"], cfg)


def block(sd, prefix, x, xa, mask, n_head, cfg, cross):
    """ResidualAttentionBlock.forward, model.py:485-528 (pre-LN)."""
    x = x + mha(sd, prefix + ".attn", layer_norm(x, sd[prefix + ".attn_ln.weight"], sd[prefix + ".attn_ln.bias"]),
                None, mask, n_head, cfg)
    if cross:
        x = x + mha(sd, prefix + ".cross_attn",
                    layer_norm(x, sd[prefix + ".cross_attn_ln.weight"], sd[prefix + ".cross_attn_ln.bias"]),
                    xa, None, n_head, cfg)
    h = layer_norm(x, sd[prefix + ".mlp_ln.weight"], sd[prefix + ".mlp_ln.bias"])
    h = linear(h, sd[prefix + ".mlp.0.weight"], sd[prefix + ".mlp.0.bias"], cfg)
    h = gelu(h)
    h = linear(h, sd[prefix + ".mlp.2.weight"], sd[prefix + ".mlp.2.bias"], cfg)
    return x + h


def conv1d(x, w, b, stride, cfg):
    """model.py:104-195 Conv1d (k=3, pad=1)."""
    if cfg.bf16:
        return F.conv1d(x.to(torch.bfloat16), w.to(torch.bfloat16), b.to(torch.bfloat16), stride=stride, padding=1)
    return F.conv1d(x, w, b, stride=stride, padding=1)


def encoder_forward(sd, dims: Dims, mel, autocast_bf16=False, taps=None):
    """AudioEncoder.forward, model.py:571-623."""
    cfg = _Cfg(autocast_bf16)
    x = gelu(conv1d(mel, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], 1, cfg))
    if taps is not None:
        taps["conv1"] = x
    x = gelu(conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], 2, cfg))
    x = x.permute(0, 2, 1)
    assert x.shape[1:] == sd["encoder.positional_embedding"].shape, "incorrect audio shape"
    x = (x + sd["encoder.positional_embedding"]).to(x.dtype)  # model.py:602
    if taps is not None:
        taps["enc_in"] = x
    for i in range(dims.n_audio_layer):
        x = block(sd, f"encoder.blocks.{i}", x, None, None, dims.n_audio_head, cfg, False)
        if taps is not None:
            taps[f"enc_block{i}"] = x
    x = layer_norm(x, sd["encoder.ln_post.weight"], sd["encoder.ln_post.bias"])
    if taps is not None:
        taps["xa"] = x
    return x


def build_padding_mask(text_len, n_ctx=448):
    """train_timestamps.py:314-315: zeros [S,S] with columns >= len set to -inf (column-only)."""
    B = len(text_len)
    m = torch.zeros(B, n_ctx, n_ctx)
    for b, L in enumerate(text_len):
        m[b, :, int(L):] = float("-inf")
    return m


def decoder_forward(sd, dims: Dims, tokens, xa, padding_mask=None, autocast_bf16=False, taps=None):
    """TextDecoder.forward without kv_cache, model.py:688-775.  Returns fp32 logits [B,S,rows]."""
    cfg = _Cfg(autocast_bf16)
    S = tokens.shape[-1]
    x = sd["decoder.token_embedding.weight"][tokens] + sd["decoder.positional_embedding"][:S]
    x = x.to(xa.dtype)  # model.py:732
    causal = torch.full((dims.n_text_ctx, dims.n_text_ctx), float("-inf")).triu_(1)  # model.py:685
    if padding_mask is not None:
        full_mask = padding_mask + causal  # model.py:740-741
    else:
        full_mask = causal[:S, :S]
    if taps is not None:
        taps["dec_in"] = x
    for i in range(dims.n_text_layer):
        x = block(sd, f"decoder.blocks.{i}", x, xa, full_mask, dims.n_text_head, cfg, True)
        if taps is not None:
            taps[f"dec_block{i}"] = x
    x = layer_norm(x, sd["decoder.ln.weight"], sd["decoder.ln.bias"])
    w = sd["decoder.token_embedding.weight"]
    if cfg.bf16:
        logits = (x.to(torch.bfloat16) @ w.to(torch.bfloat16).t()).float()  # model.py:768-770
    else:
        logits = (x @ w.t()).float()
    return logits


def forward(sd, dims: Dims, mel, tokens, padding_mask=None, autocast_bf16=False, taps=None):
    """OLMoASR.forward, model.py:856-887."""
    xa = encoder_forward(sd, dims, mel, autocast_bf16, taps)
    return decoder_forward(sd, dims, tokens, xa, padding_mask, autocast_bf16, taps)


def loss_fn(logits, targets, accumulation_steps=1):
    """train_timestamps.py:1444-1450."""
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), targets.view(-1), ignore_index=PAD_ID) / accumulation_steps


def loss_and_grads(sd, dims, mel, tokens, targets, text_len, autocast_bf16=False, loss_scale=1.0, accumulation_steps=1):
    """One micro-step: forward + CE + backward (scaler.scale(loss).backward(), train_timestamps.py:1454).
    Returns (loss, {name: grad}) with grads of the SCALED loss for every trainable tensor."""
    names = [k for k in sd if k != "encoder.positional_embedding"]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    leaves["encoder.positional_embedding"] = sd["encoder.positional_embedding"]
    pm = build_padding_mask(text_len, dims.n_text_ctx)
    logits = forward(leaves, dims, mel, tokens, pm, autocast_bf16)
    loss = loss_fn(logits, targets, accumulation_steps)
    (loss * loss_scale).backward()
    return loss.detach(), {k: leaves[k].grad for k in names}, logits.detach()


def clip_coef(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (train_timestamps.py:1510): total L2 norm, coef = min(1, max/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def adamw_step(params, grads, m, v, step, lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1):
    """torch.optim.AdamW semantics (train_timestamps.py:727-733; one param group, decoupled decay on
    every tensor).  `step` is 1-based.  In place on params/m/v dicts."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k in grads:
        p, g = params[k], grads[k]
        p.mul_(1 - lr * weight_decay)
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / bc1)


def lr_lambda(global_step, train_steps):
    """prepare_sched, train_timestamps.py:764-781."""
    warmup = math.ceil(0.002 * train_steps)
    if global_step < warmup:
        return float(global_step) / float(max(1, warmup))
    return max(0.0, float(train_steps - global_step) / float(max(1, train_steps - warmup)))


def accumulation_steps(eff_batch_size, world_size, train_batch_size):
    """prepare_sched, train_timestamps.py:764-770."""
    if eff_batch_size <= world_size * train_batch_size:
        return 1
    return eff_batch_size // (world_size * train_batch_size)


@torch.no_grad()
def greedy_decode(sd, dims, mel, initial_tokens, max_new=224, eot=50256, suppress=None, autocast_bf16=False):
    """Cache-less greedy argmax loop in the style of the reference's notebooks/ow_decoding.py:42-72
    (full re-forward each step), the only reference-internal restatement of plain greedy decoding.
    Rows that have emitted `eot` keep emitting it (whisper GreedyDecoder semantics)."""
    xa = encoder_forward(sd, dims, mel, autocast_bf16)
    B = mel.shape[0]
    toks = torch.tensor(initial_tokens, dtype=torch.long).repeat(B, 1)
    done = torch.zeros(B, dtype=torch.bool)
    n_vocab = dims.n_vocab
    for _ in range(max_new):
        logits = decoder_forward(sd, dims, toks, xa, None, autocast_bf16)[:, -1, :n_vocab]
        if suppress is not None:
            logits = logits + suppress
        nxt = logits.argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, eot), nxt)
        toks = torch.cat([toks, nxt[:, None]], dim=1)
        done |= nxt == eot
        if bool(done.all()):
            break
    return toks


# ---- synthetic batch generator, SURVEY.md §8(d) ------------------------------------------------------
SOT, EOT, NO_TIMESTAMPS, TIMESTAMP_BEGIN = 50257, 50256, 50362, 50363  # English-only GPT-2 specials


def timestamp_token(ms: int):
    """AudioTextDataset._convert_to_token_idx, train_timestamps.py:218-236: None past 30 s, else begin + ms // 20."""
    if ms > 30000:
        return None
    return TIMESTAMP_BEGIN + (ms // 20)


def build_token_sequence(segments, norm_end_ms: int, timestamp_mode: bool):
    """Token layout of one training sample from its transcript segments [(start_ms, end_ms, [text token ids])]:
    no-timestamp mode  <sot> <notimestamps> text... <eot>                      (train_timestamps.py:421-426)
    timestamp mode     <sot> <ts s0> text0 <ts e0> <ts s1> text1 <ts e1> ... <ts norm_end> <eot>
                       (_build_timestamp_sequence, train_timestamps.py:462-506; None -> fall back to no-timestamp
                       mode when a boundary lies past 30 s, :437-452)."""
    if timestamp_mode:
        rng = []
        for s, e, _ in segments:
            a, b = timestamp_token(s), timestamp_token(e)
            if a is None or b is None:
                rng = None
                break
            rng.append((a, b))
        if rng is not None:
            out = []
            for i, ((a, b), (_, _, text)) in enumerate(zip(rng, segments)):
                out.extend(([SOT] if i == 0 else []) + [a] + list(text) + [b])
            nxt = TIMESTAMP_BEGIN + (30000 // 20 if norm_end_ms > 30000 else norm_end_ms // 20)
            out.extend([nxt, EOT])
            return out, True
    flat = [t for _, _, text in segments for t in text]
    return [SOT, NO_TIMESTAMPS] + flat + [EOT], False


def pad_sample(tokens, n_text_ctx=448):
    """text_input = tokens[:-1], text_y = tokens[1:], both padded with 51864; text_len = len(text_input) = the first
    -inf column of the padding mask (train_timestamps.py:301-329)."""
    L = len(tokens)
    text_input = torch.full((n_text_ctx,), PAD_ID, dtype=torch.long)
    text_y = torch.full((n_text_ctx,), PAD_ID, dtype=torch.long)
    text_input[:L - 1] = torch.tensor(tokens[:-1], dtype=torch.long)
    text_y[:L - 1] = torch.tensor(tokens[1:], dtype=torch.long)
    return text_input, text_y, L - 1


NO_SPEECH = 50361  # <|nospeech|> of the English-only vocabulary (tokenizer.no_speech; the scripted tokenizer of the a18 pin uses it too)


def ms_of(timestamp) -> int:
    """olmoasr/utils.py:31-47 (convert_to_milliseconds) for "HH:MM:SS.mmm" strings; ints pass through."""
    if not isinstance(timestamp, str):
        return int(timestamp)
    h, m, s, ms = map(float, timestamp.replace(".", ":").split(":"))
    return int(h * 3600000 + m * 60000 + s * 1000 + ms)


def process_empty_transcript(encode, norm_end: int, only_no_ts_mode, rand):
    """AudioTextDataset._process_empty_transcript, train_timestamps.py:345-393."""
    nxt = TIMESTAMP_BEGIN + (30000 // 20 if norm_end > 30000 else norm_end // 20)
    if norm_end >= 30000:
        return [SOT, NO_TIMESTAMPS, NO_SPEECH, EOT]
    if only_no_ts_mode is True:
        return [SOT, NO_TIMESTAMPS] + list(encode("")) + [EOT]
    if rand() >= 0.5:
        return [SOT, TIMESTAMP_BEGIN] + list(encode("")) + [nxt, nxt, EOT]
    return [SOT, NO_TIMESTAMPS] + list(encode("")) + [EOT]


def process_non_empty_transcript(transcript, encode, norm_end: int, ts_mode, only_no_ts_mode, rand):
    """AudioTextDataset._process_non_empty_transcript, train_timestamps.py:395-460, on top of build_token_sequence():
    transcript = [((start, end), text)] in file order (the items of TranscriptReader's dict)."""
    transcript = list(transcript)
    if norm_end > 30000:
        if len(transcript) > 1:
            transcript = transcript[:-1]
            norm_end = transcript[-1][0][1]  # (a "HH:MM:SS.mmm" STRING from here on, as in the reference)
        only_no_ts_mode = True
    segs = [(ms_of(a), ms_of(b), list(encode(" " + text.strip()))) for (a, b), text in transcript]
    want_ts = False
    if only_no_ts_mode is not True:
        if rand() >= 0.5:
            want_ts = ts_mode is True
    tokens, timestamp_mode = build_token_sequence(segs, norm_end if want_ts else 0, want_ts)
    return tokens, timestamp_mode, norm_end


def preprocess_text(transcript, encode, norm_end, ts_mode, only_no_ts_mode, rand, n_text_ctx=448):
    """AudioTextDataset.preprocess_text, train_timestamps.py:238-343, after the transcript has been parsed: returns
    (text_input [448], text_y [448], text_len = first -inf column of the padding mask, timestamp_mode, norm_end)."""
    timestamp_mode = False
    norm_end = ms_of(norm_end)
    if not transcript:
        tokens = process_empty_transcript(encode, norm_end, only_no_ts_mode, rand)
        if only_no_ts_mode is False and norm_end < 30000:
            if rand() >= 0.5:
                timestamp_mode = True
    else:
        tokens, timestamp_mode, norm_end = process_non_empty_transcript(transcript, encode, norm_end, ts_mode, only_no_ts_mode, rand)
    text_input, text_y, text_len = pad_sample(tokens, n_text_ctx)
    return text_input, text_y, text_len, timestamp_mode, norm_end


def sched_rule(train_steps: int, world_size: int, train_batch_size: int, eff_batch_size: int):
    """prepare_sched, train_timestamps.py:739-783: (accumulation_steps, warmup_steps, lr factor as a function of the step)."""
    import math
    accum = 1 if eff_batch_size <= world_size * train_batch_size else eff_batch_size // (world_size * train_batch_size)
    warmup = math.ceil(0.002 * train_steps)

    def factor(step: int) -> float:
        if step < warmup:
            return float(step) / float(max(1, warmup))
        return max(0.0, float(train_steps - step) / float(max(1, train_steps - warmup)))
    return accum, warmup, factor


def synthetic_sample(index: int, n_text_ctx=448, timestamps: bool = False):
    """Deterministic (audio int16 [480000], text_input i64 [448], text_y i64 [448], text_len).
    timestamps=True: the same audio and text body cut into 1-4 transcript segments whose boundaries are multiples of
    20 ms inside the non-silent part of the clip, laid out in the reference's timestamp mode."""
    g = torch.Generator().manual_seed(1234 + index)
    pcm = torch.clamp(torch.randn(480000, generator=g) * 0.1, -1, 1)
    pcm = torch.round(pcm * 32767).to(torch.int16)
    n_sil = int(torch.randint(0, 240001, (1,), generator=g))
    if n_sil:
        pcm[480000 - n_sil:] = 0
    L = int(torch.randint(8, 221, (1,), generator=g))
    body = torch.randint(0, 50256, (L - 3,), generator=g)
    if not timestamps:
        tokens, _ = build_token_sequence([(0, 0, body.tolist())], 0, False)
    else:
        norm_end = (480000 - n_sil) // 16 // 20 * 20  # ms of audio before the silence, on the 20 ms grid
        n_seg = int(torch.randint(1, 5, (1,), generator=g))
        n_seg = min(n_seg, L - 3)
        cuts = sorted(int(x) for x in torch.randint(0, norm_end // 20 + 1, (2 * n_seg,), generator=g))
        split = sorted(int(x) for x in torch.randint(0, L - 3 + 1, (n_seg - 1,), generator=g))
        edges = [0] + split + [L - 3]
        segs = [(cuts[2 * i] * 20, cuts[2 * i + 1] * 20, body[edges[i]:edges[i + 1]].tolist()) for i in range(n_seg)]
        tokens, _ = build_token_sequence(segs, norm_end, True)
    text_input, text_y, text_len = pad_sample(tokens, n_text_ctx)
    return pcm, text_input, text_y, text_len


def synthetic_batch(indices, timestamps: bool = False):
    items = [synthetic_sample(i, timestamps=timestamps) for i in indices]
    pcm = torch.stack([it[0] for it in items])
    ti = torch.stack([it[1] for it in items])
    ty = torch.stack([it[2] for it in items])
    tl = torch.tensor([it[3] for it in items], dtype=torch.int32)
    return pcm, ti, ty, tl
