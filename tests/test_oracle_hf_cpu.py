"""The model oracle and the decoding oracle against an implementation that shares no code with either the reference or this repo:
transformers' WhisperForConditionalGeneration (installed offline) loaded with the SAME weights through the standard OpenAI->HF key map
(what the reference's demo/convert_openai_to_hf.py:104-123 does).  openai-whisper itself is not vendored, so this is the independent
pin for (a) the whole encoder-decoder forward -- on top of the direct pin against the unmodified reference modules,
tests/test_oracle_model.py -- and (b) greedy decoding with the suppress filters: HF forward + HF logits processors + argmax."""
import re

import pytest
import torch


def _to_hf(sd):
    out = {}
    sub = [(r"^encoder\.blocks\.(\d+)\.", r"model.encoder.layers.\1."), (r"^decoder\.blocks\.(\d+)\.", r"model.decoder.layers.\1."),
           (r"\.cross_attn_ln\.", ".encoder_attn_layer_norm."), (r"\.cross_attn\.", ".encoder_attn."), (r"\.attn_ln\.", ".self_attn_layer_norm."),
           (r"\.attn\.", ".self_attn."), (r"\.mlp_ln\.", ".final_layer_norm."), (r"\.mlp\.0\.", ".fc1."), (r"\.mlp\.2\.", ".fc2."),
           (r"\.query\.", ".q_proj."), (r"\.key\.", ".k_proj."), (r"\.value\.", ".v_proj."), (r"\.out\.", ".out_proj."),
           (r"^encoder\.ln_post\.", "model.encoder.layer_norm."), (r"^decoder\.ln\.", "model.decoder.layer_norm."),
           (r"^encoder\.conv", "model.encoder.conv"), (r"^encoder\.positional_embedding$", "model.encoder.embed_positions.weight"),
           (r"^decoder\.positional_embedding$", "model.decoder.embed_positions.weight"),
           (r"^decoder\.token_embedding\.weight$", "model.decoder.embed_tokens.weight")]
    for k, v in sd.items():
        for pat, rep in sub:
            k = re.sub(pat, rep, k)
        out[k] = v.clone()
    out["proj_out.weight"] = out["model.decoder.embed_tokens.weight"]
    return out


@pytest.fixture(scope="module")
def pair():
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from oracle import model_oracle as mo
    torch.manual_seed(0)
    dims = mo.Dims(80, 1500, 128, 2, 2, 51864, 448, 128, 2, 2)
    sd = mo.init_state_dict(dims, seed=4, train_vocab_rows=False)
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * 3.0  # separated candidates
    cfg = WhisperConfig(vocab_size=51864, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=2,
                        decoder_attention_heads=2, encoder_ffn_dim=512, decoder_ffn_dim=512, max_source_positions=1500, max_target_positions=448,
                        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, scale_embedding=False,
                        pad_token_id=50256, bos_token_id=50257, eos_token_id=50256, decoder_start_token_id=50257)
    hf = WhisperForConditionalGeneration(cfg).eval()
    missing, unexpected = hf.load_state_dict(_to_hf(sd), strict=False)
    assert not unexpected and all("proj_out" in m or "embed_positions" in m for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(1)
    mel = torch.randn(2, 80, 3000, generator=g) * 0.5
    return mo, dims, sd, hf, mel


def test_model_oracle_equals_transformers_whisper(pair):
    mo, dims, sd, hf, mel = pair
    g = torch.Generator().manual_seed(2)
    tokens = torch.randint(0, 50000, (2, 12), generator=g)
    tokens[:, 0] = 50257
    with torch.no_grad():
        want = hf(input_features=mel, decoder_input_ids=tokens).logits
        xa_hf = hf.model.encoder(mel).last_hidden_state
    got = mo.forward(sd, dims, mel, tokens)
    xa = mo.encoder_forward(sd, dims, mel)
    assert float((xa - xa_hf).abs().max()) < 2e-4
    assert float((got - want).abs().max()) < 5e-4 * max(1.0, float(want.abs().max())), float((got - want).abs().max())


def test_greedy_decode_oracle_equals_transformers_forward_plus_processors(pair):
    from transformers.generation.logits_process import SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor
    from oracle import decode_oracle as do
    mo, dims, sd, hf, mel = pair
    L = 7
    opt = do.Options(sample_len=L, without_timestamps=True)
    want_oracle = do.decode(sd, dims, mel, opt)
    sup = SuppressTokensLogitsProcessor(do.suppress_list(opt))
    begin = SuppressTokensAtBeginLogitsProcessor([220, 50256], begin_index=2)
    tokens = torch.tensor([[50257, 50362]] * mel.shape[0])
    done = torch.zeros(mel.shape[0], dtype=torch.bool)
    with torch.no_grad():
        for _ in range(L):
            logits = hf(input_features=mel, decoder_input_ids=tokens).logits[:, -1].float()
            logits = sup(tokens, begin(tokens, logits))
            nxt = logits.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, 50256), nxt)
            tokens = torch.cat([tokens, nxt[:, None]], 1)
            done |= nxt == 50256
    for b, r in enumerate(want_oracle):
        body = tokens[b, 2:].tolist()
        body = body[:body.index(50256)] if 50256 in body else body
        assert r.tokens == body, (b, r.tokens, body)


def test_timestamp_mode_greedy_equals_transformers_forward_plus_processors(pair):
    """The reference's default decoding mode (timestamp tokens on): SuppressBlank, SuppressTokens, ApplyTimestampRules in whisper's order,
    here as transformers' three processors around transformers' forward."""
    from types import SimpleNamespace
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    from oracle import decode_oracle as do
    mo, dims, sd, hf, mel = pair
    L = 9
    opt = do.Options(sample_len=L)
    want_oracle = do.decode(sd, dims, mel, opt)
    sup = SuppressTokensLogitsProcessor(do.suppress_list(opt))
    begin = SuppressTokensAtBeginLogitsProcessor([220, 50256], begin_index=1)
    cfg = SimpleNamespace(no_timestamps_token_id=50362, eos_token_id=50256, bos_token_id=50256, max_initial_timestamp_index=50,
                          _detect_timestamp_from_logprob=True)
    ts = WhisperTimeStampLogitsProcessor(cfg, begin_index=1)
    tokens = torch.tensor([[50257]] * mel.shape[0])
    done = torch.zeros(mel.shape[0], dtype=torch.bool)
    with torch.no_grad():
        for _ in range(L):
            logits = hf(input_features=mel, decoder_input_ids=tokens).logits[:, -1].float()
            logits = ts(tokens, sup(tokens, begin(tokens, logits)))
            nxt = logits.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, 50256), nxt)
            tokens = torch.cat([tokens, nxt[:, None]], 1)
            done |= nxt == 50256
    for b, r in enumerate(want_oracle):
        body = tokens[b, 1:].tolist()
        body = body[:body.index(50256)] if 50256 in body else body
        assert r.tokens == body, (b, r.tokens, body)
        assert body[0] >= 50363  # the first sampled token is a timestamp
