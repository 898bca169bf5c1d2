"""Generates tests/golden/*.npz from the UNMODIFIED reference (imported from /root/reference in the
build container) -- run as `python -m oracle.gen_golden`.  TEST INFRASTRUCTURE ONLY.

The reference ships no tests/golden vectors of its own, so these fixtures are the pin:
  ref_tiny_b2.npz : reference olmoasr.model.OLMoASR (tiny, CPU fp32 and CPU autocast-bf16), B=2 synthetic
                    batch (oracle.model_oracle.synthetic_batch([0,1])), weights = init_state_dict(seed 0):
                    logits slices, loss, per-tensor grad L2 norms, grad clip coef, post-AdamW(1 step) checksums.
  mel_hf.npz      : transformers.WhisperFeatureExtractor (an implementation independent of ours) on
                    synthetic clips 0 and 1: strided slices + per-band sums of the [80,3000] log-mel.
Fixtures are slices/checksums to stay small.
"""
import os

import numpy as np
import torch

from oracle import mel_oracle as me
from oracle import model_oracle as mo
from oracle import ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
POS = [0, 1, 2, 7, 50, 100, 219, 447]  # sequence positions sampled from the logits
VOC = 512  # first VOC vocabulary entries kept per sampled position


def logits_slice(lg):
    s = lg[:, POS, :]
    return dict(head=s[..., :VOC].numpy(), tail=s[..., -64:].numpy(), argmax=lg.argmax(-1).numpy().astype(np.int32),
                rowmax=lg.max(-1).values.numpy(), lse=torch.logsumexp(lg, -1).numpy())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    # ---- mel ------------------------------------------------------------------------------------
    from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    pcm, ti, ty, tl = mo.synthetic_batch([0, 1])
    wav = pcm.numpy().astype(np.float32) / 32768.0
    hf = np.stack([fe(w, sampling_rate=16000, return_tensors="np").input_features[0] for w in wav])
    np.savez_compressed(os.path.join(OUT, "mel_hf.npz"), first=hf[:, :, :96], last=hf[:, :, -96:],
                        strided=hf[:, :, ::37], band_sum=hf.astype(np.float64).sum(-1), vmax=hf.max((1, 2)))
    # ---- model ----------------------------------------------------------------------------------
    ref_model, _, ref_dims = ref_import.load()
    dims = mo.VARIANTS["tiny"]
    sd = mo.init_state_dict(dims, seed=0)
    net = ref_model.OLMoASR(ref_dims.VARIANT_TO_DIMS["tiny"])
    net.load_state_dict(sd, strict=True)
    mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
    pm = mo.build_padding_mask(tl)
    out = {}
    logits = net(mel, ti, pm)
    loss = torch.nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), ty.view(-1), ignore_index=mo.PAD_ID)
    loss.backward()
    for k, v in logits_slice(logits.detach()).items():
        out["fp32_" + k] = v
    out["fp32_loss"] = np.float32(loss.item())
    names = [k for k, _ in net.named_parameters()]
    out["param_names"] = np.array(names)
    out["fp32_grad_norm"] = np.array([p.grad.double().norm().item() for _, p in net.named_parameters()])
    out["fp32_grad_absmax"] = np.array([p.grad.abs().max().item() for _, p in net.named_parameters()])
    # one full optimizer step exactly as train_timestamps.py:1509-1512 (GradScaler at scale 1 is the identity)
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    out["fp32_total_norm"] = np.float64(total.item())
    opt = torch.optim.AdamW(net.parameters(), lr=1.5e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    opt.step()
    out["fp32_post_sum"] = np.array([p.detach().double().sum().item() for _, p in net.named_parameters()])
    out["fp32_post_abs_sum"] = np.array([p.detach().double().abs().sum().item() for _, p in net.named_parameters()])
    # autocast bf16 forward of the same (pre-step) weights
    net.load_state_dict(sd, strict=True)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        lb = net(mel, ti, pm)
    for k, v in logits_slice(lb.float()).items():
        out["bf16_" + k] = v
    out["bf16_vs_fp32_maxabs"] = np.float32((lb.float() - logits.detach()).abs().max().item())
    out["text_len"] = tl.numpy()
    out["pos"] = np.array(POS)
    np.savez_compressed(os.path.join(OUT, "ref_tiny_b2.npz"), **out)
    for f in os.listdir(OUT):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
