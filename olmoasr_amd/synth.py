"""Seeded synthetic training samples in the reference's data layout (token layout of
AudioTextDataset.preprocess_text, scripts/training/train_timestamps.py:238-343):

  audio : int16 PCM [480000] = round(clip(N(0, 0.1), -1, 1) * 32767) with the last U{0..240000} samples zeroed
          (the silence ``pad_or_trim`` appends), consumed as int16/32768 like train_timestamps.py:196
  tokens: [sot 50257, notimestamps 50362, body..., eot 50256], text_input = tokens[:-1], text_y = tokens[1:], both padded
          to 448 with 51864; text_len = len(text_input) (the first -inf column of the reference's padding mask, :314-315)

Sample ``index`` is a pure function of the index (generator seed 1234 + index), so every rank can materialise its own
DistributedSampler shard without communication.
"""
import torch

PAD_ID = 51864
N_SAMPLES = 480000
N_TEXT_CTX = 448


def synth_sample(index: int):
    g = torch.Generator().manual_seed(1234 + index)
    pcm = torch.clamp(torch.randn(N_SAMPLES, generator=g) * 0.1, -1, 1)
    pcm = torch.round(pcm * 32767).to(torch.int16)
    n_sil = int(torch.randint(0, 240001, (1,), generator=g))
    if n_sil:
        pcm[N_SAMPLES - n_sil:] = 0
    L = int(torch.randint(8, 221, (1,), generator=g))
    body = torch.randint(0, 50256, (L - 3,), generator=g)
    tokens = torch.cat([torch.tensor([50257, 50362]), body, torch.tensor([50256])])
    text_input = torch.full((N_TEXT_CTX,), PAD_ID, dtype=torch.long)
    text_y = torch.full((N_TEXT_CTX,), PAD_ID, dtype=torch.long)
    text_input[:L - 1] = tokens[:-1]
    text_y[:L - 1] = tokens[1:]
    return pcm, text_input, text_y, L - 1


def synth_samples(indices, device):
    items = [synth_sample(int(i)) for i in indices]
    pcm = torch.stack([it[0] for it in items]).to(device, non_blocking=True)
    ti = torch.stack([it[1] for it in items]).to(device, non_blocking=True)
    ty = torch.stack([it[2] for it in items]).to(device, non_blocking=True)
    tl = torch.tensor([it[3] for it in items], dtype=torch.int32).to(device, non_blocking=True)
    return pcm, ti, ty, tl
