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


class SynthLoader:
    """Iterator over micro-batches of the synthetic dataset with background generation -- the role the reference's
    ``DataLoader(num_workers=..., pin_memory=True, prefetch_factor=...)`` plays (train_timestamps.py:640-660): ``workers``
    threads materialise samples (torch's CPU RNG / rounding kernels release the GIL), ``depth`` batches are kept in flight
    as pinned host tensors, the H2D copies are asynchronous.  ``order`` yields the sample indices of successive batches."""

    def __init__(self, order, device, workers: int = 8, depth: int = 3):
        from concurrent.futures import ThreadPoolExecutor
        self.order = iter(order)
        self.device = device
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.depth = max(1, depth)
        self.pending = []
        self._fill()

    def _submit(self, indices):
        return [self.pool.submit(synth_sample, int(i)) for i in indices]

    def _fill(self):
        while len(self.pending) < self.depth:
            try:
                idx = next(self.order)
            except StopIteration:
                return
            self.pending.append(self._submit(idx))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.pending:
            raise StopIteration
        items = [f.result() for f in self.pending.pop(0)]
        self._fill()
        pin = torch.cuda.is_available()

        def up(t):
            return (t.pin_memory() if pin else t).to(self.device, non_blocking=True)
        pcm = up(torch.stack([it[0] for it in items]))
        ti = up(torch.stack([it[1] for it in items]))
        ty = up(torch.stack([it[2] for it in items]))
        tl = up(torch.tensor([it[3] for it in items], dtype=torch.int32))
        return pcm, ti, ty, tl

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)
