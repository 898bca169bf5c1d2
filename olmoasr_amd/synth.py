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


SOT, EOT, NO_TIMESTAMPS, TIMESTAMP_BEGIN = 50257, 50256, 50362, 50363


def _layout(tokens: torch.Tensor):
    """tokens i64 [L] -> (text_input [448], text_y [448], text_len): the teacher-forcing shift and 51864 padding of
    AudioTextDataset.preprocess_text (train_timestamps.py:301-329)."""
    L = tokens.numel()
    assert L - 1 <= N_TEXT_CTX
    text_input = torch.full((N_TEXT_CTX,), PAD_ID, dtype=torch.long)
    text_y = torch.full((N_TEXT_CTX,), PAD_ID, dtype=torch.long)
    text_input[:L - 1] = tokens[:-1]
    text_y[:L - 1] = tokens[1:]
    return text_input, text_y, L - 1


def synth_sample(index: int, timestamps: bool = False):
    """``timestamps=True`` lays the same text out in the reference's timestamp mode (``ts`` branch of
    _process_non_empty_transcript / _build_timestamp_sequence, train_timestamps.py:401-506):
    <sot> <|s0|> text0 <|e0|> <|s1|> text1 <|e1|> ... <|norm_end|> <eot>, timestamp token = 50363 + ms // 20."""
    g = torch.Generator().manual_seed(1234 + index)
    pcm = torch.clamp(torch.randn(N_SAMPLES, generator=g) * 0.1, -1, 1)
    pcm = torch.round(pcm * 32767).to(torch.int16)
    n_sil = int(torch.randint(0, 240001, (1,), generator=g))
    if n_sil:
        pcm[N_SAMPLES - n_sil:] = 0
    L = int(torch.randint(8, 221, (1,), generator=g))
    body = torch.randint(0, 50256, (L - 3,), generator=g)
    if not timestamps:
        tokens = torch.cat([torch.tensor([SOT, NO_TIMESTAMPS]), body, torch.tensor([EOT])])
        return (pcm,) + _layout(tokens)
    nb = L - 3
    norm_end_ms = (N_SAMPLES - n_sil) // 16 // 20 * 20
    n_seg = min(int(torch.randint(1, 5, (1,), generator=g)), nb)
    stamps = torch.randint(0, norm_end_ms // 20 + 1, (2 * n_seg,), generator=g).sort().values + TIMESTAMP_BEGIN
    split = torch.randint(0, nb + 1, (n_seg - 1,), generator=g).sort().values
    edges = torch.cat([torch.zeros(1, dtype=torch.long), split, torch.tensor([nb])])
    parts = [torch.tensor([SOT])]
    for i in range(n_seg):
        parts += [stamps[2 * i:2 * i + 1], body[int(edges[i]):int(edges[i + 1])], stamps[2 * i + 1:2 * i + 2]]
    parts.append(torch.tensor([TIMESTAMP_BEGIN + norm_end_ms // 20, EOT]))
    return (pcm,) + _layout(torch.cat(parts))


def supervised_span_host(text_y: torch.Tensor, text_len: torch.Tensor) -> torch.Tensor:
    """HOST int32 [B] from HOST token tensors: per sample one past the last position that can carry gradient = max(text_len, index of
    the last target != 51864 + 1) -- the bound ``OLMoASR.loss_and_backward(span=...)`` / ``oasr_train_fwd_bwd_span`` take.  The loaders
    build the token sequences on the host (like AudioTextDataset.preprocess_text, train_timestamps.py:238-343) and hand this out with
    every batch (``loader.last_span``), so the training loop never reads it back from the device."""
    assert not text_y.is_cuda and not text_len.is_cuda
    S = text_y.shape[1]
    pos = torch.arange(1, S + 1, dtype=torch.int32)
    last = ((text_y != PAD_ID).to(torch.int32) * pos).amax(dim=1)
    return torch.maximum(last, text_len.to(torch.int32).clamp(max=S)).contiguous()


def synth_samples(indices, device, timestamps: bool = False):
    items = [synth_sample(int(i), timestamps) for i in indices]
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

    def __init__(self, order, device, workers: int = 8, depth: int = 3, timestamps: bool = False):
        from concurrent.futures import ThreadPoolExecutor
        self.timestamps = timestamps
        self.order = iter(order)
        self.device = device
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.depth = max(1, depth)
        self.pending = []
        self.last_span = None  # HOST int32 [B]: supervised span of the batch the last __next__ returned (supervised_span_host)
        self._fill()

    def _submit(self, indices):
        return [self.pool.submit(synth_sample, int(i), self.timestamps) for i in indices]

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
        ty_h = torch.stack([it[2] for it in items])
        tl_h = torch.tensor([it[3] for it in items], dtype=torch.int32)
        self.last_span = supervised_span_host(ty_h, tl_h)
        pcm = up(torch.stack([it[0] for it in items]))
        ti = up(torch.stack([it[1] for it in items]))
        ty = up(ty_h)
        tl = up(tl_h)
        return pcm, ti, ty, tl

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)
