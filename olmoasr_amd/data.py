"""Real-data input path of the training loop: the reference's sample shards in, device-resident
int16 PCM + token matrices out, with the log-mel front end left to the GPU kernel (``ops.log_mel``) in the main process.

What the reference does per sample, in DataLoader worker processes (scripts/training/train_timestamps.py):
  * ``open_dicts_file`` (:577-604): ``{samples_dicts_dir}/*.jsonl.{gz,zst}``, one JSON object per line with ``audio_file`` (int16
    ``.npy``), ``subtitle_file``, ``seg_content``, ``ts_mode``, ``only_no_ts_mode``, ``norm_end``;
  * ``AudioTextDataset.preprocess_text`` (:238-343): transcript -> tokens via the whisper tokenizer, the 448 / 51864 padding and the
    column-only padding mask;
  * ``preprocess_audio`` (:175-217): ``np.load(...)/32768``, ``pad_or_trim(norm_end * 16)`` then ``pad_or_trim(480000)``, and
    ``log_mel_spectrogram`` ON THE CPU (7-9 ms per clip per core: >= 13 cores per GPU at the benchmarked step rate);
  * ``DistributedSampler(shuffle, seed=42, drop_last=False)`` (:633-638).
Here: the same shards, the same sampler (torch's own class, so the order is the reference's by construction), the audio kept as
int16 -- loader threads ``np.load`` into PINNED ring slots zero-padded exactly as the pad_or_trim chain would (``valid_samples``),
one asynchronous H2D per micro-batch on a copy stream (0.96 MB per clip instead of the reference's 0.96 MB fp32 mel + 0.8 MB
mask), the mel on the GPU.  The [448, 448] float mask never exists: ``text_len`` (its first -inf column) is what the kernels take.

Text: the whisper tokenizer (tiktoken vocabulary) is not available offline, so the tokenizer is a plug.  ``olmoasr_amd/text_layout.py``
is ``AudioTextDataset.preprocess_text`` itself (WebVTT ``seg_content`` -> the four token layouts, the reference's coin, the > 30 s rules;
pinned bit-exactly against the reference's own file running, tests/test_token_layout_ref_cpu.py): ``text_fn =
text_layout.reference_text_fn(tokenizer)`` with any object that has whisper's tokenizer attributes.  The default ``text_fn`` reads a
pre-tokenised ``"tokens"`` field (what the offline synthetic shards carry).
"""
import glob
import gzip
import io
import json
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .synth import N_SAMPLES, N_TEXT_CTX, PAD_ID, _layout, supervised_span_host


def convert_to_milliseconds(timestamp: str) -> int:
    """``HH:MM:SS.mmm`` -> ms (olmoasr/utils.py:31-47; same ValueError on a malformed stamp)."""
    try:
        h, m, s, ms = map(float, timestamp.replace(".", ":").split(":"))
        return int(h * 3600000 + m * 60000 + s * 1000 + ms)
    except (ValueError, IndexError) as e:
        raise ValueError(f"Invalid timestamp format: {timestamp}") from e


def open_dicts_file(samples_dicts_file: str) -> List[Dict]:
    """train_timestamps.py:577-604 (+ uncompressed ``.jsonl``).  ``.zst`` needs the ``zstandard`` package the reference uses."""
    if samples_dicts_file.endswith(".gz"):
        with gzip.open(samples_dicts_file, "rt") as f:
            return [json.loads(line.strip()) for line in f if line.strip()]
    if samples_dicts_file.endswith(".zst"):
        try:
            import zstandard as zstd
        except ImportError as e:  # not in the offline image
            raise RuntimeError(f"{samples_dicts_file}: reading .zst shards needs the `zstandard` package (recompress as .jsonl.gz)") from e
        out = []
        with open(samples_dicts_file, "rb") as f, zstd.ZstdDecompressor().stream_reader(f) as reader:
            for line in io.TextIOWrapper(reader, encoding="utf-8"):
                try:
                    out.append(json.loads(line))
                except json.JSONDecodeError:
                    break  # reached padding at the end (as the reference)
        return out
    with open(samples_dicts_file, "rt") as f:
        return [json.loads(line) for line in f if line.strip()]


def load_samples_dicts(samples_dicts_dir: str) -> List[Dict]:
    """train_timestamps.py:2255-2266.  Files are read in SORTED order: the reference's ``pool.imap_unordered`` makes the sample order
    (and with it every rank's shard) depend on worker timing; sorting is the reproducible member of that family."""
    files = sorted(set(glob.glob(f"{samples_dicts_dir}/*.jsonl.*") + glob.glob(f"{samples_dicts_dir}/*.jsonl")))
    if not files:
        raise FileNotFoundError(f"no *.jsonl[.gz|.zst] shard files under {samples_dicts_dir!r}")
    out: List[Dict] = []
    for f in files:
        out.extend(open_dicts_file(f))
    return out


def valid_samples(n_loaded: int, norm_end) -> int:
    """How many leading samples of the loaded array survive ``preprocess_audio`` (train_timestamps.py:196-211): with a truthy
    ``norm_end`` the clip is cut at ``norm_end * 16`` samples, then padded / cut to 30 s; everything after is zeros."""
    n = min(int(n_loaded), N_SAMPLES)
    if norm_end:
        if isinstance(norm_end, str):
            norm_end = convert_to_milliseconds(norm_end)
        n = min(n, max(0, int(norm_end) * 16))
    return n


def tokens_field_text_fn(sample: Dict) -> Tuple[Sequence[int], bool, object]:
    """Default ``text_fn``: the shard carries what ``preprocess_text`` computed -- ``tokens`` = [sot, ..., eot] before the shift,
    optionally ``timestamp_mode`` and the possibly corrected ``norm_end`` (``new_norm_end``)."""
    if "tokens" not in sample:
        raise KeyError(f"sample for {sample.get('audio_file')!r} has no pre-tokenised `tokens` field and no tokenizer is available offline: "
                       "pass text_fn=<the reference's preprocess_text> or pre-tokenise the shards (INTEGRATION.md, 'Training data')")
    return sample["tokens"], bool(sample.get("timestamp_mode", False)), sample.get("new_norm_end", sample.get("norm_end"))


class AudioTextShards:
    """``AudioTextDataset`` (train_timestamps.py:84-175) without the CPU mel and without the float mask: ``load(i)`` returns
    (pcm int16 ndarray [n_valid], text_input i64 [448], text_y i64 [448], text_len)."""

    def __init__(self, samples: List[Dict], n_text_ctx: int = N_TEXT_CTX, text_fn: Optional[Callable] = None):
        assert n_text_ctx == N_TEXT_CTX
        self.samples = samples
        self.text_fn = text_fn or tokens_field_text_fn

    def __len__(self):
        return len(self.samples)

    def load(self, index: int):
        s = self.samples[index]
        tokens, timestamp_mode, new_norm_end = self.text_fn(s)
        norm_end = s.get("norm_end")
        if timestamp_mode is True:
            norm_end = None  # full 30 s of audio in timestamp mode (:148-149)
        elif new_norm_end != norm_end:
            norm_end = new_norm_end  # adjusted end time (:150-151)
        tokens = torch.as_tensor(np.asarray(tokens, dtype=np.int64))
        if tokens.numel() - 1 > N_TEXT_CTX:
            raise ValueError(f"{s.get('subtitle_file')}: {tokens.numel() - 1} text tokens exceed n_text_ctx = {N_TEXT_CTX}")
        text_input, text_y, text_len = _layout(tokens)
        arr = np.load(s["audio_file"], mmap_mode="r")
        if arr.dtype != np.int16:
            raise TypeError(f"{s['audio_file']}: expected int16 PCM, got {arr.dtype}")
        return arr[:valid_samples(arr.shape[0], norm_end)], text_input, text_y, text_len


def sampler_indices(n: int, rank: int, world: int, epoch: int, shuffle: bool = True, seed: int = 42) -> List[int]:
    """This rank's sample order for one epoch: torch's DistributedSampler with the reference's arguments (seed=42,
    drop_last=False, train_timestamps.py:633-638) and ``set_epoch(epoch)``."""
    from torch.utils.data.distributed import DistributedSampler

    class _Len:
        def __len__(self):
            return n
    sp = DistributedSampler(_Len(), num_replicas=world, rank=rank, shuffle=shuffle, seed=seed, drop_last=False)
    sp.set_epoch(epoch)
    return list(iter(sp))


def epoch_batches(n: int, rank: int, world: int, batch: int, epoch: int = 0, cursor: int = 0, shuffle: bool = True):
    """Index lists of successive micro-batches, epoch after epoch, starting ``cursor`` samples into ``epoch`` (checkpoint resume):
    what iterating ``DataLoader(dataset, batch_size, sampler=DistributedSampler(...), drop_last=False)`` with
    ``sampler.set_epoch(epoch)`` per epoch yields -- the last micro-batch of an epoch is short when the shard is not a multiple."""
    while True:
        mine = sampler_indices(n, rank, world, epoch, shuffle)
        while cursor < len(mine):
            yield mine[cursor:cursor + batch]
            cursor += batch
        epoch, cursor = epoch + 1, 0


class ShardLoader:
    """Micro-batch iterator over ``AudioTextShards``: ``order`` yields index lists; yields (pcm int16 [B, 480000], text_input i64
    [B, 448], text_y i64 [B, 448], text_len i32 [B]) on ``device``.

    ``workers`` threads fill pinned ring slots (np.load releases the GIL; the copy into the slot is one memcpy + one memset of the
    silence); a slot's H2D runs on a private copy stream as soon as its files are in -- i.e. under the previous micro-batch's
    kernels -- into a persistent device slot; the consumer's stream waits on the copy's event.  Slots are recycled by events, never
    by host synchronisation of the compute stream.  On a CPU ``device`` (tests) the same code path runs without streams.

    The returned tensors are views of ring slots: valid until the next call, and to be consumed on the stream that is current at
    that next call (one compute stream, as in the training loop)."""

    def __init__(self, shards: AudioTextShards, order, device, batch: int, workers: int = 8, depth: int = 2):
        from concurrent.futures import ThreadPoolExecutor
        self.shards, self.order, self.device, self.B = shards, iter(order), torch.device(device), int(batch)
        self.cuda = self.device.type == "cuda"
        self.depth = max(1, int(depth))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        n_slot = self.depth + 2  # `depth` in flight + the one being consumed + the one just handed back
        pin = dict(pin_memory=True) if self.cuda else {}
        self.h = [dict(pcm=torch.zeros(self.B, N_SAMPLES, dtype=torch.int16, **pin), ti=torch.full((self.B, N_TEXT_CTX), PAD_ID, dtype=torch.int64, **pin),
                       ty=torch.full((self.B, N_TEXT_CTX), PAD_ID, dtype=torch.int64, **pin), tl=torch.zeros(self.B, dtype=torch.int32, **pin))
                  for _ in range(n_slot)]
        self.d = [{k: torch.empty_like(v, device=self.device) for k, v in slot.items()} for slot in self.h] if self.cuda else self.h
        # (high priority: a default-priority stream may share the compute stream's hardware queue on ROCm and run behind it -- see ddp.GradReducer)
        self.copy_stream = torch.cuda.Stream(self.device, priority=-1) if self.cuda else None
        self.uploaded = [None] * n_slot   # event: the slot's H2D has completed (host slot reusable, device slot readable)
        self.consumed = [None] * n_slot   # event on the consumer's stream: the device slot may be overwritten
        self.pending = []                 # [slot, futures, upload issued?, rows]
        self.next_slot = 0
        self.last = None                  # slot handed out by the previous __next__
        self.last_span = None             # HOST int32 [b]: supervised span of the batch the last __next__ returned (synth.supervised_span_host)
        self._fill()

    def _load_into(self, slot: int, row: int, index: int):
        pcm, ti, ty, tl = self.shards.load(index)
        h = self.h[slot]
        dst = h["pcm"][row].numpy()
        n = pcm.shape[0]
        dst[:n] = pcm
        dst[n:] = 0
        h["ti"][row] = ti
        h["ty"][row] = ty
        h["tl"][row] = tl

    def _fill(self):
        while len(self.pending) < self.depth:
            try:
                idx = next(self.order)
            except StopIteration:
                return
            assert 0 < len(idx) <= self.B, (len(idx), self.B)  # (the last micro-batch of an epoch may be short: drop_last=False)
            slot = self.next_slot
            self.next_slot = (slot + 1) % len(self.h)
            if self.uploaded[slot] is not None:
                self.uploaded[slot].synchronize()  # the previous H2D out of this pinned slot (long finished in steady state)
            self.pending.append([slot, [self.pool.submit(self._load_into, slot, r, int(i)) for r, i in enumerate(idx)], False, len(idx)])

    def _upload(self, entry):
        slot, futures = entry[0], entry[1]
        for f in futures:
            f.result()
        if self.cuda:
            with torch.cuda.stream(self.copy_stream):
                if self.consumed[slot] is not None:
                    self.copy_stream.wait_event(self.consumed[slot])
                for k in ("pcm", "ti", "ty", "tl"):
                    self.d[slot][k].copy_(self.h[slot][k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.uploaded[slot] = ev
        entry[2] = True

    def __iter__(self):
        return self

    def __next__(self):
        if not self.pending:
            raise StopIteration
        if self.cuda and self.last is not None:  # everything that reads the previous batch has been enqueued by now
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.consumed[self.last] = ev
        entry = self.pending.pop(0)
        if not entry[2]:
            self._upload(entry)
        slot = entry[0]
        # the token rows are still in the host slot (it is recycled n_slot batches later): the span the training step wants as a HOST
        # array comes from there, not from a device read-back
        self.last_span = supervised_span_host(self.h[slot]["ty"][:entry[3]], self.h[slot]["tl"][:entry[3]])
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.uploaded[slot])
        self._fill()
        if self.pending and not self.pending[0][2] and all(f.done() for f in self.pending[0][1]):
            self._upload(self.pending[0])  # the next batch's copy goes out now, under this batch's kernels
        self.last = slot
        d, b = self.d[slot], entry[3]
        return d["pcm"][:b], d["ti"][:b], d["ty"][:b], d["tl"][:b]

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


def write_synthetic_shards(out_dir: str, n: int, per_file: int = 64, timestamps: bool = False, compress: bool = True) -> str:
    """A shard directory in the reference's format from the seeded synthetic generator (tests, smoke runs, the loader benchmark):
    ``audio/{i:06d}.npy`` int16 clips CUT at their last non-silent sample + ``norm_end`` (ms), and ``shard_{k:04d}.jsonl[.gz]`` with
    the pre-tokenised ``tokens`` field.  Reading it back through ShardLoader reproduces ``synth_sample`` bit for bit."""
    from .synth import EOT, synth_sample
    os.makedirs(os.path.join(out_dir, "audio"), exist_ok=True)
    lines = []
    for i in range(n):
        pcm, ti, ty, tl = synth_sample(i, timestamps)
        nz = torch.nonzero(pcm)
        n_valid = int(nz[-1]) + 1 if nz.numel() else 0
        n_valid = (n_valid + 15) // 16 * 16  # norm_end is in ms: 16 samples
        path = os.path.join(out_dir, "audio", f"{i:06d}.npy")
        np.save(path, pcm[:n_valid].numpy())
        tokens = ti[:tl].tolist() + [int(ty[tl - 1])]
        assert tokens[-1] == EOT
        lines.append({"audio_file": path, "subtitle_file": f"synthetic/{i:06d}.vtt", "seg_content": "", "ts_mode": timestamps,
                      "only_no_ts_mode": not timestamps, "norm_end": n_valid // 16, "tokens": tokens, "timestamp_mode": bool(timestamps)})
    for k in range(0, n, per_file):
        body = "".join(json.dumps(x) + "\n" for x in lines[k:k + per_file])
        name = os.path.join(out_dir, f"shard_{k // per_file:04d}.jsonl" + (".gz" if compress else ""))
        if compress:
            with gzip.open(name, "wt") as f:
                f.write(body)
        else:
            with open(name, "wt") as f:
                f.write(body)
    return out_dir
