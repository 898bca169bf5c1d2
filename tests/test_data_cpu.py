"""Real-data input path (olmoasr_amd/data.py) on the CPU: shard files, the sampler, the pad_or_trim chain and batch assembly.
Reference: scripts/training/train_timestamps.py:84-217 (AudioTextDataset), :577-604 (open_dicts_file), :633-638 (sampler)."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from olmoasr_amd import data, synth


def test_convert_to_milliseconds_follows_the_reference_parser():
    assert data.convert_to_milliseconds("00:00:29.980") == 29980
    assert data.convert_to_milliseconds("01:02:03.004") == 3723004
    with pytest.raises(ValueError):
        data.convert_to_milliseconds("29.98")


@pytest.mark.parametrize("n_loaded,norm_end,want", [(480000, None, 480000), (500000, None, 480000), (100000, None, 100000),
                                                    (480000, 20000, 320000), (480000, "00:00:20.000", 320000), (100000, 20000, 100000),
                                                    (480000, 0, 480000), (480000, 31000, 480000), (480000, "", 480000)])
def test_valid_samples_is_the_pad_or_trim_chain(n_loaded, norm_end, want):
    """preprocess_audio: pad_or_trim(norm_end * 16) then pad_or_trim(480000) -- restated with numpy and compared sample by sample."""
    assert data.valid_samples(n_loaded, norm_end) == want
    rng = np.random.default_rng(0)
    arr = rng.integers(-3000, 3000, n_loaded).astype(np.int16)

    def pad_or_trim(a, length=480000):
        return a[:length] if a.shape[0] > length else np.pad(a, (0, length - a.shape[0]))
    ref = arr.astype(np.float32) / 32768.0
    if norm_end:
        ms = data.convert_to_milliseconds(norm_end) if isinstance(norm_end, str) else norm_end
        ref = pad_or_trim(pad_or_trim(ref, ms * 16))
    else:
        ref = pad_or_trim(ref)
    got = np.zeros(480000, np.int16)
    got[:want] = arr[:want]
    assert np.array_equal(got.astype(np.float32) / 32768.0, ref)


def test_shard_files_gz_plain_and_missing_tokens(tmp_path):
    rows = [{"audio_file": f"a{i}.npy", "subtitle_file": f"s{i}.vtt", "seg_content": "", "ts_mode": False, "only_no_ts_mode": True,
             "norm_end": 1000 * i} for i in range(5)]
    with gzip.open(tmp_path / "b.jsonl.gz", "wt") as f:
        f.write("".join(json.dumps(r) + "\n" for r in rows[:3]))
    with open(tmp_path / "a.jsonl", "wt") as f:
        f.write("".join(json.dumps(r) + "\n" for r in rows[3:]))
    got = data.load_samples_dicts(str(tmp_path))
    assert [g["audio_file"] for g in got] == ["a3.npy", "a4.npy", "a0.npy", "a1.npy", "a2.npy"]  # sorted file order
    with pytest.raises(KeyError, match="tokens"):
        data.AudioTextShards(got).load(0)
    with pytest.raises(FileNotFoundError):
        data.load_samples_dicts(str(tmp_path / "nothing"))


@pytest.mark.parametrize("world,n", [(1, 10), (2, 11), (8, 100)])
def test_sampler_is_torch_distributed_sampler(world, n):
    from torch.utils.data.distributed import DistributedSampler
    for epoch in (0, 3):
        seen = []
        for rank in range(world):
            mine = data.sampler_indices(n, rank, world, epoch)
            sp = DistributedSampler(list(range(n)), num_replicas=world, rank=rank, shuffle=True, seed=42, drop_last=False)
            sp.set_epoch(epoch)
            assert mine == list(sp)
            seen += mine
        assert set(seen) == set(range(n)) and len(seen) == -(-n // world) * world  # padded to a multiple of world, as the reference
    assert data.sampler_indices(n, 0, world, 0) != data.sampler_indices(n, 0, world, 1) or n < 3


@pytest.mark.parametrize("timestamps", [False, True])
def test_shard_loader_reproduces_the_synthetic_samples_bit_for_bit(tmp_path, timestamps):
    """write_synthetic_shards -> ShardLoader (CPU device) == synth_sample: the trimmed .npy + norm_end round trip rebuilds the
    zero-padded clip, the tokens field rebuilds text_input / text_y / text_len."""
    d = data.write_synthetic_shards(str(tmp_path), 7, per_file=3, timestamps=timestamps)
    samples = data.load_samples_dicts(d)
    assert len(samples) == 7 and all(os.path.getsize(s["audio_file"]) <= 960128 for s in samples)
    shards = data.AudioTextShards(samples)
    order = [[0, 1, 2], [3, 4, 5], [6, 0, 1], [2, 3, 4], [5, 6, 0]]
    loader = data.ShardLoader(shards, iter(order), "cpu", batch=3, workers=3, depth=2)
    n = 0
    for idx, (pcm, ti, ty, tl) in zip(order, loader):
        for r, i in enumerate(idx):
            want = synth.synth_sample(i, timestamps)
            assert torch.equal(pcm[r], want[0]) and torch.equal(ti[r], want[1]) and torch.equal(ty[r], want[2]) and int(tl[r]) == want[3]
        n += 1
    assert n == len(order)
    loader.close()


def test_text_fn_plug_and_timestamp_mode_use_the_full_clip(tmp_path):
    arr = (np.arange(480000) % 1000 - 500).astype(np.int16)
    np.save(tmp_path / "x.npy", arr)
    s = {"audio_file": str(tmp_path / "x.npy"), "subtitle_file": "x.vtt", "seg_content": "hello", "ts_mode": True, "only_no_ts_mode": False,
         "norm_end": 10000}
    toks = [synth.SOT, synth.NO_TIMESTAMPS, 11, 12, synth.EOT]
    sh = data.AudioTextShards([s], text_fn=lambda d: (toks, False, d["norm_end"]))
    pcm, ti, ty, tl = sh.load(0)
    assert pcm.shape[0] == 160000 and tl == 4 and ti[:4].tolist() == toks[:-1] and ty[:4].tolist() == toks[1:] and int(ti[4]) == synth.PAD_ID
    sh = data.AudioTextShards([s], text_fn=lambda d: (toks, True, d["norm_end"]))   # timestamp mode: norm_end is dropped (:148-149)
    assert sh.load(0)[0].shape[0] == 480000
    sh = data.AudioTextShards([s], text_fn=lambda d: (toks, False, 5000))          # text processing corrected norm_end (:150-151)
    assert sh.load(0)[0].shape[0] == 80000
    with pytest.raises(ValueError, match="n_text_ctx"):
        data.AudioTextShards([s], text_fn=lambda d: (list(range(460)), False, None)).load(0)


def test_load_audio_reads_wave_and_npy_without_ffmpeg(tmp_path):
    """whisper.audio.load_audio's contract (mono float32 in [-1, 1] at 16 kHz) for the codec-free formats."""
    import wave
    from olmoasr_amd.audio import load_audio
    t = np.arange(16000) / 16000.0
    tone = (0.5 * np.sin(2 * np.pi * 440.0 * t) * 32767).astype(np.int16)
    with wave.open(str(tmp_path / "mono16k.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(tone.tobytes())
    a = load_audio(str(tmp_path / "mono16k.wav"))
    assert a.dtype == np.float32 and a.shape == (16000,) and np.array_equal(a, tone.astype(np.float32) / 32768.0)
    t8 = np.arange(8000) / 8000.0
    st = np.stack([(0.5 * np.sin(2 * np.pi * 440.0 * t8) * 32767).astype(np.int16)] * 2, axis=1)
    with wave.open(str(tmp_path / "stereo8k.wav"), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(8000); w.writeframes(st.tobytes())
    b = load_audio(str(tmp_path / "stereo8k.wav"))
    assert b.shape == (16000,) and float(np.abs(b[200:-200] - a[200:-200]).max()) < 2e-2  # channel mean + 8 -> 16 kHz polyphase
    np.save(tmp_path / "clip.npy", tone)
    assert np.array_equal(load_audio(str(tmp_path / "clip.npy")), a)
    with pytest.raises(RuntimeError, match="Failed to load audio"):
        (tmp_path / "x.mp3").write_bytes(b"not audio")
        load_audio(str(tmp_path / "x.mp3"))
