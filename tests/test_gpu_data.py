"""On-device input path (olmoasr_amd/data.py) on the GPU: shard files -> pinned ring slots -> copy stream -> int16 PCM on the device
-> log-mel kernel; the training entry point fed from shard files.  Reference: scripts/training/train_timestamps.py:84-217, 577-660."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_loader_on_device_matches_the_generator_and_the_mel(tmp_path):
    from olmoasr_amd import data, ops, synth
    d = data.write_synthetic_shards(str(tmp_path), 11, per_file=4)
    shards = data.AudioTextShards(data.load_samples_dicts(d))
    gen = data.epoch_batches(len(shards), 0, 1, 4, shuffle=False)
    order = [next(gen) for _ in range(7)]  # 4, 4, 3 | 4, 4, 3 | 4: two epochs and a bit, short last micro-batches
    assert [len(o) for o in order] == [4, 4, 3, 4, 4, 3, 4] and order[0] == [0, 1, 2, 3] and order[2] == [8, 9, 10]
    loader = data.ShardLoader(shards, iter(order), DEV, batch=4, workers=4, depth=2)
    for idx in order:
        pcm, ti, ty, tl = next(loader)
        assert pcm.shape == (len(idx), 480000) and pcm.dtype == torch.int16 and pcm.is_cuda
        want = synth.synth_samples(idx, DEV)
        mel = ops.log_mel(pcm)
        for got, w in zip((pcm, ti, ty, tl), want):
            assert torch.equal(got, w)
        assert torch.equal(mel, ops.log_mel(want[0]))
    with pytest.raises(StopIteration):
        next(loader)
    loader.close()


def _train_main():
    spec = importlib.util.spec_from_file_location("tt_gpu_data", os.path.join(ROOT, "scripts", "training", "train_timestamps.py"))
    tt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tt)
    return tt


def test_training_from_shard_files_equals_training_from_the_generator(tmp_path):
    """Same samples, same order (shuffle off, world 1): the shard-fed run must reproduce the generator-fed run's losses step by step,
    across epoch boundaries (16 samples, 8 per optimizer step, 5 steps)."""
    from olmoasr_amd import data
    tt = _train_main()
    d = data.write_synthetic_shards(str(tmp_path / "shards"), 16, per_file=5)
    common = ["--model_variant=tiny", "--eff_batch_size=8", "--train_batch_size=4", "--train_steps=5", "--lr=1e-3", "--train_log_freq=1",
              "--ckpt_freq=0", f"--ckpt_dir={tmp_path}", f"--run_id_dir={tmp_path}/ids", "--shuffle=False"]
    a = tt.main(common + ["--exp_name=gen", "--n_synthetic=16"])
    b = tt.main(common + ["--exp_name=shards", f"--samples_dicts_dir={d}", "--synthetic=False"])
    assert len(a) == len(b) == 5
    for ra, rb in zip(a, b):
        assert abs(ra["train_loss"] - rb["train_loss"]) < 2e-4 * abs(ra["train_loss"]), (ra, rb)


def test_training_from_shards_with_shuffle_and_a_short_last_batch(tmp_path):
    from olmoasr_amd import data
    tt = _train_main()
    d = data.write_synthetic_shards(str(tmp_path / "shards"), 10, per_file=10, compress=False)
    log = tt.main(["--model_variant=tiny", "--eff_batch_size=8", "--train_batch_size=4", "--train_steps=4", "--lr=1e-3", "--train_log_freq=1",
                   "--ckpt_freq=0", f"--ckpt_dir={tmp_path}", f"--run_id_dir={tmp_path}/ids", "--exp_name=rag", f"--samples_dicts_dir={d}",
                   "--synthetic=False", "--shuffle=True"])
    assert len(log) == 4 and all(not r["found_inf"] and r["train_loss"] == r["train_loss"] for r in log)
    assert log[-1]["train_loss"] < log[0]["train_loss"]


def test_transcribe_accepts_a_wave_file_path(tmp_path):
    """model.transcribe("file.wav") (olmoasr/transcribe.py:147-148 via whisper.audio.load_audio) == transcribe(samples)."""
    import wave
    import numpy as np
    from olmoasr_amd.config.model_dims import ModelDimensions
    from olmoasr_amd.model import OLMoASR
    g = torch.Generator().manual_seed(0)
    pcm = (torch.randn(16000 * 35, generator=g) * 0.1).clamp_(-1, 1)
    i16 = torch.round(pcm * 32767).to(torch.int16).numpy()
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(i16.tobytes())
    net = OLMoASR(ModelDimensions(80, 1500, 384, 6, 1, 51864, 448, 384, 6, 1), device=DEV, seed=3, inference=True)
    opts = dict(temperature=0.0, logprob_threshold=None, no_speech_threshold=None, sample_len=6)
    a = net.transcribe(str(tmp_path / "a.wav"), **opts)
    b = net.transcribe(i16.astype(np.float32) / 32768.0, **opts)
    assert a["tokens"] == b["tokens"] and len(a["segments"]) == len(b["segments"]) >= 2
