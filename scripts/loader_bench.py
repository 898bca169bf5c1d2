"""Throughput of the on-device input path alone (olmoasr_amd/data.py: shard files -> pinned slots -> H2D -> log-mel on the GPU),
to be read against the clip rate the training step consumes (OLMoASR-medium: 256 clips per 1.57 s = 163 clips/s per GPU).
usage: python scripts/loader_bench.py [n_clips] [batch] [workers...]"""
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from olmoasr_amd import data, ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    workers = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8, 16]
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        d = data.write_synthetic_shards(tmp, n, per_file=128)
        shards = data.AudioTextShards(data.load_samples_dicts(d))
        print(json.dumps({"event": "shards written", "clips": n, "seconds": round(time.time() - t0, 1),
                          "host_cores": len(os.sched_getaffinity(0))}), flush=True)
        for w in workers:
            gen = data.epoch_batches(len(shards), 0, 1, B, shuffle=True)
            loader = data.ShardLoader(shards, gen, "cuda", batch=B, workers=w, depth=2)
            for _ in range(2):  # warm-up: page cache, pinned slots, mel workspace
                ops.log_mel(next(loader)[0])
            torch.cuda.synchronize()
            t0 = time.time()
            nb = max(4, 2 * n // B)
            for _ in range(nb):
                pcm, ti, ty, tl = next(loader)
                mel = ops.log_mel(pcm)
            torch.cuda.synchronize()
            dt = time.time() - t0
            loader.close()
            print(json.dumps({"loader_threads": w, "micro_batch": B, "clips_per_s": round(nb * B / dt, 1), "audio_s_per_s": round(nb * B * 30 / dt, 1),
                              "ms_per_micro_batch": round(1e3 * dt / nb, 2)}), flush=True)


if __name__ == "__main__":
    main()
