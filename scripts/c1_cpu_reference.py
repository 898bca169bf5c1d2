"""BASELINE config C1: OLMoASR-tiny, batch 2 synthetic 30 s clips, CPU-only PyTorch DDP world_size 1 (gloo) -- the reference's own
CPU-runnable case (plumbing, no GPU).  Runs the UNMODIFIED reference model (imported from /root/reference when it is mounted:
build container only) through one full train step the way scripts/training/train_timestamps.py does it (:1440-1454, 1509-1512:
forward, cross_entropy(ignore_index=51864), backward through DDP, clip_grad_norm_(1.0), AdamW), fp32, and prints audio-seconds per
second (median of 3 after a warm-up) plus the parity of the CPU oracle (the repo's restatement) on the same batch.  Measurement
infrastructure: not imported by the product."""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mel_oracle as me  # noqa: E402
from oracle import model_oracle as mo  # noqa: E402
from oracle import ref_import  # noqa: E402


def main():
    import torch.distributed as dist
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=0, world_size=1)
    pcm, ti, ty, tl = mo.synthetic_batch([0, 1])
    t0 = time.time()
    mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
    t_mel = time.time() - t0
    pm = mo.build_padding_mask(tl)
    dims = mo.VARIANTS["tiny"]
    sd = mo.init_state_dict(dims, seed=0)
    out = {"config": "C1: OLMoASR-tiny, B=2 synthetic 30 s clips, CPU fp32, DDP gloo world_size 1", "cores": len(os.sched_getaffinity(0)),
           "cpu_mel_s_per_clip": round(t_mel / 2, 4)}
    if ref_import.available():
        ref_model, _, ref_dims = ref_import.load()
        net = ref_model.OLMoASR(ref_dims.VARIANT_TO_DIMS["tiny"])
        net.load_state_dict(sd, strict=True)
        ddp = torch.nn.parallel.DistributedDataParallel(net)
        opt = torch.optim.AdamW(ddp.parameters(), lr=1.5e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
        times, loss0 = [], None
        for i in range(4):
            t0 = time.time()
            opt.zero_grad()
            logits = ddp(mel, ti, pm)
            loss = torch.nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), ty.view(-1), ignore_index=mo.PAD_ID)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ddp.parameters(), 1.0)
            opt.step()
            times.append(time.time() - t0)
            loss0 = float(loss) if loss0 is None else loss0
        step = statistics.median(times[1:])
        out.update({"impl": "unmodified reference (olmoasr.model.OLMoASR under torch DDP/gloo)", "step_s": round(step, 3),
                    "audio_seconds_per_second": round(2 * 30 / step, 2), "first_loss": round(loss0, 6)})
        lo, _, _ = mo.loss_and_grads(sd, dims, mel, ti, ty, tl)
        out["oracle_first_loss"] = round(float(lo), 6)
        out["oracle_vs_reference_loss_abs_diff"] = abs(float(lo) - loss0)
    else:
        times = []
        for i in range(4):
            t0 = time.time()
            lo, grads, _ = mo.loss_and_grads(sd, dims, mel, ti, ty, tl)
            times.append(time.time() - t0)
        step = statistics.median(times[1:])
        out.update({"impl": "oracle restatement (reference not mounted)", "step_s": round(step, 3), "audio_seconds_per_second": round(60 / step, 2),
                    "first_loss": round(float(lo), 6)})
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
