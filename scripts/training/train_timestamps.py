#!/usr/bin/env python
"""torchrun entry point of the MI355X-native training path -- flag-compatible with the part of the reference's
``scripts/training/train_timestamps.py::main`` (train_timestamps.py:2098-2134) that drives the hot loop.

    torchrun --nnodes 1:1 --nproc_per_node 8 --master-addr 127.0.0.1 scripts/training/train_timestamps.py \
        --model_variant medium --precision bfloat16 --eff_batch_size 2048 --train_batch_size 32 --train_steps 100 \
        --lr 1.5e-3 --betas '(0.9, 0.98)' --eps 1e-6 --weight_decay 0.1 --max_grad_norm 1.0 --synthetic True

What is kept from the reference loop (citations into /root/reference/scripts/training/train_timestamps.py):
  env rank discovery :2227-2230 * setup("nccl") :564-574 * accumulation rule + warmup/linear-decay LambdaLR :764-781 *
  GradScaler semantics (enabled for bf16 too, :2349: scale 65536, x2 every 2000 clean steps, x0.5 + skipped step on
  inf/nan) * per-step clip(1.0)+AdamW :1509-1512 * throughput metric audio_min_per_GPU_second :1525-1527 * loss
  all-reduce every train_log_freq :1553 * checkpoint dict keys and file names :930-972.
What is replaced: DataLoader/tokenizer/W&B/eval plumbing (out of scope) -> a seeded synthetic dataset
with the reference's token layout, sharded like DistributedSampler (:633-638); model/loss/backward/optimizer/DDP -> the
HIP engine (olmoasr_amd).  Data is int16 PCM on the device; log-mel runs on the GPU.
"""
import argparse
import ast
import glob
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HARDWARE_TO_FLOPS = {"H100": 900 * 10 ** 12, "L40": 366 * 10 ** 12, "A100": 312 * 10 ** 12, "MI355X": 2500 * 10 ** 12}


def str2bool(v):
    return str(v).lower() in ("1", "true", "yes")


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model_variant", default="tiny")
    ap.add_argument("--exp_name", default="olmoasr_amd_run")
    ap.add_argument("--job_type", default="train")
    ap.add_argument("--ckpt_dir", default="checkpoints")
    ap.add_argument("--log_dir", default="logs")
    ap.add_argument("--eff_batch_size", type=int, default=512)
    ap.add_argument("--train_batch_size", type=int, default=8)
    ap.add_argument("--train_steps", type=int, default=10)
    ap.add_argument("--epoch_steps", type=int, default=0)
    ap.add_argument("--lr", type=float, default=1.5e-3)
    ap.add_argument("--betas", default="(0.9, 0.98)")
    ap.add_argument("--eps", type=float, default=1e-6)
    ap.add_argument("--weight_decay", type=float, default=0.1)
    ap.add_argument("--max_grad_norm", type=float, default=1.0)
    ap.add_argument("--precision", default="bfloat16", choices=["bfloat16"])
    ap.add_argument("--hardware", default="MI355X")
    ap.add_argument("--train_log_freq", type=int, default=5)
    ap.add_argument("--ckpt_freq", type=int, default=0)
    ap.add_argument("--synthetic", type=str2bool, default=True)
    ap.add_argument("--n_synthetic", type=int, default=4096, help="size of the synthetic dataset (samples)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--num_workers", type=int, default=8, help="sample-generation threads (the reference's DataLoader workers)")
    ap.add_argument("--bucket_cap_mb", type=float, default=128.0)
    ap.add_argument("--reducer", default="allreduce", choices=["allreduce", "direct"])
    ap.add_argument("--resume", type=str2bool, default=False, help="continue from the latest checkpoint of --exp_name")
    ap.add_argument("--ckpt_file_name", default="", help="checkpoint to resume from: a path, or a file-name prefix in the run dir")
    return ap.parse_args(argv)


class GradScalerState:
    """torch.cuda.amp.GradScaler defaults (init 65536, growth 2, backoff 0.5, interval 2000), host-side bookkeeping;
    the inf check and the unscale run inside the fused optimizer kernel."""

    def __init__(self):
        self.scale, self.growth_tracker = 65536.0, 0

    def update(self, found_inf: bool):
        if found_inf:
            self.scale *= 0.5
            self.growth_tracker = 0
        else:
            self.growth_tracker += 1
            if self.growth_tracker == 2000:
                self.scale *= 2.0
                self.growth_tracker = 0

    def state_dict(self):
        return {"scale": self.scale, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                "_growth_tracker": self.growth_tracker}

    def load_state_dict(self, sd):
        self.scale, self.growth_tracker = float(sd["scale"]), int(sd["_growth_tracker"])


def accumulation_steps(eff_batch_size, world_size, train_batch_size):
    """prepare_sched, train_timestamps.py:764-770."""
    if eff_batch_size <= world_size * train_batch_size:
        return 1
    return eff_batch_size // (world_size * train_batch_size)


def lr_lambda(global_step, train_steps):
    """prepare_sched, train_timestamps.py:771-781."""
    warmup = math.ceil(0.002 * train_steps)
    if global_step < warmup:
        return float(global_step) / float(max(1, warmup))
    return max(0.0, float(train_steps - global_step) / float(max(1, train_steps - warmup)))


def save_ckpt(net, opt_state, scaler, global_step, local_step, epoch, args, dims, rank, cursor=0, lr=0.0, betas=(0.9, 0.98)):
    """Checkpoint dict with the reference's keys (train_timestamps.py:930-972); ``_ddp`` file has ``module.`` keys.  The
    optimizer entry is torch.optim.AdamW's own state_dict layout, so either side can load the other's file."""
    if rank != 0:
        return None
    os.makedirs(os.path.join(args.ckpt_dir, args.exp_name), exist_ok=True)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    base = {"global_step": global_step, "local_step": local_step, "epoch": epoch, "best_eval_wer": None,
            "optimizer_state_dict": net.optimizer_state_dict(step=global_step, lr=lr, betas=betas, eps=args.eps,
                                                             weight_decay=args.weight_decay),
            "scaler_state_dict": scaler.state_dict(),
            "scheduler_state_dict": {"last_epoch": global_step, "_step_count": global_step + 1, "base_lrs": [args.lr]},
            "dims": dims.__dict__, "data_cursor": cursor}
    tag = f"latesttrain_{global_step:08d}_{args.model_variant}_" + "_".join(["ddp", "fp16"])
    paths = []
    for suffix, prefix in (("ddp", "module."), ("non_ddp", "")):
        ck = dict(base)
        ck["model_state_dict"] = {prefix + k: v for k, v in sd.items()}
        p = os.path.join(args.ckpt_dir, args.exp_name, f"{tag}_{suffix}.pt")
        torch.save(ck, p)
        paths.append(p)
    return paths


def find_ckpt(args):
    """File selection of the reference's load_ckpt (train_timestamps.py:1012-1030): explicit path, or the latest
    ``*_ddp.pt`` of this experiment."""
    if args.ckpt_file_name and "/" in args.ckpt_file_name:
        return args.ckpt_file_name
    pat = os.path.join(args.ckpt_dir, args.exp_name, f"{args.ckpt_file_name or '*'}_*_{args.model_variant}_*_ddp.pt")
    files = [f for f in glob.glob(pat) if not f.endswith("_non_ddp.pt")]
    if not files:
        raise FileNotFoundError(f"no checkpoint matches {pat}")
    return max(files, key=lambda f: int(os.path.basename(f).split("_")[1]))


def load_ckpt(net, scaler, path):
    """Resume (train_timestamps.py:975-1074): model (``module.`` keys), AdamW moments, GradScaler, counters."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    net.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in ck["model_state_dict"].items()})
    net.refresh_shadow()
    opt_steps = net.load_optimizer_state_dict(ck["optimizer_state_dict"])
    scaler.load_state_dict(ck["scaler_state_dict"])
    assert opt_steps in (0, ck["global_step"]), (opt_steps, ck["global_step"])
    return ck["global_step"], ck["local_step"], ck["epoch"], int(ck.get("data_cursor", 0))


def main(argv=None):
    args = parse_args(argv)
    from olmoasr_amd import ddp, ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd.synth import SynthLoader

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    betas = ast.literal_eval(args.betas) if isinstance(args.betas, str) else args.betas
    dims = VARIANT_TO_DIMS[args.model_variant]
    net = OLMoASR(dims, device=dev, seed=args.seed)
    ddp.broadcast_parameters(net.flat_params)  # DDP ctor _sync_module_states
    net.refresh_shadow()
    opt_state = net.init_optimizer_state()
    reducer = ddp.GradReducer(net.flat_grads, net.grad_segments, bucket_cap_mb=args.bucket_cap_mb, algo=args.reducer) if world_size > 1 else None
    scaler = GradScalerState()
    accum = accumulation_steps(args.eff_batch_size, world_size, args.train_batch_size)
    mine = ddp.shard_indices(args.n_synthetic, rank, world_size)
    loss_buf = torch.zeros(1, device=dev)
    global_step, local_step, cursor, epoch = 0, 0, 0, 0
    if args.resume or args.ckpt_file_name:
        global_step, local_step, epoch, cursor = load_ckpt(net, scaler, find_ckpt(args))
    log = []
    if rank == 0:
        print(json.dumps({"event": "start", "world_size": world_size, "accumulation_steps": accum, "model": args.model_variant,
                          "params": net.flat_params.numel(), "hardware_peak_flops": HARDWARE_TO_FLOPS.get(args.hardware)}), flush=True)
    def batch_order(start):  # the sampler: this rank's shard, cyclic, train_batch_size indices per micro-batch
        c = start
        while True:
            yield [mine[(c + j) % len(mine)] for j in range(args.train_batch_size)]
            c += args.train_batch_size
            if c >= len(mine):
                c = 0

    loader = SynthLoader(batch_order(cursor), dev, workers=args.num_workers)
    while global_step < args.train_steps:
        start_step = time.time()
        net.zero_grad()
        for i in range(accum):
            cursor += args.train_batch_size
            if cursor >= len(mine):
                cursor, epoch = 0, epoch + 1
            pcm, ti, ty, tl = next(loader)
            mel = ops.log_mel(pcm)
            last = i == accum - 1
            net.loss_and_backward(mel, ti, ty, tl, loss_scale=scaler.scale, accumulation_steps=accum, loss_out=loss_buf,
                                  accumulate_loss=i > 0, segment_events=reducer.segment_events() if (reducer and last) else None)
            local_step += 1
        div = 1.0
        if reducer:
            reducer.reduce()
            div = reducer.grad_divisor
        lr = args.lr * lr_lambda(global_step, args.train_steps)
        stats = net.optim_step(step=global_step + 1, lr=lr, inv_loss_scale=1.0 / (scaler.scale * div), max_grad_norm=args.max_grad_norm,
                               betas=betas, eps=args.eps, weight_decay=args.weight_decay)
        found_inf = bool(stats[1].item() != 0)  # host sync once per optimizer step, like scaler.step()
        scaler.update(found_inf)
        global_step += 1
        time_per_step = time.time() - start_step
        throughput = ((args.train_batch_size * accum * 30) / 60) / time_per_step  # audio_min_per_GPU_second (:1525-1527)
        if global_step % args.train_log_freq == 0 or global_step == 1:
            t = loss_buf.clone()
            if world_size > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if rank == 0:
                rec = {"global_step": global_step, "train_loss": float(t) / world_size, "lr": lr, "loss_scale": scaler.scale,
                       "time_per_step": round(time_per_step, 4), "audio_min_per_GPU_second": round(throughput, 3),
                       "audio_sec_per_sec_node": round(throughput * 60 * world_size, 1), "found_inf": found_inf}
                log.append(rec)
                print(json.dumps(rec), flush=True)
        if args.ckpt_freq and global_step % args.ckpt_freq == 0:
            save_ckpt(net, opt_state, scaler, global_step, local_step, epoch, args, dims, rank, cursor, lr, betas)
    loader.close()
    if args.ckpt_freq:
        save_ckpt(net, opt_state, scaler, global_step, local_step, epoch, args, dims, rank, cursor,
                  args.lr * lr_lambda(max(global_step - 1, 0), args.train_steps), betas)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
