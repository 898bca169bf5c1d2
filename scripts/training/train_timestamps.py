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
What is replaced: tokenizer/W&B/eval plumbing (out of scope); model/loss/backward/optimizer/DDP -> the HIP
engine (olmoasr_amd).  Data: ``--samples_dicts_dir DIR --synthetic False`` reads the reference's shard files (:577-604, :2255-2266)
through the on-device input path (olmoasr_amd/data.py: DistributedSampler order, int16 clips via pinned ring slots and a copy
stream, log-mel on the GPU, ``text_len`` instead of the [448, 448] mask -- SURVEY.md section 8f-1; tokens come pre-tokenised or from
a ``text_fn`` plug because the whisper tokenizer is not vendored); the default is a seeded synthetic dataset with the reference's
token layout, sharded like DistributedSampler (:633-638).
"""
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


# The reference's entry is ``Fire(main)`` (train_timestamps.py:2098-2134): flags are ``--name=value`` / ``--name value`` with
# Python-literal values (``--betas='(0.9, 0.98)'``, ``--pin_memory=True``, ``--ckpt_file_name=None``).  Same flag names and
# defaults here; the flags of subsystems that are out of scope (data curation, eval sets, W&B) are
# ACCEPTED -- the reference launcher (configs/job_configs/training/filtered/*_sn.sh:65-100) passes all of them -- and
# reported as ignored.
REFERENCE_FLAGS = dict(
    model_variant="tiny", exp_name="olmoasr_amd_run", job_type="train", samples_dicts_dir=None, train_steps=10, epoch_steps=0,
    ckpt_file_name=None, ckpt_dir="checkpoints", log_dir="logs", eval_dir="data/eval", run_id_dir="run_ids", lr=1.5e-3,
    betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1, max_grad_norm=1.0, eff_batch_size=256, train_batch_size=8, eval_batch_size=32,
    num_workers=10, prefetch_factor=2, pin_memory=True, shuffle=True, persistent_workers=True, run_eval=False, train_log_freq=20000,
    eval_freq=20000, ckpt_freq=2500, verbose=False, precision="bfloat16", hardware="MI355X", async_eval=False, eval_script_path=None,
    eval_wandb_log=False, eval_on_gpu=True)
IGNORED_FLAGS = ("eval_dir", "eval_batch_size", "pin_memory", "persistent_workers", "verbose",
                 "async_eval", "eval_script_path", "eval_wandb_log", "eval_on_gpu", "job_type", "log_dir")
NATIVE_FLAGS = dict(  # additions of this implementation
    synthetic=True, n_synthetic=4096, timestamps=False, seed=0, bucket_cap_mb=128.0, reducer="allreduce", resume=False,
    force_dist=False,  # run the RCCL group / bucketed exchange / sharded step even at world_size 1 (single-GPU rehearsal of the N > 1 path)
    zero_stage=0,  # zero_stage 1: AdamW moments sharded over the ranks (olmoasr_amd/zero.py; the reference's FSDP script's role)
    span_backward=True,  # decoder backward over the supervised span only (oasr_train_fwd_bwd_span; same loss / gradients, forward over all 448)
    span_forward=True)  # the decoder's forward leaves the padded positions out too (their logits are read by nothing; -4.7 % more); False = forward over all 448


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _literal(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def parse_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = Args({**REFERENCE_FLAGS, **NATIVE_FLAGS})
    given = set()
    i = 0
    while i < len(argv):
        tok = argv[i]
        if not tok.startswith("--"):
            raise SystemExit(f"unexpected positional argument {tok!r} (flags are --name=value, as with the reference's Fire entry)")
        if "=" in tok:
            name, val = tok[2:].split("=", 1)
        elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
            name, val = tok[2:], argv[i + 1]
            i += 1
        else:
            name, val = tok[2:], "True"  # bare flag
        name = name.replace("-", "_")
        if name not in args:
            raise SystemExit(f"unknown flag --{name}; known: {sorted(args)}")
        args[name] = _literal(val)
        given.add(name)
        i += 1
    ignored = sorted(f for f in given if f in IGNORED_FLAGS)
    if ignored and int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"event": "ignored_flags", "flags": ignored,
                          "why": "data curation / eval sets / W&B are outside the hot path (SURVEY.md section 2)"}), flush=True)
    if args.ckpt_file_name in (None, "None"):
        args.ckpt_file_name = ""            # train_timestamps.py:2198-2199
    if isinstance(args.betas, str):
        args.betas = ast.literal_eval(args.betas)
    if args.precision == "float16":
        raise SystemExit("--precision float16: the MI355X-native engine computes in bfloat16 (production, = the reference's "
                         "autocast(bfloat16) path) or float32 (validation kernels); the fp16 autocast path of the reference "
                         "(train_timestamps.py:2128 default) has no native counterpart -- pass --precision bfloat16")
    if args.precision not in ("bfloat16", "float32"):
        raise SystemExit(f"--precision must be bfloat16 | float32 | float16, got {args.precision!r}")
    return args


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


TAGS = ["ddp-train", "grad-acc", "fp16"]  # train_timestamps.py:2201-2206 (part of the checkpoint file names)


def scheduler_state(lr, train_steps, global_step):
    """``LambdaLR.state_dict()`` of the reference's scheduler after ``global_step`` steps (what its load_ckpt feeds to
    ``scheduler.load_state_dict``, :1056): produced by a real LambdaLR on a dummy parameter, so every key this torch
    version expects is present (function lambdas are saved as None, as torch does)."""
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=lr)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda step: lr_lambda(step, train_steps))
    sd = sched.state_dict()
    sd["last_epoch"] = global_step
    sd["_step_count"] = global_step + 1
    sd["_last_lr"] = [lr * lr_lambda(global_step, train_steps)]
    return sd


def build_checkpoint(model_sd, optimizer_sd, scaler_sd, *, global_step, local_step, epoch, dims, lr, train_steps, cursor=0,
                     optimizer_steps=None, best_eval_wer=None):
    """The reference's checkpoint dict (train_timestamps.py:930-958), un-prefixed model keys.  ``dims`` is stored as a
    ``types.SimpleNamespace`` of the ModelDimensions fields: the reference's load_ckpt does ``OLMoASR(dims=ckpt['dims'])``
    (attribute access, :1036) and gen_inf_ckpt reads ``dims.__dict__`` -- both work on it, and it unpickles without this
    package (or the reference's) being importable."""
    import types
    fields = dims if isinstance(dims, dict) else dims.__dict__
    return {"global_step": global_step, "local_step": local_step, "epoch": epoch, "best_eval_wer": best_eval_wer,
            "model_state_dict": model_sd, "optimizer_state_dict": optimizer_sd, "scaler_state_dict": scaler_sd,
            "scheduler_state_dict": scheduler_state(lr, train_steps, global_step), "dims": types.SimpleNamespace(**fields),
            "data_cursor": cursor, "optimizer_steps": global_step if optimizer_steps is None else optimizer_steps}


def run_dir(args, run_id):
    return os.path.join(args.ckpt_dir, f"{args.exp_name}_{run_id}")  # {ckpt_dir}/{exp_name}_{run_id} (:956)


def save_ckpt(net, scaler, global_step, local_step, epoch, args, dims, rank, run_id, cursor=0, optimizer_steps=None,
              file_name="latesttrain", sharded=None):
    """save_ckpt (train_timestamps.py:894-972): the ``_ddp`` file carries ``module.`` keys, the ``_non_ddp`` file plain ones."""
    moments = sharded.gather_state() if sharded is not None else None  # (collective: every rank takes part)
    if rank != 0:
        return None
    os.makedirs(run_dir(args, run_id), exist_ok=True)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    steps = global_step if optimizer_steps is None else optimizer_steps
    base = build_checkpoint(sd, net.optimizer_state_dict(step=steps, lr=args.lr * lr_lambda(global_step, args.train_steps), betas=args.betas,
                                                         eps=args.eps, weight_decay=args.weight_decay, moments=moments),
                            scaler.state_dict(), global_step=global_step, local_step=local_step, epoch=epoch, dims=dims, lr=args.lr,
                            train_steps=args.train_steps, cursor=cursor, optimizer_steps=steps)
    tag = f"{file_name}_{global_step:08}_{args.model_variant}_{'_'.join(TAGS)}"
    paths = []
    for suffix, prefix in (("ddp", "module."), ("non_ddp", "")):
        ck = dict(base)
        ck["model_state_dict"] = {prefix + k: v for k, v in sd.items()}
        p = os.path.join(run_dir(args, run_id), f"{tag}_{suffix}.pt")
        torch.save(ck, p)
        paths.append(p)
    return paths


def find_ckpt(args, run_id):
    """File selection of the reference's load_ckpt (train_timestamps.py:1012-1030): an explicit path, a file-name prefix in
    the run directory, or the latest ``*_fp16_ddp.pt`` of this run."""
    name = args.ckpt_file_name
    if name and "/" in name:
        return name
    pat = os.path.join(run_dir(args, run_id), f"{name or '*'}_*_{args.model_variant}_*_fp16_ddp.pt")
    files = [f for f in glob.glob(pat) if not f.endswith("_non_ddp.pt")]
    if not files:
        raise FileNotFoundError(f"no checkpoint matches {pat}")
    return max(files, key=lambda f: int(os.path.basename(f).split("_")[1]))


def load_ckpt(net, scaler, path, sharded=None):
    """Resume (train_timestamps.py:975-1074): model (``module.`` keys), AdamW moments, GradScaler, counters.  Accepts files
    written by the reference (``dims`` pickled as ``olmoasr.config.model_dims.ModelDimensions``: see hub.load_checkpoint)."""
    from olmoasr_amd import hub
    ck = hub.load_checkpoint(path)
    net.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in ck["model_state_dict"].items()})
    net.refresh_shadow()
    if sharded is not None:  # fill full-length scratch moments, keep this rank's range
        full = (torch.zeros_like(net.flat_params), torch.zeros_like(net.flat_params))
        opt_steps = net.load_optimizer_state_dict(ck["optimizer_state_dict"], into=full)
        sharded.load_state(*full)
        del full
    else:
        opt_steps = net.load_optimizer_state_dict(ck["optimizer_state_dict"])
    scaler.load_state_dict(ck["scaler_state_dict"])
    # torch's AdamW does not advance its step on a GradScaler-skipped iteration: opt_steps <= global_step
    if not 0 <= opt_steps <= ck["global_step"]:  # (checkpoints written elsewhere may count differently: report, do not refuse)
        print(f"[load_ckpt] optimizer step {opt_steps} outside [0, global_step = {ck['global_step']}]")
    return ck["global_step"], ck["local_step"], ck["epoch"], int(ck.get("data_cursor", 0)), int(ck.get("optimizer_steps", opt_steps))


def get_run_id(args, rank, world_size):
    """Run id bookkeeping of main() (train_timestamps.py:2182-2196): {run_id_dir}/{exp_name}.txt names the run whose
    checkpoint directory is {ckpt_dir}/{exp_name}_{run_id}; a new id is drawn when there is none (the reference takes
    W&B's)."""
    path = os.path.join(args.run_id_dir, f"{args.exp_name}.txt")
    run_id = None
    if os.path.exists(path):
        run_id = open(path).read().strip()
        if not os.path.exists(run_dir(args, run_id)):
            run_id = None
    if run_id is None:
        import uuid
        obj = [uuid.uuid4().hex[:8] if rank == 0 else None]
        if world_size > 1:
            dist.broadcast_object_list(obj, src=0)
        run_id = obj[0]
        if rank == 0:
            os.makedirs(args.run_id_dir, exist_ok=True)
            with open(path, "w") as f:
                f.write(run_id)
    return run_id


def gen_pred(logits, text_y):
    """gen_pred (train_timestamps.py:1077-1122) at token level: ``pred = argmax(softmax(logits))`` per position (a HIP
    kernel over the fp32 logits: argmax is invariant under softmax), predictions cut after the first <|endoftext|>
    (remove_after_endoftext), targets with the 51864 padding filtered out and closed by <|endoftext|>.  Decoding ids to text
    (tokenizer.decode_with_timestamps) and the WER on text need the un-vendored tokenizer / jiwer: callers get ids."""
    from olmoasr_amd import ops
    B, S, V = logits.shape
    pred, _ = ops.pick_tokens(logits.reshape(B * S, V), want_logprob=False)
    preds, tgts = [], []
    for row in pred.view(B, S).cpu().tolist():
        preds.append(row[:row.index(50256) + 1] if 50256 in row else row)
    for row in text_y.cpu().tolist():
        row = [t for t in row if t != 51864]
        tgts.append((row[:row.index(50256)] if 50256 in row else row) + [50256])
    return preds, tgts


def token_error_rate(preds, tgts):
    """Token-level analogue of calc_pred_wer (:1125-1180): (substitutions + deletions + insertions) / reference tokens."""
    errs = n = 0
    for p, t in zip(preds, tgts):
        prev = list(range(len(t) + 1))
        for i, a in enumerate(p, 1):
            cur = [i]
            for j, b in enumerate(t, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a != b)))
            prev = cur
        errs += prev[-1]
        n += len(t)
    return errs / max(1, n)


def evaluate(net, indices, dev, timestamps=False, sample_len=32, batch=8):
    """evaluate() of the reference (train_timestamps.py:1835-1919) on held-out SYNTHETIC clips (the eval sets themselves are
    out of scope): greedy ``model.decode(audio_input, DecodingOptions(language="en", without_timestamps=True))`` (:1916-1919),
    scored as token error rate against the transcript ids."""
    from olmoasr_amd import ops
    from olmoasr_amd.decoding import DecodingOptions
    from olmoasr_amd.synth import synth_samples
    preds, tgts = [], []
    for i in range(0, len(indices), batch):
        pcm, ti, ty, tl = synth_samples(indices[i:i + batch], dev, timestamps)
        res = net.decode(ops.log_mel(pcm), DecodingOptions(language="en", without_timestamps=True, sample_len=sample_len))
        preds += [r.tokens for r in res]
        tgts += [[t for t in row[1:n] if t < 50257][:sample_len] for row, n in zip(ti.cpu().tolist(), tl.cpu().tolist())]
    return token_error_rate(preds, tgts)


def host_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def main(argv=None):
    args = parse_args(argv)
    from olmoasr_amd import ddp, ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR
    from olmoasr_amd.synth import SynthLoader

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))          # env rank discovery, train_timestamps.py:2227-2230
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = bool(args.force_dist) and world_size == 1
    own_group = (world_size > 1 or force_dist) and not dist.is_initialized()
    if own_group and not force_dist:
        # world_size > 1: every rank must name the SAME rendezvous, so it has to come from the launcher (torchrun exports both).  A
        # per-process default would make each rank wait on a different port until the RCCL timeout instead of failing here.
        missing = [k for k in ("MASTER_ADDR", "MASTER_PORT") if k not in os.environ]
        if missing:
            raise SystemExit(f"train_timestamps.py: WORLD_SIZE={world_size} but {' and '.join(missing)} not set -- launch with "
                             f"torchrun (train_timestamps.py:2227-2230 reads the same environment), or export them for every rank")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if own_group:
        if force_dist:  # the single-rank rehearsal needs no launcher: loopback rendezvous on a port of its own
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world_size, pg_options=ddp.rccl_options())  # RCCL over xGMI

    betas = tuple(args.betas)
    dims = VARIANT_TO_DIMS[args.model_variant]
    net = OLMoASR(dims, device=dev, seed=args.seed, compute_dtype=args.precision)
    ddp.broadcast_parameters(net.flat_params)  # DDP ctor _sync_module_states
    net.refresh_shadow()
    sharded = None
    if int(args.zero_stage) == 1:
        from olmoasr_amd import zero
        sharded = zero.ShardedOptimizer(net.flat_params, net.flat_grads, zero.NativeBackend(net), force=force_dist)  # moments for the owned range only
    else:
        net.init_optimizer_state()
    reducer = (ddp.GradReducer(net.flat_grads, net.grad_segments, bucket_cap_mb=args.bucket_cap_mb, algo=args.reducer, force=force_dist)
               if (world_size > 1 or force_dist) and sharded is None else None)
    scaler = GradScalerState()
    accum = accumulation_steps(args.eff_batch_size, world_size, args.train_batch_size)
    mine = ddp.shard_indices(args.n_synthetic, rank, world_size)
    loss_buf = torch.zeros(1, device=dev)
    global_step, local_step, cursor, epoch, optimizer_steps = 0, 0, 0, 0, 0
    run_id = get_run_id(args, rank, world_size)
    if args.resume or args.ckpt_file_name:
        global_step, local_step, epoch, cursor, optimizer_steps = load_ckpt(net, scaler, find_ckpt(args, run_id), sharded)
    # every rank of a node shares the host cores: the reference's --num_workers is per rank, but oversubscribing a 16-core
    # cgroup with 8 x 10 generator threads would make the loader, not the GPU, set the pace
    workers = max(1, min(int(args.num_workers), host_cores() // max(1, local_world)))
    log = []
    if rank == 0:
        print(json.dumps({"event": "start", "world_size": world_size, "accumulation_steps": accum, "model": args.model_variant,
                          "precision": args.precision, "params": net.flat_params.numel(), "run_id": run_id, "loader_threads": workers,
                          "hardware_peak_flops": HARDWARE_TO_FLOPS.get(args.hardware)}), flush=True)

    def batch_order(start):  # the sampler: this rank's shard, cyclic, train_batch_size indices per micro-batch
        c = start
        while True:
            yield [mine[(c + j) % len(mine)] for j in range(args.train_batch_size)]
            c += args.train_batch_size
            if c >= len(mine):
                c = 0

    # --samples_dicts_dir (with --synthetic False): the reference's shard files through the on-device input path (olmoasr_amd/data.py)
    real_data = bool(args.samples_dicts_dir) and not bool(args.synthetic)
    if real_data:
        from olmoasr_amd import data
        shards = data.AudioTextShards(data.load_samples_dicts(args.samples_dicts_dir))
        per_rank = -(-len(shards) // world_size)  # DistributedSampler pads every rank's share to ceil(n / world)
        if rank == 0:
            print(json.dumps({"event": "data", "samples": len(shards), "per_rank": per_rank, "shuffle": bool(args.shuffle)}), flush=True)
        loader = data.ShardLoader(shards, data.epoch_batches(len(shards), rank, world_size, args.train_batch_size, epoch, cursor, bool(args.shuffle)),
                                  dev, batch=args.train_batch_size, workers=workers, depth=max(1, int(args.prefetch_factor)))
    else:
        per_rank = len(mine)
        loader = SynthLoader(batch_order(cursor), dev, workers=workers, depth=max(1, int(args.prefetch_factor)), timestamps=bool(args.timestamps))
    held_out = [args.n_synthetic + i for i in range(8)]  # evaluate(): clips outside every rank's training shard
    while global_step < args.train_steps:
        start_step = time.time()
        net.zero_grad()
        log_now = ((global_step + 1) % args.train_log_freq) == 0  # gen_pred condition of the reference (:1480)
        preds, tgts = [], []
        for i in range(accum):
            pcm, ti, ty, tl = next(loader)
            cursor += args.train_batch_size  # (position in this rank's epoch order; a short last batch also ends the epoch)
            if cursor >= per_rank:
                cursor, epoch = 0, epoch + 1
            last = i == accum - 1
            # (a logging step wants the logits back: it takes the plain step; every other step limits the decoder's backward to the span
            # and lets the encoder's transpose apply the log-mel floor / scale instead of a second pass over the tensor)
            use_span = bool(args.span_backward) and not log_now
            mel, clip_max = ops.log_mel(pcm, finalize=False) if use_span else (ops.log_mel(pcm), None)
            _, logits = net.loss_and_backward(mel, ti, ty, tl, loss_scale=scaler.scale, accumulation_steps=accum, loss_out=loss_buf,
                                              accumulate_loss=i > 0, return_logits=log_now,
                                              segment_events=reducer.segment_events() if (reducer and last) else None,
                                              span=loader.last_span if use_span else None, mel_clip_max=clip_max,
                                              span_forward=use_span and bool(args.span_forward))
            if log_now:
                p_, t_ = gen_pred(logits, ty)
                preds += p_
                tgts += t_
            local_step += 1
        div = 1.0
        if reducer:
            reducer.reduce()
            div = reducer.grad_divisor
        lr = args.lr * lr_lambda(global_step, args.train_steps)
        # torch's AdamW advances its per-parameter step only when GradScaler lets the step through: the bias correction
        # follows the number of APPLIED steps, not global_step
        hyper = dict(step=optimizer_steps + 1, lr=lr, max_grad_norm=args.max_grad_norm, betas=betas, eps=args.eps, weight_decay=args.weight_decay)
        if sharded is not None:
            stats = sharded.step(inv_loss_scale=1.0 / (scaler.scale * sharded.grad_divisor), **hyper)
        else:
            stats = net.optim_step(inv_loss_scale=1.0 / (scaler.scale * div), **hyper)
        found_inf = bool(stats[1].item() != 0)  # host sync once per optimizer step, like scaler.step()
        scaler.update(found_inf)
        if not found_inf:
            optimizer_steps += 1
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
                if preds:
                    rec["train_token_error_rate"] = round(token_error_rate(preds, tgts), 4)
                log.append(rec)
                print(json.dumps(rec), flush=True)
        if args.run_eval and args.eval_freq and global_step % int(args.eval_freq) == 0 and rank == 0:
            print(json.dumps({"event": "eval", "global_step": global_step,
                              "token_error_rate": round(evaluate(net, held_out, dev, bool(args.timestamps)), 4)}), flush=True)
        if args.ckpt_freq and global_step % args.ckpt_freq == 0:
            save_ckpt(net, scaler, global_step, local_step, epoch, args, dims, rank, run_id, cursor, optimizer_steps, sharded=sharded)
    loader.close()
    if args.ckpt_freq and (global_step % args.ckpt_freq) != 0:
        save_ckpt(net, scaler, global_step, local_step, epoch, args, dims, rank, run_id, cursor, optimizer_steps, sharded=sharded)
    if own_group:
        dist.barrier()
        dist.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
