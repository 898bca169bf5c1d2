#!/usr/bin/env python
"""bench.py -- OLMoASR DDP training step on MI355X (BASELINE.json metric: audio-seconds/sec/node, train step).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one optimizer step of the reference's train() loop (scripts/training/train_timestamps.py:1405-1549) on a
per-GPU batch of 256 synthetic 30 s clips (BASELINE.json configs[2]: medium, global 2048 on 8 GPUs = 256/GPU; weak
scaling), executed as micro-batches with gradient accumulation exactly like `accumulation_steps` there:
  int16 PCM (resident in HBM) -> log-mel (HIP) -> encoder/decoder forward -> CE(ignore pad)/accum -> backward
  (-> bucketed RCCL all-reduce of the flat fp32 gradient arena, overlapped with the backward of the last micro-batch)
  -> fused unscale + clip_grad_norm_(1.0) + AdamW.
Prints ONE JSON line on rank 0 (driver contract) with `roofline` (dominant kernel = the bf16 MFMA GEMM, measured with
HIP events on its own stream in one extra, identical step) and `cpu_baseline` (the CPU oracle on the host cores,
bounded sample, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"

class PowerSampler:
    """Samples the amdgpu hwmon package power and shader clock of one GPU from a thread (sysfs reads, no subprocess) while a step runs."""

    def __init__(self, index):
        import glob
        self.files = {}
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "freq1_input"))]
        base = None
        try:  # the visible device's PCI address picks its sysfs node (a box may expose more cards than the job can see)
            pr = torch.cuda.get_device_properties(index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for c in cards:
                if os.path.basename(os.path.realpath(os.path.join(c, "..", ".."))).lower().startswith(want):
                    base = c
        except Exception:
            base = None
        self.matched_by = "pci address" if base else "busiest card"
        self.candidates = [base] if base else cards  # no match: sample every card and report the one drawing the most power
        self.samples, self._stop, self._thread = {c: [] for c in self.candidates}, False, None

    @staticmethod
    def _file(base, names):
        for n in names:
            if os.path.exists(os.path.join(base, n)):
                return os.path.join(base, n)
        return None

    @staticmethod
    def _read(path):
        try:
            return float(open(path).read().strip())
        except Exception:
            return None

    def _run(self):
        files = {c: (self._file(c, ("power1_average", "power1_input")), self._file(c, ("freq1_input",))) for c in self.candidates}
        while not self._stop:
            for c, (pf, ff) in files.items():
                self.samples[c].append((self._read(pf) if pf else None, self._read(ff) if ff else None))
            time.sleep(0.05)

    def start(self):
        import threading
        if self.candidates:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop = True
        self._thread.join()

        def avg_power(c):
            pw = [p for p, _ in self.samples[c] if p]
            return sum(pw) / len(pw) if pw else 0.0
        best = max(self.candidates, key=avg_power)
        pw = [p for p, _ in self.samples[best] if p]
        ck = [c for _, c in self.samples[best] if c]
        capf = self._file(best, ("power1_cap",))
        cap = self._read(capf) if capf else None
        return {"samples": len(self.samples[best]), "package_power_w_avg": round(sum(pw) / len(pw) / 1e6, 1) if pw else None,
                "package_power_w_max": round(max(pw) / 1e6, 1) if pw else None, "power_cap_w": round(cap / 1e6, 1) if cap else None,
                "sclk_ghz_avg": round(sum(ck) / len(ck) / 1e9, 3) if ck else None, "sclk_ghz_min": round(min(ck) / 1e9, 3) if ck else None,
                "device_matched_by": self.matched_by, "source": "amdgpu hwmon sysfs, 50 ms period, one profiled step"}

PAD_ID = 51864


def train_flops_per_sample(d, L, T=1500, S=448, V=51865, F=3000, n_mels=80):
    """Algorithmic dense FLOPs of one training sample = 3 x forward (SURVEY.md section 2.2 / BASELINE.md section 2)."""
    conv = 2 * F * n_mels * 3 * d + 2 * T * d * 3 * d
    enc_layer = 24 * T * d * d + 4 * T * T * d
    dec_layer = (8 * S * d * d + 4 * S * S * d) + (4 * S * d * d + 4 * T * d * d + 4 * S * T * d) + 16 * S * d * d
    logits = 2 * S * d * V
    return 3 * (conv + L * enc_layer + L * dec_layer + logits)


STEP_MODES = ("span-forward", "span-backward", "plain")


def decoder_rows(mode, spans, S=448):
    """(forward rows, backward rows) of the decoder's token-row matrices for one batch under each step mode.  `spans` = per-sample
    supervised span (one past the last position that can carry gradient); the span steps round it up to whole 64-position chunks.
      plain          the reference's computation shape: every sample padded to S positions, forward and backward
      span-backward  forward over S positions, backward over the chunks that can carry gradient (round 4's step)
      span-forward   forward AND backward over those chunks (the padded positions' logits are nobody's output) -- the default
    A batch whose every span reaches the last chunk (e.g. every text_len = 447) makes all three the same."""
    assert mode in STEP_MODES, mode
    B = len(spans)
    act = sum((int(x) + 63) // 64 * 64 for x in spans)
    act = min(act, B * S)
    return (B * S if mode != "span-forward" else act), (B * S if mode == "plain" else act)


def executed_flops_per_sample(d, L, bwd_rows_mean, fwd_rows_mean=448, T=1500, S=448, V=51865, F=3000, n_mels=80):
    """FLOPs a step actually executes per sample: the decoder's forward over `fwd_rows_mean` token rows per clip, its backward (2 x
    forward) over `bwd_rows_mean` (decoder_rows() / B).  Attention of the decoder: the query side shrinks with the rows (self-attention:
    keys too); the encoder side and the cross-attention K / V projections always run in full."""
    conv = 2 * F * n_mels * 3 * d + 2 * T * d * 3 * d
    enc_layer = 24 * T * d * d + 4 * T * T * d

    def dec(Sq):
        return (8 * Sq * d * d + 4 * Sq * Sq * d) + (4 * Sq * d * d + 4 * T * d * d + 4 * Sq * T * d) + 16 * Sq * d * d
    fwd = conv + L * enc_layer + L * dec(fwd_rows_mean) + 2 * fwd_rows_mean * d * V
    bwd = 2 * (conv + L * enc_layer + L * dec(bwd_rows_mean) + 2 * bwd_rows_mean * d * V)
    return fwd + bwd


def spread(ms):
    """min / median / max of a list of per-step milliseconds (HIP events between the steps of the timed region)."""
    import statistics
    return {"min": round(min(ms), 2), "median": round(statistics.median(ms), 2), "max": round(max(ms), 2), "n": len(ms)}


def synth_batch(indices, device):
    """SURVEY.md section 8(d)'s generator, sample by sample on the host (olmoasr_amd/synth.py: seed 1234 + sample index, bit-equal
    to oracle.model_oracle.synthetic_sample, tests/test_synth_cpu.py): int16 PCM ~ round(clip(N(0, 0.1)) * 32767) with a zeroed
    tail of U{0..240000} samples, tokens [sot, notimestamps, body..., eot] padded with 51864 to 448.  Runs before the timed
    region; the clips are resident in HBM when it starts."""
    from concurrent.futures import ThreadPoolExecutor
    from olmoasr_amd.synth import synth_sample
    with ThreadPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)))) as pool:
        items = list(pool.map(synth_sample, [int(i) for i in indices]))
    pcm = torch.stack([it[0] for it in items]).to(device)
    ti = torch.stack([it[1] for it in items]).to(device)
    ty = torch.stack([it[2] for it in items]).to(device)
    tl = torch.tensor([it[3] for it in items], dtype=torch.int32).to(device)
    return pcm, ti, ty, tl


def _flush_c_stdio():
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def host_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_baseline(variant, sd, budget_s=75.0):
    """The CPU oracle (oracle/model_oracle.py == the reference's algorithm, pinned by tests/golden) timed on this host's
    cores: ONE 30 s clip through the full step (log-mel + forward + CE + backward + clip + AdamW) with the SAME weights the
    GPU model holds, fp32 and under the reference's autocast(bfloat16) rounding points; per dtype 1 warm-up run, then the
    median of up to 3 timed runs; repeats stop early once `budget_s` of CPU time is spent so the
    default bench run stays within minutes.  Returns (json block, fp32 loss, fp32 logits of the clip) -- the last two feed
    the `parity` block."""
    import statistics

    import numpy as np
    from oracle import mel_oracle as me
    from oracle import model_oracle as mo
    cores = host_cores()
    torch.set_num_threads(cores)
    dims = mo.VARIANTS[variant]
    pcm, ti, ty, tl = mo.synthetic_batch([0])
    spent = [0.0]
    keep = {}

    # Where the reference tree is readable (the build container; never the GPU box) the UNMODIFIED reference module is what gets timed:
    # olmoasr.model.OLMoASR + F.cross_entropy(ignore_index) + clip_grad_norm_ + torch AdamW, the lines of train_timestamps.py:1440-1454,
    # 1509-1512 (kind "reference"; the log-mel in front of it stays the oracle's: whisper.audio is third party and absent).  Otherwise the
    # oracle restatement (kind "port").
    from oracle import ref_import
    use_ref = ref_import.available()
    ref_model = ref_dims = None
    if use_ref:
        ref_model, _, ref_dims = ref_import.load()

    def one_reference(bf16):
        net = ref_model.OLMoASR(ref_dims.ModelDimensions(**dims.__dict__))
        net.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        opt = torch.optim.AdamW(net.parameters(), lr=1.5e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
        t0 = time.time()
        mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
        pm = mo.build_padding_mask(tl)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=bf16):
            logits = net(mel, ti, pm)
        loss = torch.nn.functional.cross_entropy(logits.float().view(-1, logits.shape[-1]), ty.view(-1), ignore_index=mo.PAD_ID)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        dt = time.time() - t0
        spent[0] += dt
        if not bf16 and "loss" not in keep:
            keep["loss"], keep["logits"] = float(loss), logits.detach().float()
        return dt

    def one_port(bf16):
        w = {k: v.clone() for k, v in sd.items()}
        t0 = time.time()
        mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
        loss, grads, logits = mo.loss_and_grads(w, dims, mel, ti, ty, tl, autocast_bf16=bf16)
        _, coef = mo.clip_coef(grads, 1.0)
        names = list(grads)
        for n in names:
            grads[n].mul_(coef)
        m = {n: torch.zeros_like(w[n]) for n in names}
        v = {n: torch.zeros_like(w[n]) for n in names}
        mo.adamw_step(w, grads, m, v, step=1, lr=1.5e-3)
        dt = time.time() - t0
        spent[0] += dt
        if not bf16 and "loss" not in keep:
            keep["loss"], keep["logits"] = float(loss), logits
        return dt

    one = one_reference if use_ref else one_port

    res = {}
    for tag, bf16 in (("fp32", False), ("autocast_bf16", True)):
        one(bf16)  # warm-up (thread pool, oneDNN primitive caches, page faults of the 3 GB of weights + state)
        times = [one(bf16)]
        while len(times) < 3 and spent[0] < budget_s:
            times.append(one(bf16))
        res[tag] = {"audio_s_per_s": round(30.0 / statistics.median(times), 3), "median_s": round(statistics.median(times), 2),
                    "runs": len(times)}
    out = {"value": res["fp32"]["audio_s_per_s"], "unit": "audio-seconds/sec", "cores": cores, "kind": "reference" if use_ref else "port",
           "fp32": res["fp32"], "autocast_bf16": res["autocast_bf16"],
           "sample": f"1 clip x 30 s, OLMoASR-{variant} full step (log-mel + fwd + CE + bwd + clip + AdamW), torch "
                     f"{torch.get_num_threads()} threads; per dtype 1 warm-up + median of the timed runs; value = fp32 (the "
                     f"reference's CPU default); " + ("the UNMODIFIED reference module (olmoasr.model.OLMoASR from the mounted reference tree) "
                                                       "behind the oracle's log-mel" if use_ref else
                                                       "/root/reference is not on the GPU box, so the restatement (pinned to it by tests/) is what is timed")}
    return out, keep["loss"], keep["logits"]


def hbm_kernel_rooflines(net, dims, mb, dev):
    """Achieved HBM GB/s of the bandwidth-bound kernels at the benchmarked shapes (torch events on the launch stream,
    5 launches each after a warm-up) against the algorithmic bytes DESIGN.md section 3 states for them."""
    from olmoasr_amd import ops
    d = dims.n_audio_state
    rows = mb * dims.n_audio_ctx
    BF = torch.bfloat16
    out = {}

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps * 1e-3

    def line(name, t, nbytes, note):
        out[name] = {"achieved_GBps": round(nbytes / t / 1e9, 1), "frac_of_8TBps": round(nbytes / t / 8e12, 3), "us": round(t * 1e6, 1),
                     "alg_bytes": int(nbytes), "shape": note}

    x = torch.randn(rows, d, device=dev).to(BF)
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    line("ln_fwd_kernel", timed(lambda: ops.layernorm_fwd(x, gamma, beta)), rows * (4 * d + 8), f"[{rows}, {d}] bf16 in/out + fp32 stats")
    dy = torch.randn(rows, d, device=dev).to(BF)
    line("ln_bwd_kernel", timed(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dy)), rows * (8 * d + 8),
         f"[{rows}, {d}]: dy, x, dres in, dx out (+ stats)")
    del x, y, dy
    pcm = torch.zeros(mb, 480000, dtype=torch.int16, device=dev).random_(-3000, 3000)
    line("logmel_fft (round 6: the quad-lane register kernel logmel_quad; floor fused into the encoder's transpose)", timed(lambda: ops.log_mel(pcm, finalize=False)), mb * 1.92e6,
         f"{mb} clips: int16 PCM in + fp32 [80,3000] log10 mel power + per-clip maximum out (what the timed step runs)")
    line("logmel_fft+finalize (whisper.audio.log_mel_spectrogram's own output)", timed(lambda: ops.log_mel(pcm)), mb * 1.92e6,
         f"{mb} clips: the same + the in-place floor / scale pass (3.84 MB per clip actually move)")
    del pcm
    Vp, V = (dims.n_vocab + 1 + 127) // 128 * 128, dims.n_vocab + 1
    rows_d = mb * dims.n_text_ctx
    lg = torch.randn(rows_d, Vp, device=dev, dtype=BF)
    tgt = torch.randint(0, 50000, (rows_d,), device=dev)
    line("ce_kernel", timed(lambda: ops.cross_entropy_(lg, V, tgt, PAD_ID)), rows_d * 4 * Vp, f"[{rows_d}, {Vp}] bf16 logits -> dlogits in place")
    del lg
    # KV-cached greedy decode step (an HBM-bound sub-path): one new token for ONE sequence streams every decoder weight
    # once (bf16: 14 d^2 per layer + the tied logits matrix), the cross-attention K/V of every layer and the self-attention cache so far
    L_dec, S_pos = dims.n_text_layer, 32
    xa = torch.randn(1, dims.n_audio_ctx, d, device=dev).to(BF)
    state = net.kv_cache_begin(xa)
    tok = torch.tensor([50257], device=dev)
    for _ in range(S_pos):  # fill positions 0 .. S_pos-1, then time steps at a fixed position (rewinding the cursor: same work each time)
        net.kv_cache_step(state, tok)

    def dstep():
        state["pos"] = S_pos
        net.kv_cache_step(state, tok)
    dbytes = 2 * (L_dec * 14 * d * d + V * d) + 2 * L_dec * (dims.n_audio_ctx * 2 * d + S_pos * 2 * d) + 4 * V
    line("decode_step(B=1)", timed(dstep, reps=20), dbytes,
         f"one KV-cached decoder step of this model at position {S_pos}: bf16 decoder weights + tied logits matrix once, cross K/V [{dims.n_audio_ctx}, {2 * d}] x {L_dec} layers, self K/V so far; "
         f"round 6: ONE persistent launch for the decoder stack on every CU (csrc/decode_wide.hip: a few weight rows per workgroup, fp32 FMA dot products, "
         f"{8 * L_dec} exchanges through per-workgroup phase flags) + embedding + logits launches; bound by the helper wave's dependent-instruction chain per phase "
         f"(~3-4 us against ~1.3 us of memory round trips, profiles/r06_decode_wide.txt); round 5's one-XCD team (csrc/decode_xcd.hip) measured 0.066-0.069, the multi-launch step 0.049")
    del state, xa
    n = net.flat_params.numel()
    # (step count irrelevant for the timing; gradients are whatever the last step left, state is restored by nobody: run last)
    line("grad_stats+adamw_kernel", timed(lambda: net.optim_step(step=1, lr=0.0), reps=3), n * (28 + 2 + 4),
         f"{n} params: 16 B read + 12 B written + 2 B bf16 shadow + 4 B norm pass")
    return out


def self_launch(n, argv, module="torch.distributed.run", extra_env=None):
    """`python bench.py --gpus N` (N > 1) outside a launcher: re-run this file as N ranks on this node and relay rank 0's stdout.
    Fails with a plain message -- not an assert -- when the node has fewer devices."""
    import socket
    import subprocess
    n_dev = n if os.environ.get("OASR_BENCH_STUB") == "1" else torch.cuda.device_count()
    if n_dev < n:
        print(f"bench.py: --gpus {n} needs {n} devices on this node, {n_dev} visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:  # a free loopback port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", module, "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if extra_env:
        env.update(extra_env)
    return subprocess.call(cmd, env=env)


def stub_main(args):
    """OASR_BENCH_STUB=1: everything bench.py does AROUND the step -- rank environment, process group, warm-up, barrier-bracketed
    timing of exactly K steps, max / min over ranks, the `ddp` block of the line, `--reducer both`, one JSON line from rank 0 -- with a
    stand-in step (the real GradReducer over gloo on a small CPU arena), so the multi-rank launch path is covered on a machine without
    GPUs.  Never a measurement."""
    from olmoasr_amd import ddp
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.ones(4096)
    segs = [(0, 1024), (1024, 2048), (3072, 1024)]
    algos = ("allreduce", "direct") if args.reducer == "both" else (args.reducer,)
    reducers = {a: ddp.GradReducer(buf, segs, bucket_cap_mb=args.bucket_mb, algo=a, timing=True) for a in algos} if world > 1 else {}

    def one_step(algo):
        if world > 1:
            reducers[algo].reduce()
            buf.div_(world)
        time.sleep(0.002)

    def timed(algo):
        for _ in range(args.warmup):
            one_step(algo)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        per = []
        for _ in range(args.steps):
            t1 = time.perf_counter()
            one_step(algo)
            per.append(1000.0 * (time.perf_counter() - t1))
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        lo = hi = el
        if world > 1:
            t = torch.tensor([el, -el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hi, lo = float(t[0]), float(-t[1])
        return hi, lo, per
    elapsed, lo, per = timed(algos[0])
    block = None
    if world > 1:
        block = {"reducer": algos[0], "bucket_mb": args.bucket_mb, "buckets": len(reducers[algos[0]].buckets),
                 "rank_step_ms_min": round(1000 * lo / args.steps, 3), "rank_step_ms_max": round(1000 * elapsed / args.steps, 3),
                 **reducers[algos[0]].comm_report()}
        if len(algos) > 1:
            el2, _, per2 = timed(algos[1])
            block["reducer_ab"] = {algos[0]: {"ms_per_step": round(1000 * elapsed / args.steps, 3)},
                                   algos[1]: {"ms_per_step": round(1000 * el2 / args.steps, 3), "per_step_ms": spread(per2), **reducers[algos[1]].comm_report()}}
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(world * args.steps / elapsed, 3), "unit": "steps/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3),
                          "per_step_ms": spread(per), "ddp": block,
                          "data": "stub (OASR_BENCH_STUB=1: launch-path rehearsal on CPU, not a measurement)",
                          "buf_ok": bool(float(buf[0]) == 1.0)}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--variant", default="medium")
    ap.add_argument("--per-gpu-batch", type=int, default=256)
    ap.add_argument("--micro-batch", type=int, default=0,
                    help="clips per micro-batch (gradient accumulation over per-gpu-batch / micro-batch); 0 = the largest of "
                         "128/64/32/... whose saved activations fit in free HBM with --hbm-margin-gib to spare (the SAME rule at every N)")
    ap.add_argument("--hbm-margin-gib", type=float, default=16.0,
                    help="HBM left free beside the saved-activation workspace when the micro-batch is chosen: transient tensors of a step "
                         "(< 1 GiB) + RCCL's channel buffers at N > 1 (~1-2 GiB); one world-independent number so N = 1 and N = 8 pick the same micro-batch")
    ap.add_argument("--traffic-json", default=next((p for p in (os.path.join(ROOT, "profiles", f) for f in ("r06_hbm_traffic.json", "r05_hbm_traffic.json"))
                                                    if os.path.exists(p)), os.path.join(ROOT, "profiles", "r06_hbm_traffic.json")),
                    help="per-symbol HBM bytes per launch from the rocprofv3 PMC passes (scripts/pmc_traffic.py)")
    ap.add_argument("--bucket-mb", type=float, default=128.0)
    ap.add_argument("--reducer", default="allreduce", choices=["allreduce", "direct", "both"],
                    help="gradient exchange per bucket: one RCCL all-reduce, or reduce-scatter + all-gather in place (all xGMI links); "
                         "'both' times the two back to back in one launch (headline = allreduce, the other under `reducer_ab`)")
    ap.add_argument("--trim-padding", action="store_true",
                    help="opt-in, NOT the reference's computation shape: run the decoder over ceil16(max text_len) of each "
                         "micro-batch instead of the padded 448 positions (same loss and gradients; see DESIGN.md)")
    ap.add_argument("--reference-shape", "--full-backward", dest="reference_shape", action="store_true",
                    help="headline = the PLAIN step: decoder forward and backward over all 448 padded positions, the reference's computation shape")
    ap.add_argument("--span-backward-only", action="store_true",
                    help="headline = round 4's step: decoder forward over all 448 positions, backward over the supervised span")
    ap.add_argument("--span-forward", action="store_true", help="(the default since round 5; accepted for old command lines)")
    ap.add_argument("--ab-steps", type=int, default=-1,
                    help="steps timed for each of the two NON-headline step modes in the same process (1 warm-up each); -1 = min(steps, 3), 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves (torch.distributed.run, loopback rendezvous)
    # and let rank 0's JSON line through -- the shape of the driver's N = 1 command works for every N.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])

    if os.environ.get("OASR_BENCH_STUB") == "1":  # tests/test_ddp_cpu.py: the launch / rendezvous / rank-0-prints plumbing on CPU (gloo)
        return stub_main(args)

    from olmoasr_amd import _native as N
    from olmoasr_amd import ddp, ops
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    from olmoasr_amd.model import OLMoASR

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices on this node, {n_dev} visible (rank {rank})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # OASR_BENCH_FORCE_DDP=1: take the multi-GPU code path (RCCL group, broadcast, event-driven bucket all-reduce, barriers) with
    # a world of one -- a single-GPU rehearsal of what the N > 1 launches execute
    ddp_path = world > 1 or os.environ.get("OASR_BENCH_FORCE_DDP") == "1"
    if ddp_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, pg_options=ddp.rccl_options())  # "nccl" is RCCL on ROCm

    dims = VARIANT_TO_DIMS[args.variant]
    net = OLMoASR(dims, device=dev, seed=0)
    ddp.broadcast_parameters(net.flat_params)
    net.refresh_shadow()
    net.init_optimizer_state()
    reducers = {}
    if ddp_path:
        for algo in (("allreduce", "direct") if args.reducer == "both" else (args.reducer,)):
            reducers[algo] = ddp.GradReducer(net.flat_grads, net.grad_segments, bucket_cap_mb=args.bucket_mb, algo=algo, force=world == 1, timing=True)
    head_algo = "allreduce" if args.reducer == "both" else args.reducer

    B = args.per_gpu_batch
    free_gib = torch.cuda.mem_get_info(dev)[0] / 2**30
    ws_gib = lambda cand: N.lib().oasr_workspace_bytes(net._ctx, cand, dims.n_text_ctx, 1) / 2**30  # noqa: E731
    if args.micro_batch > 0:
        mb, mb_rule = min(args.micro_batch, B), "--micro-batch"
    else:  # no-recompute training keeps ~1.9 GiB of activations per medium clip: 128 clips = 240 GiB of the 288
        # ONE margin for every world size (round 4 added 8 GiB at N > 1, which put the 8-GPU choice within 1 GiB of flipping to 64 and
        # would have booked a micro-batch change as scaling loss): the candidates are tried in the same order against the same rule
        mb = 1
        for cand in (128, 64, 32, 16, 8, 4, 2):
            if cand <= B and B % cand == 0 and ws_gib(cand) + args.hbm_margin_gib <= free_gib:
                mb = cand
                break
        mb_rule = f"largest of 128/64/.. with workspace + {args.hbm_margin_gib:g} GiB <= free HBM"
        if ddp_path:  # same choice on every rank
            t = torch.tensor([mb], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            mb = int(t)
    assert B % mb == 0
    accum = B // mb
    mb_pref = next((c for c in (128, 64, 32, 16, 8, 4, 2, 1) if c <= B and B % c == 0), 1)
    if mb < mb_pref and args.micro_batch <= 0 and rank == 0:  # loud: a smaller micro-batch is ~3 % of the step and must not pass for scaling loss
        print(f"bench.py: WARNING micro-batch {mb} < {mb_pref}: free HBM {free_gib:.1f} GiB < workspace {ws_gib(mb_pref):.1f} + margin "
              f"{args.hbm_margin_gib:g} GiB; lines at different N are comparable only at equal micro_batch", file=sys.stderr, flush=True)
    # sample i of the global batch goes to rank i % world (DistributedSampler rule, shuffle off)
    pcm, ti, ty, tl = synth_batch(range(rank, B * world, world), dev)
    loss_buf = torch.zeros(1, device=dev)
    loss_scale = 65536.0  # GradScaler() initial scale; the reference keeps it enabled for bf16 (train_timestamps.py:2349)
    state = {"step": 0}

    # (the data loader knows the token counts on the host; one sync here, outside the timed region)
    ctx = [None] * accum
    if args.trim_padding:
        ctx = [min(448, (int(tl[i * mb:(i + 1) * mb].max()) + 15) // 16 * 16) for i in range(accum)]
    # supervised span per sample, on the HOST like the loaders hand it out (olmoasr_amd/synth.py::supervised_span_host,
    # train_timestamps.py:238-343 builds the sequences there): one past the last position whose target is not the ignore index
    sp = OLMoASR.supervised_span(ty, tl)
    spans = [sp[i * mb:(i + 1) * mb].contiguous() for i in range(accum)]
    head_mode = "plain" if (args.reference_shape or args.trim_padding) else ("span-backward" if args.span_backward_only else "span-forward")
    rows = {m: decoder_rows(m, sp.tolist(), dims.n_text_ctx) for m in STEP_MODES}

    def one_step(mode=head_mode, reducer_name=head_algo):
        reducer = reducers.get(reducer_name)
        state["step"] += 1
        net.zero_grad()
        for i in range(accum):
            sl = slice(i * mb, (i + 1) * mb)
            last = i == accum - 1
            seg = reducer.segment_events() if (reducer and last) else None
            if mode == "plain":
                net.loss_and_backward(ops.log_mel(pcm[sl]), ti[sl], ty[sl], tl[sl], loss_scale=loss_scale, accumulation_steps=accum,
                                      loss_out=loss_buf, accumulate_loss=i > 0, segment_events=seg, text_ctx=ctx[i])
            else:
                # (span steps: whisper's floor-at-max-minus-8 / (x + 4) / 4 lines ride in the encoder's time-major transpose instead of
                # a second pass over the log-mel tensor)
                mel, clip_max = ops.log_mel(pcm[sl], finalize=False)
                net.loss_and_backward(mel, ti[sl], ty[sl], tl[sl], loss_scale=loss_scale, accumulation_steps=accum, loss_out=loss_buf,
                                      accumulate_loss=i > 0, segment_events=seg, span=spans[i], span_forward=mode == "span-forward",
                                      mel_clip_max=clip_max)
        div = 1.0
        if reducer:
            reducer.reduce()
            div = reducer.grad_divisor
        net.optim_step(step=state["step"], lr=1.5e-4, inv_loss_scale=1.0 / (loss_scale * div), max_grad_norm=1.0)

    def barrier():
        torch.cuda.synchronize(dev)
        if ddp_path:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(n_warm, n_steps, **kw):
        """n_warm untimed steps, then EXACTLY n_steps between barrier + synchronize on both sides (wall clock, max over ranks);
        one HIP event between the steps gives the per-step spread of this rank without a host synchronisation inside the region."""
        for _ in range(n_warm):
            one_step(**kw)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
        barrier()
        t0 = time.perf_counter()
        for k in range(n_steps):
            evs[k].record()
            one_step(**kw)
        evs[n_steps].record()
        barrier()
        el = time.perf_counter() - t0
        per = [evs[k].elapsed_time(evs[k + 1]) for k in range(n_steps)]
        rank_el = [el, el]
        if ddp_path:
            t = torch.tensor([el, -el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, rank_el = float(t[0]), [float(-t[1]), float(t[0])]
        return el, per, rank_el

    elapsed, per_step_ms, rank_el = timed(args.warmup, args.steps)
    final_loss = float(loss_buf)
    found_inf = float(net._opt_stats[1])
    comm = None
    if ddp_path:
        comm = reducers[head_algo].comm_report()  # of the LAST timed step (HIP events recorded inside reduce())

    # ---- the other two step modes, a few steps each, SAME process / box / clock regime (outside the headline's timed region):
    # the exact A/B the headline rests on travels with the line instead of living in a builder-run profile
    ab = {}
    n_ab = min(args.steps, 3) if args.ab_steps < 0 else args.ab_steps
    if n_ab > 0 and not args.trim_padding:
        for m in STEP_MODES:
            if m != head_mode:
                el_m, per_m, _ = timed(1, n_ab, mode=m)
                ab[m] = {"ms_per_step": round(1000.0 * el_m / n_ab, 2), "steps": n_ab, "per_step_ms": spread(per_m), "final_loss": round(float(loss_buf), 4)}
    reducer_ab = None
    if args.reducer == "both" and ddp_path:
        reducer_ab = {"allreduce": {"ms_per_step": round(1000.0 * elapsed / args.steps, 2), **(comm or {})}}
        el_d, per_d, _ = timed(1, args.steps, reducer_name="direct")
        reducer_ab["direct"] = {"ms_per_step": round(1000.0 * el_d / args.steps, 2), "per_step_ms": spread(per_d), **reducers["direct"].comm_report()}

    # ---- live roofline of the dominant kernel: one more identical step with every GEMM launch bracketed by HIP events
    # on the launch stream (olmoasr_amd/csrc/gemm.hip); aggregated by kernel SYMBOL so it can be compared line by line
    # with `rocprofv3 --kernel-trace --stats` of this same command (profiles/).
    roof = None
    if not args.no_profile:
        lib = N.lib()
        lib.oasr_profile_gemm(1)
        sampler = PowerSampler(local_rank if ddp_path else 0)
        sampler.start()
        one_step()
        torch.cuda.synchronize(dev)
        power = sampler.stop()
        ms = (ctypes.c_double * 4)()
        fl = (ctypes.c_double * 4)()
        cnt = (ctypes.c_int64 * 4)()
        buf = ctypes.create_string_buffer(65536)
        N.check(lib.oasr_profile_gemm_collect(ms, fl, cnt, buf, 65536), "profile_collect")
        lib.oasr_profile_gemm(0)
        sym = {}
        for line in buf.value.decode().strip().split("\n"):
            if line:
                name, n, t, f = line.split("\t")
                sym[name] = {"launches": int(n), "ms": float(t), "flops": float(f)}
        # launches made on the span step's lowest-priority side streams are reported apart ("symbol [side]", csrc/gemm.hip GemmProfile::lane):
        # their begin-to-end spans are queueing times.  Main-stream launches of the decoder phases, which share the chip with that filler
        # ("symbol [shared]"), give up compute units to it and run longer for it.  The dominant kernel and its roofline are taken over the
        # launches that have the chip to themselves; every row is in by_symbol, and main_stream_all gives the symbol's whole main-stream average
        main_sym = {k: v for k, v in sym.items() if not k.endswith("]")}
        dom = max(main_sym, key=lambda k: main_sym[k]["ms"])
        dsym = main_sym[dom]
        shared = sym.get(dom + " [shared]")
        main_all = {"launches": dsym["launches"] + (shared["launches"] if shared else 0), "ms": dsym["ms"] + (shared["ms"] if shared else 0.0),
                    "flops": dsym["flops"] + (shared["flops"] if shared else 0.0)}
        achieved = dsym["flops"] / dsym["ms"] / 1e9
        traffic = traffic_src = None
        if args.traffic_json and os.path.exists(args.traffic_json):  # HBM bytes per launch from the PMC passes (profiles/)
            tj = json.load(open(args.traffic_json))
            traffic = tj.get(dom, {}).get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_src = (f"REPLAYED from {os.path.relpath(args.traffic_json, ROOT)} (separate rocprofv3 --pmc passes of this "
                               f"command, scripts/pmc_traffic.py; {tj.get('_meta', {}).get('collected', 'date not recorded')}) -- NOT measured in this run")
        layouts = {0: "NT (forward)", 1: "NN (dgrad)", 2: "TN", 3: "TN (wgrad)"}
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": round(1000.0 * dsym["ms"] / dsym["launches"], 2), "launches_per_step": dsym["launches"],
                "alg_flops_per_launch": round(dsym["flops"] / dsym["launches"], 1),
                "by_symbol": {k: {"launches": v["launches"], "avg_us": round(1000.0 * v["ms"] / v["launches"], 2),
                                  "tflops": round(v["flops"] / v["ms"] / 1e9, 1)} for k, v in sorted(sym.items(), key=lambda kv: -kv[1]["ms"])},
                "by_symbol_note": "'[side]' = launches on the span step's lowest-priority side streams: begin-to-end spans that include waiting for "
                                  "compute units (queueing times, overlapping the main stream in wall time); '[shared]' = main-stream launches of the decoder "
                                  "phases, running beside that filler; `kernel` / `frac` / `avg_launch_us` = the symbol's launches with the chip to themselves",
                "main_stream_all": {"launches": main_all["launches"], "avg_us": round(1000.0 * main_all["ms"] / main_all["launches"], 2),
                                    "tflops": round(main_all["flops"] / main_all["ms"] / 1e9, 1),
                                    "frac": round(main_all["flops"] / main_all["ms"] / 1e9 / PEAK_BF16_TFLOPS, 4)},
                "by_layout_tflops": {layouts[k]: round(fl[k] / ms[k] / 1e9, 1) for k in range(4) if cnt[k]},
                "gemm_ms_per_step": round(sum(ms), 2)}
        # where the board actually was during the profiled step (live hwmon sample).  Reference points that were NOT measured in this
        # run -- the package-cap ceilings of a dense bf16 GEMM on these boxes -- are in profiles/ (r02_gemm_vs_vendor.txt: vendor library
        # 1484 TFLOP/s at 1400 W / 1.83 GHz, this repo's layer GEMM 1307 at 1.65 GHz) and are not replayed into this line.
        roof["power_limited"] = {"sampled_during_profiled_step": power}

    out = None
    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        value = world * B * 30.0 * args.steps / elapsed
        d_, L_ = dims.n_audio_state, dims.n_audio_layer
        fl_sample = sum(train_flops_per_sample(d_, L_, S=(c or 448)) for c in ctx) / len(ctx)
        step_tflops = B * fl_sample / (elapsed / args.steps) / 1e12  # ALGORITHMIC flops (3 x forward over the padded context) per second

        def fl_exec_of(mode):
            if args.trim_padding:
                return fl_sample
            fr, br = rows[mode]
            return executed_flops_per_sample(d_, L_, br / B, fr / B)
        fl_exec = fl_exec_of(head_mode)
        shape_txt = {"plain": "448 (padded, as the reference), forward and backward",
                     "span-backward": "448 forward (padded, as the reference); backward over the supervised span",
                     "span-forward": "forward AND backward over the supervised span (the padded positions' logits are computed by the reference "
                                     "and read by nothing: train_timestamps.py:1440-1450 sees them only through ignore_index)"}[head_mode]
        if head_mode != "plain":
            shape_txt += f": mean {rows[head_mode][1] / B:.1f} of 448 token rows per clip (whole 64-position chunks; exact: the rows left out are zeros)"
        if args.trim_padding:
            shape_txt = "trimmed to ceil16(max text_len) per micro-batch: %s (opt-in, not the reference shape)" % ctx
        out = {
            "metric": "audio-seconds/sec/node (train step)", "value": round(value, 1), "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (SURVEY 8d generator: sample i = seed 1234 + i on the host, sample i -> rank i mod world; random-init weights)",
            "config": {"workload": f"OLMoASR-{args.variant} bf16 train step, {B} x 30 s synthetic clips per GPU "
                                   f"({accum} micro-batches of {mb}, grad accumulation), global batch {world * B}",
                       "global_batch": world * B, "micro_batch": mb, "micro_batch_rule": mb_rule, "micro_batch_auto_reduced": bool(mb < mb_pref and args.micro_batch <= 0),
                       "free_hbm_gib": round(free_gib, 2), "workspace_gib": round(ws_gib(mb), 2), "hbm_margin_gib": args.hbm_margin_gib,
                       "parallelism": f"dp{world}" + (f" ({head_algo} gradient exchange, {args.bucket_mb:g} MiB buckets)" if world > 1 else ""),
                       "optimizer": "AdamW fused (unscale+clip+step), loss scale 65536",
                       "step_mode": head_mode, "decoder_positions": shape_txt,
                       # span steps: which GEMMs run on the engine's lowest-priority side streams (csrc/engine.hip Runner::side_mode; bit 0: the decoder
                       # backward's weight gradients over the active rows, bit 2: the cross-attention key|value gradients).  Their HIP-event / rocprof
                       # durations are begin-to-end spans that include waiting for compute units: by_symbol sums overlap in wall time
                       "side_streams": int(N.lib().oasr_span_side_streams()) if head_mode != "plain" else 0},
            # per-step spread of the timed region on rank 0 (HIP events between the steps): a 1 % kernel change vs box noise
            "per_step_ms": spread(per_step_ms),
            # the same process timed the other step modes right after the headline (a few steps each): what the headline's shape buys
            "plain_step_ms": ab.get("plain", {}).get("ms_per_step") if head_mode != "plain" else round(ms_per_step, 2),
            "span_bwd_ms": ab.get("span-backward", {}).get("ms_per_step") if head_mode != "span-backward" else round(ms_per_step, 2),
            "span_fwd_ms": ab.get("span-forward", {}).get("ms_per_step") if head_mode != "span-forward" else round(ms_per_step, 2),
            "step_modes_same_run": ab,
            # TWO fractions, to be read together.  *_algorithmic prices the reference's work (3 x forward over 448 padded positions =
            # SURVEY 8(d)'s FLOPs/sample) -- the MFU-style number the north_star target is written in; *_executed prices only the
            # multiplications this step really performs (rows that are exact zeros in the reference are not multiplied)
            "step_model_tflops_per_gpu": round(step_tflops, 1),
            "step_frac_of_mfma_peak": round(step_tflops / PEAK_BF16_TFLOPS, 4),
            "step_frac_algorithmic": round(step_tflops / PEAK_BF16_TFLOPS, 4),
            "step_frac_executed": round(B * fl_exec / (elapsed / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "executed_over_algorithmic_flops": round(fl_exec / fl_sample, 4),
            "step_executed_tflops_per_gpu": round(B * fl_exec / (elapsed / args.steps) / 1e12, 1),
            "decoder_rows_fwd_bwd": {m: list(rows[m]) for m in STEP_MODES},
            "final_loss": round(final_loss, 4), "found_inf": found_inf,
        }
        for m, r in ab.items():
            r["step_frac_executed"] = round(B * fl_exec_of(m) / (r["ms_per_step"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
        # (the driver's record keeps `config` and `roofline` whole and only NAMES the other keys: the numbers a reader needs next to the
        # headline -- what the reference-shape step costs in the same run, and what the matrix cores really multiply -- ride in both)
        out["config"].update({"plain_step_ms": out["plain_step_ms"], "span_bwd_ms": out["span_bwd_ms"],
                              "step_frac_algorithmic": out["step_frac_algorithmic"], "step_frac_executed": out["step_frac_executed"],
                              "executed_over_algorithmic_flops": out["executed_over_algorithmic_flops"]})
        if roof:
            roof.update({"step_frac_algorithmic": out["step_frac_algorithmic"], "step_frac_executed": out["step_frac_executed"],
                         "plain_step_ms": out["plain_step_ms"], "ms_per_step": out["ms_per_step"]})
        if ddp_path:
            # what a multi-rank line needs to diagnose itself (DESIGN section 6 says which key answers which question)
            out["ddp"] = {"reducer": head_algo, "bucket_mb": args.bucket_mb, "buckets": len(reducers[head_algo].buckets),
                          "rank_step_ms_min": round(1000.0 * rank_el[0] / args.steps, 2), "rank_step_ms_max": round(1000.0 * rank_el[1] / args.steps, 2),
                          **(comm or {})}
            if reducer_ab:
                out["ddp"]["reducer_ab"] = reducer_ab
        if roof:
            out["roofline"] = roof
        span_check = None
        if world == 1 and head_mode != "plain" and not args.no_profile:
            # the step that was timed against the step it replaces, at the benchmarked micro-batch, on this model's weights (outside the
            # timed region): micro-batch 0 through the plain step (decoder forward + backward over all 448 positions) and through the headline step
            sl = slice(0, mb)
            net.zero_grad()
            l_full, _ = net.loss_and_backward(ops.log_mel(pcm[sl]), ti[sl], ty[sl], tl[sl], loss_scale=loss_scale)
            g_full = net.flat_grads.clone()
            net.zero_grad()
            m_raw, m_max = ops.log_mel(pcm[sl], finalize=False)
            l_span, _ = net.loss_and_backward(m_raw, ti[sl], ty[sl], tl[sl], loss_scale=loss_scale, span=spans[0], span_forward=head_mode == "span-forward",
                                              mel_clip_max=m_max)
            torch.cuda.synchronize(dev)
            span_check = {"what": f"micro-batch 0 ({mb} clips) of the timed step: the {head_mode} step vs the plain step (forward + backward over all 448 positions), same weights",
                          "loss_span": float(l_span), "loss_full": float(l_full),
                          "grad_rel_l2": float((net.flat_grads.double() - g_full.double()).norm() / g_full.double().norm()),
                          "active_rows_of_total": round(rows[head_mode][1] / (B * 448.0), 4)}
            del g_full
        if world == 1 and not args.no_cpu_baseline:
            # parity of the benchmarked model itself (outside the timed region): clip 0 of the oracle's generator through the
            # HIP step and through the fp32 CPU oracle with the SAME weights (taken before the hbm-kernel timings touch them)
            from oracle import model_oracle as mo
            net._workspace = None  # the 240 GiB of saved-activation slots are no longer needed: one clip from here on
            torch.cuda.empty_cache()
            sd_cpu = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
            p_pcm, p_ti, p_ty, p_tl = mo.synthetic_batch([0])
            net.zero_grad()
            p_loss, p_logits = net.loss_and_backward(ops.log_mel(p_pcm.to(dev)), p_ti.to(dev), p_ty.to(dev), p_tl.to(dev), return_logits=True)
            torch.cuda.synchronize(dev)
            p_loss, p_logits = float(p_loss), p_logits.cpu()
            out["cpu_baseline"], o_loss, o_logits = cpu_baseline(args.variant, sd_cpu)
            nvalid = int(p_tl[0])
            dl = (p_logits[0, :nvalid] - o_logits[0, :nvalid]).abs()
            out["parity"] = {"what": "clip 0 of the seeded generator, this model's weights after the timed steps: HIP bf16 step vs CPU fp32 oracle",
                             "loss": round(p_loss, 5), "oracle_loss": round(o_loss, 5), "max_dlogit": round(float(dl.max()), 4),
                             "mean_dlogit": round(float(dl.mean()), 5), "logit_scale": round(float(o_logits[0, :nvalid].abs().max()), 2),
                             "argmax_agree": round(float((p_logits[0, :nvalid].argmax(-1) == o_logits[0, :nvalid].argmax(-1)).float().mean()), 4)}
            del p_logits, o_logits
        if span_check:
            out.setdefault("parity", {})["span_step_vs_plain_step"] = span_check
        if world == 1 and not args.no_profile:
            net._workspace = None
            torch.cuda.empty_cache()
            roof_h = hbm_kernel_rooflines(net, dims, mb, dev)
            if roof:
                out["roofline"]["hbm_kernels"] = roof_h
            else:
                out["hbm_kernels"] = roof_h
    # The JSON line must be the LAST line on stdout: RCCL prints its version banner (NCCL_DEBUG=VERSION is set on the GPU
    # boxes) through C stdio, which is block-buffered when stdout is a pipe and would otherwise surface after our line, at
    # process exit.  Every rank flushes C stdio, then the group is torn down, then rank 0 prints.
    if ddp_path:
        dist.barrier()
        _flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
