"""world_size-2 gloo tests (CPU) of the data-parallel path: bucket planning over the flat gradient arena, the
reducer's SUM + folded 1/world mean, parameter broadcast and the DistributedSampler partition rule."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from olmoasr_amd import ddp


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_plan_buckets_tiles_arena_and_respects_completion_order():
    # decoder segments, then the embedding (arena tail) completes BEFORE the encoder segments that precede it in memory
    segs = [(0, 10), (10, 30), (40, 30), (100, 50), (70, 20), (90, 10)]
    b = ddp.plan_buckets(segs, cap_elems=45)
    assert sum(n for _, n, _ in b) == 150
    assert b == [(0, 40, 1), (40, 30, 2), (100, 50, 3), (70, 30, 5)]
    assert ddp.plan_buckets(segs, cap_elems=10 ** 9) == [(0, 70, 2), (100, 50, 3), (70, 30, 5)]
    assert ddp.plan_buckets([(0, 0), (0, 5)], 4) == [(0, 5, 1)]


def test_shard_indices_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    data = list(range(11))
    for world in (1, 2, 4, 8):
        for rank in range(world):
            s = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=False, drop_last=False)
            assert list(iter(s)) == ddp.shard_indices(len(data), rank, world)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        segs = [(0, 100), (100, 300), (400, 200), (800, 200), (600, 200)]
        params = torch.full((n,), float(rank + 1))
        ddp.broadcast_parameters(params, src=0)
        assert torch.equal(params, torch.ones(n))
        g = torch.Generator().manual_seed(rank)
        local = torch.randn(n, generator=g)
        flat = local.clone()
        red = ddp.GradReducer(flat, segs, bucket_cap_mb=300 * 4 / (1 << 20))
        assert [x[:2] for x in red.buckets] == [(0, 100), (100, 300), (400, 200), (800, 200), (600, 200)]
        red.reduce()
        mean = flat / red.grad_divisor
        want = sum(torch.randn(n, generator=torch.Generator().manual_seed(r)) for r in range(world)) / world
        assert torch.allclose(mean, want, atol=1e-6)
        # the all-links variant (reduce-scatter and all-gather as all-to-alls), ragged bucket sizes included
        flat2 = local.clone()
        red2 = ddp.GradReducer(flat2, [(0, 333), (333, 667)], bucket_cap_mb=400 * 4 / (1 << 20), algo="direct")
        red2.reduce()
        assert torch.allclose(flat2 / red2.grad_divisor, want, atol=1e-6)
        # weak-scaling data partition: every sample index is owned by exactly one rank
        mine = ddp.shard_indices(16, rank, world)
        allidx = [None] * world
        dist.all_gather_object(allidx, mine)
        assert sorted(sum(allidx, [])) == list(range(16))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_synth_loader_order_and_determinism():
    """The background loader (the reference's DataLoader role) yields the sampler's batches in order, bit-identical to
    direct generation, whatever the worker count / prefetch depth."""
    from olmoasr_amd.synth import SynthLoader, synth_samples
    order = [[0, 1], [2, 3], [4, 5], [1, 0]]
    dev = torch.device("cpu")
    for workers, depth in ((1, 1), (4, 3)):
        got = list(SynthLoader(order, dev, workers=workers, depth=depth))
        assert len(got) == len(order)
        for idx, batch in zip(order, got):
            ref = synth_samples(idx, dev)
            assert all(torch.equal(a, b) for a, b in zip(batch, ref))
            assert batch[0].dtype == torch.int16 and batch[0].shape == (2, 480000) and batch[3].dtype == torch.int32


# ---- the whole data-parallel optimizer step, world 2 vs world 1, gradients from the CPU oracle --------------------------
def _oracle_flat_grads(sd, dims, idx, loss_scale, accum, names):
    """Gradients of (loss / accum) * loss_scale for the samples ``idx`` as one flat fp32 vector (arena stand-in)."""
    import numpy as np
    from oracle import mel_oracle as me
    from oracle import model_oracle as mo
    pcm, ti, ty, tl = mo.synthetic_batch(idx)
    mel = torch.from_numpy(me.log_mel_batch(pcm.numpy(), dtype=np.float32))
    loss, grads, _ = mo.loss_and_grads(sd, dims, mel, ti, ty, tl, loss_scale=loss_scale, accumulation_steps=accum)
    return float(loss), torch.cat([grads[n].reshape(-1) for n in names])


def _step_from_flat(sd, names, flat, inv_scale, lr=1e-3):
    """unscale -> clip_grad_norm_(1.0) -> AdamW step 1 (train_timestamps.py:1509-1512) from a flat gradient vector."""
    from oracle import model_oracle as mo
    grads, off = {}, 0
    for n in names:
        k = sd[n].numel()
        grads[n] = (flat[off:off + k] * inv_scale).view_as(sd[n]).clone()
        off += k
    total, coef = mo.clip_coef(grads, 1.0)
    for n in names:
        grads[n].mul_(coef)
    params = {n: sd[n].clone() for n in names}
    m = {n: torch.zeros_like(params[n]) for n in names}
    v = {n: torch.zeros_like(params[n]) for n in names}
    mo.adamw_step(params, grads, m, v, step=1, lr=lr)
    return float(total), params


def _dp_worker(rank, world, port, q, algo):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(4)
        from oracle import model_oracle as mo
        dims = mo.Dims(80, 1500, 64, 1, 1, 51864, 448, 64, 1, 1)
        sd = mo.init_state_dict(dims, seed=0)
        names = [k for k in sd if k != "encoder.positional_embedding"]
        scale = 1024.0
        # global batch of 4 samples: rank r holds indices[r::world] (DistributedSampler), one micro-batch of 2 per rank
        mine = ddp.shard_indices(4, rank, world)
        assert mine == [rank, rank + 2]
        loss, flat = _oracle_flat_grads(sd, dims, [70 + i for i in mine], scale, 1, names)
        n = flat.numel()
        segs = [(0, n // 3), (n // 3, n // 2 - n // 3), (n // 2, n - n // 2)]
        red = ddp.GradReducer(flat, segs, bucket_cap_mb=1.0, algo=algo)
        assert len(red.buckets) == 3
        red.reduce()                                              # SUM over ranks ...
        inv = 1.0 / (scale * red.grad_divisor)                   # ... turned into DDP's mean inside the unscale factor
        norm, params = _step_from_flat(sd, names, flat, inv)
        lt = torch.tensor([loss])
        dist.all_reduce(lt)
        # (numpy, pickled by value: torch tensors travel as file descriptors the parent can only fetch while this process lives)
        q.put((rank, "ok", norm, float(lt) / world, {k: v.numpy().copy() for k, v in list(params.items())[:3]}, float(sum(p.double().sum() for p in params.values()))))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "err: " + traceback.format_exc(), 0, 0, None, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["allreduce", "direct"])
def test_data_parallel_step_world2_equals_world1(algo):
    """one optimizer step on a global batch of 4 clips computed (a) by two ranks holding two clips
    each -- sampler shard, per-rank backward of the loss-scaled loss, bucketed SUM of the flat gradient arena, 1/(scale *
    world) folded into the unscale factor, clip, AdamW -- and (b) by one rank accumulating the same two micro-batches
    (accumulation_steps = 2).  DDP's mean over ranks == the accumulation rule's 1/accum (train_timestamps.py:1450), so norms,
    losses and updated weights must agree to summation-order noise."""
    from oracle import model_oracle as mo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, algo)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
    # world 1, two accumulated micro-batches with the SAME per-rank sample split
    torch.set_num_threads(8)
    dims = mo.Dims(80, 1500, 64, 1, 1, 51864, 448, 64, 1, 1)
    sd = mo.init_state_dict(dims, seed=0)
    names = [k for k in sd if k != "encoder.positional_embedding"]
    scale = 1024.0
    l0, f0 = _oracle_flat_grads(sd, dims, [70, 72], scale, 2, names)
    l1, f1 = _oracle_flat_grads(sd, dims, [71, 73], scale, 2, names)
    norm1, params1 = _step_from_flat(sd, names, f0 + f1, 1.0 / scale)
    for rank, _, norm, loss, head, checksum in res:
        assert abs(norm - norm1) < 1e-5 * norm1, (norm, norm1)
        assert abs(loss - (l0 + l1)) < 1e-5
        for k, v in head.items():
            assert torch.allclose(torch.from_numpy(v), params1[k], atol=2e-6), k
        assert abs(checksum - float(sum(p.double().sum() for p in params1.values()))) < 1e-2
    assert res[0][2] == res[1][2]  # both ranks hold bit-identical results after the exchange


# ---- ZeRO-1 schedule (olmoasr_amd/zero.py) under gloo, with a torch stand-in for the two range kernels -------------------
class _TorchRangeBackend:
    """CPU stand-in with the semantics of oasr_grad_sumsq_range / oasr_optim_step_range (fused unscale + clip + AdamW)."""

    def __init__(self, p, g):
        self.p, self.g = p, g

    def alloc(self, n):
        return torch.zeros(n)

    def sumsq(self, off, n):
        x = self.g[off:off + n]
        return torch.tensor([float((x.double() ** 2).sum()), float((~torch.isfinite(x)).any())])

    def step(self, off, n, m, v, stats, *, step, lr, inv_loss_scale, max_grad_norm, betas, eps, weight_decay):
        if float(stats[1]) != 0:
            return
        total = float(stats[0]) ** 0.5 * inv_loss_scale
        coef = min(1.0, max_grad_norm / (total + 1e-6))
        g = self.g[off:off + n] * (coef * inv_loss_scale)
        p = self.p[off:off + n]
        p.mul_(1 - lr * weight_decay)
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = (v.sqrt() / (1 - betas[1] ** step) ** 0.5).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - betas[0] ** step))

    def after_gather(self):
        pass


def _zero_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from olmoasr_amd import zero
        n = 1003 * 4 + 4  # not a multiple of world * 4: exercises the tail
        p = torch.randn(n, generator=torch.Generator().manual_seed(0))
        g = torch.randn(n, generator=torch.Generator().manual_seed(10 + rank)) * 1024.0  # this rank's loss-scaled gradients
        opt = zero.ShardedOptimizer(p, g, _TorchRangeBackend(p, g))
        assert opt.len == zero.shard_range(n, rank, world)[1] and opt.m.numel() == opt.len < n
        outs = []
        for step in (1, 2):
            if step == 2:
                g.copy_(torch.randn(n, generator=torch.Generator().manual_seed(20 + rank)) * 1024.0)
            stats = opt.step(step=step, lr=1e-2, inv_loss_scale=1.0 / (1024.0 * opt.grad_divisor))
            outs.append((p.numpy().copy(), float(stats[0])))  # numpy: pickled by value
        q.put((rank, "ok", outs, opt.off, opt.len))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None, 0, 0))
    finally:
        dist.destroy_process_group()


def test_zero1_sharded_step_equals_replicated_step():
    """Two ranks, each keeping AdamW moments for half of the arena only: after reduce-scatter / partial norms / range step /
    all-gather both hold the parameters a single replicated optimizer produces from the mean gradient -- for two consecutive
    steps (the sharded moments carry over)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for pr in procs:
        pr.join(30)
    assert all(r[1] == "ok" for r in res), res
    n = 1003 * 4 + 4
    p = torch.randn(n, generator=torch.Generator().manual_seed(0))
    ref = _TorchRangeBackend(p, torch.zeros(n))
    m, v = torch.zeros(n), torch.zeros(n)
    for step, seeds in ((1, (10, 11)), (2, (20, 21))):
        ref.g = sum(torch.randn(n, generator=torch.Generator().manual_seed(s)) * 1024.0 for s in seeds)
        ref.step(0, n, m, v, ref.sumsq(0, n), step=step, lr=1e-2, inv_loss_scale=1.0 / 2048.0, max_grad_norm=1.0, betas=(0.9, 0.98), eps=1e-6,
                 weight_decay=0.1)
        for rank, _, outs, off, ln in res:
            got, sumsq = torch.from_numpy(outs[step - 1][0]), outs[step - 1][1]
            assert torch.allclose(got, p, atol=1e-6, rtol=0), (step, rank, float((got - p).abs().max()))
            assert abs(sumsq - float(ref.sumsq(0, n)[0])) < 1e-3 * sumsq
    assert res[0][3] == 0 and res[1][3] == res[0][4] and res[1][3] + res[1][4] == n  # the two ranges tile the arena


# ---- ddp.DistributedDataParallel (the reference's DDP(model, device_ids=[...]) call) under gloo, world 2, on a stand-in module -----------
class _ArenaModule(torch.nn.Module):
    """What the wrapper needs of OLMoASR: flat parameter / gradient arenas, their segments, refresh_shadow, and an autograd-free backward that
    writes the arena and then fires ``_autograd_post_backward`` (model.py::_TrainStep.backward does exactly that on the GPU)."""

    def __init__(self, n, rank):
        super().__init__()
        self.w = torch.nn.Parameter(torch.full((n,), float(rank + 1)))
        self.flat_params = self.w.data
        self.flat_grads = torch.zeros(n)
        self.grad_segments = [(0, n // 4), (n // 4, n - n // 4)]
        self.refreshed = 0

    def refresh_shadow(self):
        self.refreshed += 1

    def forward(self, x):
        return x

    def fake_backward(self, g):
        self.flat_grads += g
        post = getattr(self, "_autograd_post_backward", None)
        if post is not None:
            post()


def _wrapper_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1001
        base = _ArenaModule(n, rank)
        model = ddp.DistributedDataParallel(base, device_ids=[rank], bucket_cap_mb=1000 * 4 / (1 << 20))
        assert torch.equal(base.flat_params, torch.ones(n)) and base.refreshed == 1          # rank 0's parameters everywhere
        assert list(model.state_dict()) == ["module.w"]
        g = [torch.randn(n, generator=torch.Generator().manual_seed(10 * r + k)) for r in range(world) for k in range(2)]
        mine = [g[2 * rank], g[2 * rank + 1]]
        mean = [sum(g[2 * r + k] for r in range(world)) / world for k in range(2)]
        # two micro-batches without no_sync: reduced after each backward; the already-averaged part stays put
        model(torch.zeros(1))
        base.fake_backward(mine[0])
        assert torch.allclose(base.flat_grads, mean[0], atol=1e-6)
        model(torch.zeros(1))
        base.fake_backward(mine[1])
        assert torch.allclose(base.flat_grads, mean[0] + mean[1], atol=1e-6)
        # the same window with no_sync on the first micro-batch: one exchange, same result
        base.flat_grads.zero_()
        with model.no_sync():
            model(torch.zeros(1))
            base.fake_backward(mine[0])
            assert torch.equal(base.flat_grads, mine[0])
        model(torch.zeros(1))
        base.fake_backward(mine[1])
        assert torch.allclose(base.flat_grads, mean[0] + mean[1], atol=1e-6)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_ddp_wrapper_world2_mean_allreduce_and_no_sync():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wrapper_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


def test_bench_self_launches_ranks_without_torchrun():
    """`python bench.py --gpus N` is the shape of the driver's N = 1 command: for N > 1 outside a launcher bench.py starts the ranks
    itself (torch.distributed.run, loopback rendezvous) and rank 0 prints the single JSON line.  Covered here on CPU with the stub step
    (OASR_BENCH_STUB=1: gloo all-reduce, same barrier / max-over-ranks / rank-0-prints plumbing); without devices the real path must
    fail with a plain message, not an assert."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OASR_BENCH_STUB="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-profile",
                        "--no-cpu-baseline", "--reducer", "both"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["buf_ok"] is True
    # a multi-rank line diagnoses itself: per-step spread, the `ddp` block (exchange shape, slowest / fastest rank, exposed communication --
    # null on gloo, HIP events on RCCL) and, with --reducer both, the two exchange spellings timed back to back in the one launch
    assert set(out["per_step_ms"]) == {"min", "median", "max", "n"} and out["per_step_ms"]["n"] == 3
    d = out["ddp"]
    for k in ("reducer", "bucket_mb", "buckets", "rank_step_ms_min", "rank_step_ms_max", "exposed_comm_ms", "comm_span_ms", "comm_lead_ms", "reducer_ab"):
        assert k in d, k
    assert d["reducer"] == "allreduce" and d["buckets"] >= 1 and d["rank_step_ms_min"] <= d["rank_step_ms_max"]
    assert set(d["reducer_ab"]) == {"allreduce", "direct"} and d["reducer_ab"]["direct"]["ms_per_step"] > 0
    # the real path on a machine with fewer devices than ranks: a message and a non-zero exit code
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("OASR_BENCH_STUB")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode != 0 and "needs 2 devices" in r.stderr and "AssertionError" not in r.stderr, r.stderr[-2000:]


# ---- world 4 and 8 under gloo (round 6: the node the scaling bench runs on has 8 ranks; no such node is available to the builder) -----------
@pytest.mark.parametrize("world", [4, 8])
def test_grad_reducer_world4_and_8_gloo(world):
    """The world-2 worker at the world sizes of the scaling bench: allreduce buckets in completion order, algo="direct" (reduce-scatter +
    all-gather in place on the arena slice) on buckets of 333 and 667 elements -- neither divisible by 4 or 8, so every rank's slice has a
    ragged tail --, parameter broadcast and the DistributedSampler partition."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_shard_range_tiles_the_arena_for_every_world_size():
    from olmoasr_amd import zero
    for n in (4, 8, 1003 * 4 + 4, 762_321_920 // 4 * 4, 37 * 4):  # (4 * k: the arena is padded to whole float4s)
        for world in (1, 2, 3, 4, 8):
            rs = [zero.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(world - 1)) and rs[-1][0] + rs[-1][1] == n
            assert all(off % 4 == 0 and ln % 4 == 0 for off, ln in rs) and all(ln >= 0 for _, ln in rs)
            assert max(ln for _, ln in rs[:-1] or [(0, 0)]) <= rs[-1][1] or world == 1  # the last rank takes the rest


@pytest.mark.parametrize("world", [4, 8])
def test_zero1_sharded_step_world4_and_8(world):
    """ZeRO-1 at the scaling bench's world sizes: n = 4016 elements is not a multiple of world * 4 (the last rank's range is longer), two
    consecutive steps (the sharded moments carry over), every rank ends with the parameters of ONE replicated AdamW on the mean gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for pr in procs:
        pr.join(60)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res if r[1] != "ok"][:1]
    n = 1003 * 4 + 4
    p = torch.randn(n, generator=torch.Generator().manual_seed(0))
    ref = _TorchRangeBackend(p, torch.zeros(n))
    m, v = torch.zeros(n), torch.zeros(n)
    for step, base in ((1, 10), (2, 20)):
        ref.g = sum(torch.randn(n, generator=torch.Generator().manual_seed(base + r)) * 1024.0 for r in range(world))
        ref.step(0, n, m, v, ref.sumsq(0, n), step=step, lr=1e-2, inv_loss_scale=1.0 / (1024.0 * world), max_grad_norm=1.0, betas=(0.9, 0.98), eps=1e-6,
                 weight_decay=0.1)
        for rank, _, outs, off, ln in res:
            got = torch.from_numpy(outs[step - 1][0])
            assert torch.allclose(got, p, atol=2e-6, rtol=0), (step, rank, float((got - p).abs().max()))
    offs = [(r[3], r[4]) for r in res]
    assert offs[0][0] == 0 and all(offs[i][0] + offs[i][1] == offs[i + 1][0] for i in range(world - 1)) and offs[-1][0] + offs[-1][1] == n
