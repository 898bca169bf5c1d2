"""world_size-2 gloo tests (CPU) of the data-parallel path: bucket planning over the flat gradient arena, the
reducer's SUM + folded 1/world mean, parameter broadcast and the DistributedSampler partition rule."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from olmoasr_amd import ddp


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
    port = 29500 + os.getpid() % 2000
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
