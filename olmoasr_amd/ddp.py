"""Data-parallel gradient exchange over the flat gradient arena.

Replaces ``DistributedDataParallel(model, device_ids=[local_rank])`` of the reference
(scripts/training/train_timestamps.py:2330) and its reducer:

  * C2  init-time parameter broadcast      -> one ``broadcast`` of the flat fp32 parameter arena
  * C3  per-forward buffer broadcast       -> dropped: the sinusoid buffer is a pure function of the dims
  * C4  bucketed gradient all-reduce       -> buckets are CONTIGUOUS RANGES of the flat gradient arena in the order the
        engine finishes them (``OLMoASR.grad_segments``).  The engine records one HIP event per segment while the
        backward is being enqueued; each bucket's all-reduce waits for its last segment's event on a side stream, so
        RCCL traffic overlaps the remaining backward.  It fires once per accumulation window (the reference fires on
        every micro-batch because it never uses ``no_sync``; sum-of-means == mean-of-sums up to rounding order).
        xGMI is point-to-point (7 links/GPU), so buckets are large (default 128 MiB) to amortise RCCL launch cost and
        let RCCL spread one collective over all links; the SUM is turned into the reference's mean by folding
        1/world_size into the optimizer's unscale factor (no extra pass over the gradients).

Works with any ``torch.distributed`` backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU for the unit tests.
"""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def plan_buckets(segments: Sequence[Tuple[int, int]], cap_elems: int) -> List[Tuple[int, int, int]]:
    """Merge gradient segments (offset, numel), given in completion order, into buckets (offset, numel, last_segment)
    of at most ``cap_elems`` elements (a single larger segment becomes its own bucket).  Only arena-contiguous
    neighbours are merged, so every bucket is one contiguous range."""
    buckets = []
    cur = None
    for i, (off, n) in enumerate(segments):
        if n == 0:
            continue
        if cur is not None and cur[0] + cur[1] == off and cur[1] + n <= cap_elems:
            cur = (cur[0], cur[1] + n, i)
        else:
            if cur is not None:
                buckets.append(cur)
            cur = (off, n, i)
    if cur is not None:
        buckets.append(cur)
    return buckets


class GradReducer:
    """``algo``: "allreduce" (default: one RCCL all-reduce per bucket, RCCL picks rings/trees) or "direct": the all-reduce
    spelled as its two halves, ``reduce_scatter_tensor`` + ``all_gather_into_tensor``, IN PLACE on the arena slice (rank r
    owns chunk r of the bucket: the reduce-scatter's output and the all-gather's input alias that chunk, RCCL's in-place
    form -- no staging buffers, no copies).  On a fully connected xGMI node each half talks to all 7 peers at once
    (~5 ms vs ~35 ms per 3 GB at medium for a one-link ring); it is also the building block of a
    ZeRO-1 step (run the optimizer on the owned chunk between the two halves).  Not the default: no multi-GPU node was
    available to time it against RCCL's own all-reduce; numerically it is a different summation order."""

    def __init__(self, flat_grads: torch.Tensor, segments: Sequence[Tuple[int, int]], bucket_cap_mb: float = 128.0,
                 group: Optional[dist.ProcessGroup] = None, force: bool = False, algo: str = "allreduce", timing: bool = False):
        assert algo in ("allreduce", "direct")
        # timing=True (bench.py): every reduce() brackets the exchange with three timing-enabled HIP events -- backward enqueued to its end
        # on the compute stream, first bucket started / last bucket done on the communication stream -- read back by comm_report()
        self.timing = bool(timing) and flat_grads.is_cuda
        self._tev = None
        # Overlap self-check in training (timing=False): every `overlap_check_every`-th reduce() records the same three events and the NEXT
        # reduce() reads them (they have long completed: no synchronisation).  The exchange overlaps the backward only because the
        # communication stream has a hardware queue of its own (see comm_stream below) -- an empirical property of the runtime; if a driver
        # or torch update ever folds the two streams into one queue again, the first bucket starts AFTER the backward has ended
        # (comm_lead_ms <= 0) and every step silently pays the whole exchange.  Warned about once, loudly.
        self.overlap_check_every = 64
        self._n_reduce = 0
        self._pending_check = None
        self._warned_no_overlap = False
        self.algo = algo
        self.flat = flat_grads
        self.force = force  # run the collectives even at world_size 1 (single-GPU test of the event/stream path)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.segments = list(segments)
        self.buckets = plan_buckets(self.segments, int(bucket_cap_mb * (1 << 20) // flat_grads.element_size()))
        covered = sum(n for _, n, _ in self.buckets)
        assert covered == flat_grads.numel(), "gradient segments must tile the arena"
        self.cuda = flat_grads.is_cuda
        # None: leave the SUM (the fused path folds 1/world into the optimizer's unscale).  A float (DistributedDataParallel sets
        # 1/world): the buckets come back as the MEAN -- ReduceOp.AVG inside the collective on RCCL (no extra pass over the
        # gradients at all), SUM followed by a per-bucket scale on the communication stream elsewhere (gloo has no AVG).
        self.mean_scale: Optional[float] = None
        # HIGH-PRIORITY stream: on ROCm a default-priority stream may share its hardware queue with the compute stream -- its commands
        # then run in submission order behind the whole backward and nothing overlaps (measured at world 1: first bucket started 0.01 ms
        # AFTER the end of the backward, `comm_lead_ms` in bench.py's `ddp` block, profiles/r05_ddp_overlap.txt); a stream of another
        # priority gets a queue of its own, and the short RCCL kernels are scheduled ahead of the GEMMs that fill the chip
        self.comm_stream = torch.cuda.Stream(flat_grads.device, priority=-1) if self.cuda else None
        self.events = None
        if self.cuda:
            # torch creates the underlying HIP event lazily at the first record(); the engine needs the raw handles
            # (Event.cuda_event), so materialise them now -- a null handle would silently turn the per-bucket waits into no-ops.
            self.events = [torch.cuda.Event() for _ in self.segments]
            with torch.cuda.device(flat_grads.device):
                for e in self.events:
                    e.record()
            assert all(e.cuda_event for e in self.events), "HIP event handles were not created"

    @property
    def grad_divisor(self) -> float:
        """Gradients hold the SUM over ranks after ``reduce``; divide by this (fold it into the unscale factor)."""
        return float(self.world)

    def segment_events(self):
        """Pass to ``OLMoASR.loss_and_backward(segment_events=...)`` on the LAST micro-batch of a window."""
        return self.events

    def _sum_bucket(self, t: torch.Tensor):
        """SUM (or, with ``mean_scale``, MEAN) over ranks of the 1-D fp32 view ``t``, in place."""
        avg = self.mean_scale is not None and self.world > 1
        in_op = avg and dist.get_backend(self.group) == "nccl"  # RCCL: ncclAvg
        op = dist.ReduceOp.AVG if in_op else dist.ReduceOp.SUM
        if self.algo == "allreduce" or self.world == 1:
            dist.all_reduce(t, op=op, group=self.group)
        else:
            W, n = self.world, t.numel()
            per = n // W
            rank = dist.get_rank(self.group)
            if per:
                body = t[: per * W]
                mine = body[rank * per:(rank + 1) * per]
                dist.reduce_scatter_tensor(mine, body, op=op, group=self.group)       # rank r ends up owning chunk r's sum
                dist.all_gather_into_tensor(body, mine, group=self.group)             # every rank fetches every chunk
            if per * W < n:  # fewer than W trailing elements
                dist.all_reduce(t[per * W:], op=op, group=self.group)
        if avg and not in_op:
            t.mul_(self.mean_scale)

    def reduce(self, use_events: bool = True):
        """All-reduce (SUM) every bucket.  With events: bucket k starts as soon as its gradients are final."""
        if self.world == 1 and not self.force:
            return
        if not self.cuda:
            for off, n, _ in self.buckets:
                self._sum_bucket(self.flat[off:off + n])
            return
        main = torch.cuda.current_stream(self.flat.device)
        if not use_events:
            self.comm_stream.wait_stream(main)
        self._n_reduce += 1
        self._check_overlap_sample()
        sample = (not self.timing and use_events and len(self.buckets) > 1 and self.overlap_check_every > 0
                  and self._n_reduce >= 2 and (self._n_reduce - 2) % self.overlap_check_every == 0)  # (never the first step: warm-up)
        tev = None
        if self.timing or sample:
            tev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tev[0].record(main)  # every kernel of the backward has been enqueued in front of this
        with torch.cuda.stream(self.comm_stream):
            for k, (off, n, last) in enumerate(self.buckets):
                if use_events:
                    self.comm_stream.wait_event(self.events[last])
                if tev is not None and k == 0:
                    tev[1].record(self.comm_stream)
                self._sum_bucket(self.flat[off:off + n])
            if tev is not None:
                tev[2].record(self.comm_stream)
        if sample:
            self._pending_check = tev
        else:
            self._tev = tev
        main.wait_stream(self.comm_stream)

    def _check_overlap_sample(self):
        tev, self._pending_check = self._pending_check, None
        if tev is None or self._warned_no_overlap or not tev[2].query():
            return
        lead = tev[1].elapsed_time(tev[0])  # first bucket started -> end of the backward
        if lead <= 0.0:
            self._warned_no_overlap = True
            import warnings
            warnings.warn(f"GradReducer: the gradient exchange did not overlap the backward (first bucket started {-lead:.2f} ms AFTER the "
                          f"backward ended; exchange {tev[1].elapsed_time(tev[2]):.2f} ms fully exposed per step).  The communication stream "
                          "no longer has a hardware queue of its own -- see GradReducer.comm_stream / ddp.rccl_options()")

    def comm_report(self) -> dict:
        """Of the most recent ``reduce()`` (``timing=True``; synchronises on its last event):
          exposed_comm_ms -- end of the backward on the compute stream -> last bucket done: the part of the exchange the optimizer step
                             had to WAIT for, i.e. what overlap did not hide (0 when the last bucket finished before the backward did)
          comm_span_ms    -- first bucket started -> last bucket done: how long the exchange was in flight (overlapped or not)
          comm_lead_ms    -- first bucket started -> end of the backward: how much backward the exchange had to hide under"""
        if not self._tev:
            return {"exposed_comm_ms": None, "comm_span_ms": None, "comm_lead_ms": None}
        e_bwd, e_first, e_done = self._tev
        e_done.synchronize()
        e_bwd.synchronize()
        return {"exposed_comm_ms": round(max(0.0, e_bwd.elapsed_time(e_done)), 3), "comm_span_ms": round(e_first.elapsed_time(e_done), 3),
                "comm_lead_ms": round(e_first.elapsed_time(e_bwd), 3)}


class DistributedDataParallel(torch.nn.Module):
    """``torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])`` (train_timestamps.py:2330) for the native model,
    on top of its autograd bridge: constructor = parameter broadcast from rank 0 (``_sync_module_states``), ``forward`` = the wrapped
    module's, and after every backward the gradient arena is all-reduced to the MEAN over ranks (bucketed in completion order on a
    side stream, each bucket waiting only for its own segments' events, i.e. overlapped with the rest of the backward) -- so
    ``p.grad`` holds what torch's DDP would leave there, ``state_dict()`` keys carry the ``module.`` prefix (:935), and the reference's
    loop runs unchanged.  Like torch's class it reduces on EVERY backward unless inside ``no_sync()``; under gradient accumulation
    without ``no_sync`` the already-averaged part is averaged again (a no-op: it is identical on all ranks), as in torch.
    The fused path (``loss_and_backward`` + ``GradReducer`` + ``optim_step``) does one exchange per window and no extra pass.
    Bucket size: the GradReducer default (128 MiB: few, large collectives for 7 x 153 GB/s point-to-point xGMI links), not torch's
    25 MiB, which is tuned for NVSwitch-class all-reduce latency."""

    def __init__(self, module, device_ids=None, output_device=None, bucket_cap_mb: float = None, process_group=None, algo: str = "allreduce", **_ignored):
        super().__init__()
        self.module = module
        broadcast_parameters(module.flat_params, group=process_group)
        module.refresh_shadow()
        self.reducer = GradReducer(module.flat_grads, module.grad_segments, group=process_group, algo=algo, force=dist.is_initialized(),
                                   **({} if bucket_cap_mb is None else {"bucket_cap_mb": bucket_cap_mb}))
        self.require_backward_grad_sync = True
        self._sync_this_backward = False  # decided at forward time, as torch's DDP does: a forward under no_sync() records no events
        module._autograd_post_backward = self._after_backward
        # SUM -> mean without a pass of its own: the 1/world rides in the reducer (``GradReducer.mean_scale``: applied per bucket on
        # the communication stream, behind that bucket's collective and under the rest of the backward)
        self.reducer.mean_scale = 1.0 / self.reducer.world if self.reducer.world > 1 else None

    def _after_backward(self):
        if not self._sync_this_backward:
            return
        self.reducer.reduce()

    def forward(self, *args, **kwargs):
        self._sync_this_backward = self.require_backward_grad_sync
        self.module._autograd_segment_events = self.reducer.segment_events() if self._sync_this_backward else None
        return self.module(*args, **kwargs)

    def no_sync(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self.require_backward_grad_sync = self.require_backward_grad_sync, False
            try:
                yield
            finally:
                self.require_backward_grad_sync = old
        return ctx()


def rccl_options():
    """``pg_options`` for ``init_process_group("nccl", ...)``: RCCL's kernels go to a HIGH-PRIORITY stream (a hardware queue of their own,
    scheduled ahead of the GEMMs that fill the chip) -- same reason as ``GradReducer.comm_stream``.  None where the build has no NCCL backend."""
    try:
        return dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:  # pragma: no cover
        return None


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """DDP constructor's ``_sync_module_states`` (C2) on the flat arena."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def shard_indices(n_samples: int, rank: int, world: int) -> List[int]:
    """DistributedSampler(shuffle=False, drop_last=False) partition (train_timestamps.py:633-638): pad by wrapping to
    a multiple of world, then indices[rank::world]."""
    idx = list(range(n_samples))
    total = (n_samples + world - 1) // world * world
    idx += idx[: total - n_samples]
    return idx[rank:total:world]
