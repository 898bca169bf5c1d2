"""ZeRO-1: optimizer-state sharding over the flat arenas.

The reference's memory-saving variant is FSDP (scripts/training/train_fsdp_timestamps.py:2665-2719: parameters, gradients and
AdamW state sharded per ``ResidualAttentionBlock``).  With one flat fp32 arena per quantity, the MI355X-native counterpart is
the classic ZeRO-1 schedule on contiguous ranges -- rank ``r`` owns ``[r * per, (r + 1) * per)`` (the last rank also the
< world-size tail):

    backward (every rank, full gradients)
    reduce_scatter_tensor(grads[own], grads)          in place on the arena (xGMI: all 7 links at once)
    own partial sum of squares  ->  all_reduce(2 floats)   = the global norm / found_inf that clip_grad_norm_ and GradScaler need
    fused unscale + clip + AdamW on the OWN range      exp_avg / exp_avg_sq exist for that range only: 8 B/param * (W-1)/W saved
    all_gather_into_tensor(params, params[own])        in place;  then the bf16 compute copies are refreshed

Per step it moves the same bytes as the all-reduce it replaces (reduce-scatter + all-gather ARE its two halves).  Numerically the
result equals the replicated step bit for bit given the same reduced gradients (each element's update depends only on its own
g, m, v, p and on the two global scalars).  ``backend`` abstracts the two range kernels so the collective schedule is testable
on CPU with gloo; the product backend is liboasr (``oasr_grad_sumsq_range`` / ``oasr_optim_step_range``).
"""
from typing import Optional

import torch
import torch.distributed as dist

from . import _native as N


def shard_range(numel: int, rank: int, world: int, align: int = 4):
    """(offset, length) of rank's range: ``per = numel // world`` rounded down to ``align``; the last rank takes the rest."""
    per = (numel // world) // align * align
    off = rank * per
    return off, (numel - off) if rank == world - 1 else per


class NativeBackend:
    """The two range kernels of liboasr on the model's arenas."""

    def __init__(self, net):
        self.net = net
        self.stats = torch.zeros(2, device=net.flat_params.device, dtype=torch.float32)
        self.scratch = torch.zeros(8192, device=net.flat_params.device, dtype=torch.uint8)

    def alloc(self, n):
        return torch.zeros(n, device=self.net.flat_params.device, dtype=torch.float32)

    def sumsq(self, off, n):
        with torch.cuda.device(self.stats.device):
            N.check(N.lib().oasr_grad_sumsq_range(self.net._ctx, off, n, N.ptr(self.stats), N.ptr(self.scratch), N.stream_ptr()), "sumsq_range")
        return self.stats

    def step(self, off, n, m, v, stats, **h):
        with torch.cuda.device(self.stats.device):
            N.check(N.lib().oasr_optim_step_range(self.net._ctx, off, n, N.ptr(m), N.ptr(v), N.ptr(stats), float(h["inv_loss_scale"]),
                                                  float(h["max_grad_norm"]), float(h["lr"]), float(h["betas"][0]), float(h["betas"][1]),
                                                  float(h["eps"]), float(h["weight_decay"]), int(h["step"]), N.stream_ptr()), "optim_step_range")

    def after_gather(self):
        self.net.refresh_shadow()


class ShardedOptimizer:
    def __init__(self, flat_params: torch.Tensor, flat_grads: torch.Tensor, backend, group: Optional[dist.ProcessGroup] = None,
                 force: bool = False):
        self.p, self.g, self.backend, self.group = flat_params, flat_grads, backend, group
        self.force = force and dist.is_initialized()  # run the collectives even at world_size 1 (single-GPU rehearsal over RCCL)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        n = flat_params.numel()
        self.per = (n // self.world) // 4 * 4
        self.off, self.len = shard_range(n, self.rank, self.world)
        self.m, self.v = backend.alloc(self.len), backend.alloc(self.len)  # optimizer state of the owned range only

    @property
    def grad_divisor(self) -> float:
        return float(self.world)

    def state_bytes_saved(self) -> int:
        return 8 * (self.p.numel() - self.len)

    def step(self, *, step: int, lr: float, inv_loss_scale: float = 1.0, max_grad_norm: float = 1.0, betas=(0.9, 0.98), eps: float = 1e-6,
             weight_decay: float = 0.1):
        """``inv_loss_scale`` must already contain 1 / world (the reduced gradients are SUMs, like GradReducer's).  Returns the
        global stats tensor [sum g^2 (scaled), found_inf]."""
        W, per, n = self.world, self.per, self.p.numel()
        comm = W > 1 or self.force
        if comm:
            body = self.g[: per * W]
            dist.reduce_scatter_tensor(body[self.rank * per:(self.rank + 1) * per], body, op=dist.ReduceOp.SUM, group=self.group)
            if per * W < n:  # the tail belongs to the last rank
                dist.reduce(self.g[per * W:], dst=dist.get_global_rank(self.group, W - 1) if self.group else W - 1, op=dist.ReduceOp.SUM,
                            group=self.group)
        stats = self.backend.sumsq(self.off, self.len)
        if comm:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        self.backend.step(self.off, self.len, self.m, self.v, stats, step=step, lr=lr, inv_loss_scale=inv_loss_scale,
                          max_grad_norm=max_grad_norm, betas=betas, eps=eps, weight_decay=weight_decay)
        if comm:
            body = self.p[: per * W]
            dist.all_gather_into_tensor(body, body[self.rank * per:(self.rank + 1) * per], group=self.group)
            if per * W < n:
                dist.broadcast(self.p[per * W:], src=dist.get_global_rank(self.group, W - 1) if self.group else W - 1, group=self.group)
        self.backend.after_gather()
        return stats

    def gather_state(self):
        """Full-length (exp_avg, exp_avg_sq) on every rank -- for checkpoints in torch.optim.AdamW's layout
        (OLMoASR.optimizer_state_dict(moments=...)); transient, 8 B/param."""
        n, W, per = self.p.numel(), self.world, self.per
        out = []
        for t in (self.m, self.v):
            full = torch.zeros(n, device=t.device, dtype=t.dtype)
            full[self.off:self.off + self.len] = t
            if W > 1:
                dist.all_gather_into_tensor(full[: per * W], full[self.rank * per:(self.rank + 1) * per].clone(), group=self.group)
                if per * W < n:
                    dist.broadcast(full[per * W:], src=dist.get_global_rank(self.group, W - 1) if self.group else W - 1, group=self.group)
            out.append(full)
        return tuple(out)

    def load_state(self, m_full: torch.Tensor, v_full: torch.Tensor):
        self.m.copy_(m_full[self.off:self.off + self.len])
        self.v.copy_(v_full[self.off:self.off + self.len])
