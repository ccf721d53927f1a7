"""
Multi-GPU execution of the RIME path: one process per GPU, baselines sharded across ranks,
`torch.distributed` collectives (backend 'nccl' == RCCL over xGMI on ROCm; 'gloo' in CPU tests).

The reference has no collectives: its `DistributedLogProb` (optim.py:1391-1566) copies
parameters to each device, runs the per-device closures one after another in a Python loop and
sums gradients on device 0.  Here:
  * visibilities are independent across baselines AND across channels.  Either axis can be
    sharded in contiguous blocks (order preserved: the gathered tensor equals the single-GPU
    layout) from replicated sky / beam parameters -- no data-path collective inside the kernels.
    Baseline blocks suit the baseline-formulation kernels; CHANNEL blocks suit the antenna-factored
    matrix-core kernels, whose cost does not depend on how many of the antenna pairs are requested
    (and channel blocks also shard the per-channel sky / beam preparation and their gradients);
  * forward: optional all-gather of the (Npol, Npol, Nbl/W, Nt, Nf) visibility blocks
    (differentiable: its backward hands each rank the slice of the upstream gradient that
    belongs to its baselines -- no communication);
  * backward: one all-reduce (sum) of the parameter gradients, bucketed into a single flat
    buffer per dtype so the ring runs over few, large messages.
"""
import numpy as np
import os

import torch
import torch.distributed as dist


# RIME_DIST_SIMPLE=1: always use the list-based all_gather + copies (debugging aid for a new fabric / backend)
_SIMPLE = os.environ.get('RIME_DIST_SIMPLE', '0') == '1'


def shard_bounds(n, world_size):
    """balanced contiguous partition of range(n): list of (start, stop) per rank"""
    base, extra = divmod(int(n), int(world_size))
    out, s = [], 0
    for r in range(world_size):
        e = s + base + (1 if r < extra else 0)
        out.append((s, e))
        s = e
    return out


def shard_baselines(bls, rank=None, world_size=None):
    """this rank's contiguous block of a baseline list"""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    s, e = shard_bounds(len(bls), world_size)[rank]
    return list(bls[s:e])


class _AllGatherCat(torch.autograd.Function):
    """all-gather of ragged per-rank blocks concatenated along `dim` (differentiable: the
    backward hands every rank the slice of the upstream gradient that belongs to its block)"""
    @staticmethod
    def forward(ctx, x, counts, dim, group):
        world = len(counts)
        rank = dist.get_rank(group)
        nmax = max(counts)
        ctx.counts, ctx.rank, ctx.dim = counts, rank, dim
        if min(counts) == nmax and x.shape[dim] == nmax and not _SIMPLE:
            # equal blocks: one collective into a [world, ...] buffer, then a single re-layout (none at all
            # when every axis before `dim` has length 1, i.e. the baseline-sharded visibility tensor)
            xin = x.contiguous()
            buf = torch.view_as_real(xin) if xin.is_complex() else xin
            out = torch.empty((world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
            dist.all_gather_into_tensor(out.view(-1), buf.reshape(-1), group=group)     # flat: backends differ on shapes
            if xin.is_complex():
                out = torch.view_as_complex(out)
            shape = list(x.shape)
            shape[dim] = world * nmax
            return out.movedim(0, dim).reshape(shape)
        pad = x
        if x.shape[dim] < nmax:
            padshape = list(x.shape)
            padshape[dim] = nmax - x.shape[dim]
            pad = torch.cat([x, x.new_zeros(padshape)], dim=dim)
        pad = pad.contiguous()
        if pad.is_complex():
            buf = torch.view_as_real(pad)
            parts = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(parts, buf, group=group)
            parts = [torch.view_as_complex(p) for p in parts]
        else:
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad, group=group)
        return torch.cat([p.narrow(dim, 0, c) for p, c in zip(parts, counts)], dim=dim)

    @staticmethod
    def backward(ctx, g):
        s = sum(ctx.counts[:ctx.rank])
        return g.narrow(ctx.dim, s, ctx.counts[ctx.rank]).contiguous(), None, None, None


def all_gather_vis(vis_local, counts=None, group=None, dim=2):
    """
    Gather per-rank visibility blocks into the full (Npol, Npol, Nbl, Nt, Nf) tensor on every
    rank, rank blocks in rank order along `dim` (2 = baseline-sharded, 4 = frequency-sharded).
    `counts`: block sizes per rank (default: exchanged with an all_gather of the local size).
    """
    world = dist.get_world_size(group)
    if counts is None:
        n = torch.tensor([vis_local.shape[dim]], device=vis_local.device)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n, group=group)
        counts = [int(x.item()) for x in ns]
    return _AllGatherCat.apply(vis_local, tuple(counts), dim, group)


def all_gather_block_grads(param, dim, bounds, group=None):
    """
    Gradient exchange for a replicated parameter whose slices along `dim` are each used by exactly
    one rank (frequency sharding of per-channel sky / beam parameters): the local .grad is non-zero
    only inside this rank's (start, stop) block, so an all-gather of the blocks rebuilds the full
    gradient with half the bytes of an all-reduce.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    s, e = bounds[rank]
    if param.grad is None:
        param.grad = torch.zeros_like(param)
    nmax = max(b - a for a, b in bounds)
    g = param.grad
    if (not _SIMPLE and min(b - a for a, b in bounds) == nmax and g.is_contiguous() and not g.is_complex()
            and all(n == 1 for n in g.shape[:dim]) and bounds[0][0] == 0 and bounds[-1][1] == g.shape[dim]):
        # equal blocks that are contiguous runs of the gradient in rank order: in-place all-gather
        flat = g.view(-1)
        n = flat.numel() // world
        dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n], group=group)
        return
    blk = param.grad.narrow(dim, s, e - s)
    if e - s < nmax:
        padshape = list(blk.shape)
        padshape[dim] = nmax - (e - s)
        blk = torch.cat([blk, blk.new_zeros(padshape)], dim=dim)
    blk = blk.contiguous()
    parts = [torch.empty_like(blk) for _ in range(world)]
    dist.all_gather(parts, blk, group=group)
    for (a, b), part in zip(bounds, parts):
        param.grad.narrow(dim, a, b - a).copy_(part.narrow(dim, 0, b - a))


def all_reduce_grads(params, group=None, average=False):
    """
    Sum .grad of the given parameters over ranks, in place.  Gradients are flattened into one
    bucket per dtype (complex viewed as real), reduced with a single all_reduce each and
    scattered back.  Parameters without a .grad on this rank contribute zeros.
    """
    params = [p for p in params if p is not None]
    by_dtype = {}
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        by_dtype.setdefault((p.grad.dtype, p.grad.device), []).append(p)
    world = dist.get_world_size(group)
    for (dt, dev), ps in by_dtype.items():
        views = [torch.view_as_real(p.grad).reshape(-1) if p.grad.is_complex() else p.grad.reshape(-1)
                 for p in ps]
        flat = torch.cat(views)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        o = 0
        for p, v in zip(ps, views):
            n = v.numel()
            v.copy_(flat[o:o + n])
            o += n


def reduce_scalar(x, group=None):
    """sum of a scalar tensor (e.g. a shard-local chi^2) over ranks"""
    y = x.detach().clone()
    dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
    return y


class ShardedRIME:
    """
    Baseline-sharded driver around a RIME factory.  `make_rime(sim_bls)` must build a RIME for the
    given baseline list with this rank's replica of the sky / beam models.  forward() returns
    this rank's VisData block, or the gathered visibilities with gather=True.
    """
    def __init__(self, make_rime, all_bls, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.all_bls = list(all_bls)
        self.bounds = shard_bounds(len(self.all_bls), self.world)
        self.counts = [e - s for s, e in self.bounds]
        s, e = self.bounds[self.rank]
        self.local_bls = self.all_bls[s:e]
        self.rime = make_rime(self.local_bls)

    def forward(self, gather=False, **kw):
        vd = self.rime(**kw)
        if gather:
            vd.data = all_gather_vis(vd.data, self.counts, self.group, dim=2)
            vd._set_bls(self.all_bls)
        return vd

    __call__ = forward

    def sync_grads(self, params=None):
        if params is None:
            params = [p for p in self.rime.parameters()]
        all_reduce_grads(params, self.group)
