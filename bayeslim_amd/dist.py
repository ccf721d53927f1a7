"""
Multi-GPU execution of the RIME path: one process per GPU, visibilities sharded across ranks,
`torch.distributed` collectives (backend 'nccl' == RCCL over xGMI on ROCm; 'gloo' in CPU tests).

The reference has no collectives: its `DistributedLogProb` (optim.py:1391-1566) copies
parameters to each device, runs the per-device closures one after another in a Python loop and
sums gradients on device 0 (optim.py:1539-1566).  Here:
  * visibilities are independent across baselines AND across channels; sky / beam parameters are
    replicated; no collective inside the kernels.  Partitions:
      - contiguous baseline blocks (`shard_bounds`): right for the baseline-formulation kernels;
      - baseline TILE shards (`plan_tile_shards`): the north-star's baseline partition for the
        antenna-factored matrix-core kernels, whose cost is per 32 x 32 antenna-pair tile -- whole
        blocks of the pair matrix (ops._antenna_blocks) are dealt to the ranks, so the MFMA work is
        sharded; every rank regenerates the E operands of the antennas its tiles touch;
      - contiguous channel blocks: shard MFMA work AND operand generation AND the per-channel
        sky / beam preparation (the faster partition for those kernels);
  * forward: all-gather of the visibility blocks, differentiable (its backward hands each rank the
    slice of the upstream gradient that belongs to its block -- no communication), available as an
    ASYNC pair (`all_gather_vis_start` / `.wait()`) so that the gather of one time chunk runs while
    the kernels of the next chunk run (`pipelined_step`);
  * backward: gradients of replicated parameters are summed in place, one collective per large
    gradient (small ones share a flat bucket), started by `GradSync` from autograd's
    post-accumulate hooks so that the exchange of a finished gradient overlaps the rest of the
    backward; per-channel parameters under channel sharding exchange their disjoint blocks with an
    in-place all-gather instead (half the bytes).
"""
import os

import numpy as np
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


# ---------------------------------------------------------------------------------------------
# baseline-tile shards for the antenna-factored matrix-core kernels
# ---------------------------------------------------------------------------------------------
# cost model, in MFMA issue slots per 16 pixels (one v_mfma_f32_32x32x16_f16 = 1): forward + backward
# matrix work of a block plus the operand generation of its antenna rows.  GEN_PER_ROW from the C4
# profile (profiles/r01/pmc_v19.json): 128 rows cost 0.46 / 0.56 x the 100 forward MFMAs of the block.
GEN_PER_ROW = 0.64


def _cap(n):
    return 32 if n <= 32 else (64 if n <= 64 else 128)


def _item_cost(ni, nj):
    """cost of a block with ni row antennas against nj column antennas (nj None: diagonal block)"""
    if nj is None:
        ta = (ni + 31) // 32
        return 7 * ta + 12 * (ta * (ta - 1) // 2) + 12 * (ta * (ta + 1) // 2) + 2 * GEN_PER_ROW * 32 * ta
    ci, cj = _cap(ni), _cap(nj)
    if (min(ci, cj), max(ci, cj)) not in ((32, 32), (32, 64), (64, 64), (128, 128)):
        ci = cj = 128
    return 2 * 12 * (ci // 32) * (cj // 32) + 2 * GEN_PER_ROW * (ci + cj)


def plan_tile_shards(bl_ants, Nant, world_size, bl_mp=None, ant_model=None, groups=(128, 64, 32)):
    """
    Deal the blocks of the antenna pair matrix (ops._antenna_blocks: groups of <= g antennas,
    diagonal + cross blocks, per beam-model pair) to `world_size` ranks, longest item first to the
    least-loaded rank, for the group size g -- with cross blocks whole or cut into slabs of 32 row
    antennas (a rank's geometry drops the antennas its shard of a block does not touch) -- that gives the
    smallest maximum load.  bl_ants: antenna-INDEX pairs of all baselines.  Returns dict(group=g,
    rank_bls=[baseline indices (ascending) per rank], load=[cost per rank], order=concatenation of
    rank_bls, inverse=its inverse permutation (gathered[..., inverse] restores the original baseline
    order)), or None when the pair set cannot be represented (a pair listed twice).
    """
    from . import ops
    best = None
    for g in groups:
        blocks = ops._antenna_blocks(bl_ants, Nant, bl_mp, ant_model, group=g)
        if blocks is None:
            return None
        for slabs in (False, True):
            items = []                                 # (cost, baseline slots, antennas touched)
            for blk in blocks:
                d, c = blk['direct'], blk['conj']
                ni = len(blk['ants_i'])
                if blk['ants_j'] is None or not slabs or ni <= 32:
                    slots = np.concatenate([d[d >= 0], c[c >= 0]])
                    items.append((_item_cost(ni, None if blk['ants_j'] is None else len(blk['ants_j'])), slots,
                                  set(blk['ants_i']) | set(blk['ants_j'] or ())))
                else:
                    for r0 in range(0, ni, 32):
                        dd, cc = d[r0:r0 + 32], c[r0:r0 + 32]
                        slots = np.concatenate([dd[dd >= 0], cc[cc >= 0]])
                        if len(slots):
                            cols = np.nonzero(((dd >= 0) | (cc >= 0)).any(0))[0]
                            items.append((_item_cost(min(32, ni - r0), len(cols)), slots,
                                          set(blk['ants_i'][r0:r0 + 32]) | {blk['ants_j'][k] for k in cols}))
            if slabs and len(items) == len(blocks):
                continue                               # nothing was cut: same as the whole-block variant
            load = [0.0] * world_size
            owner = [[] for _ in range(world_size)]
            ants = [set() for _ in range(world_size)]
            for k in sorted(range(len(items)), key=lambda k: -items[k][0]):
                # least-loaded rank; ties go to the rank that already generates these antennas
                r = min(range(world_size), key=lambda r: (round(load[r], 6), -len(ants[r] & items[k][2]), r))
                load[r] += items[k][0]
                owner[r].append(k)
                ants[r] |= items[k][2]
            if best is None or max(load) < 0.98 * max(best['load']):
                rank_bls = [sorted(int(i) for k in owner[r] for i in items[k][1]) for r in range(world_size)]
                best = dict(group=g, rank_bls=rank_bls, load=load, nblocks=[len(o) for o in owner], slabs=slabs)
    if best is None or min(len(b) for b in best['rank_bls']) == 0:
        return None                                    # fewer blocks than ranks: the caller falls back to contiguous blocks
    order = np.asarray([i for bl in best['rank_bls'] for i in bl], dtype=np.int64)
    assert len(order) == len(bl_ants) and len(set(order.tolist())) == len(order)
    inverse = np.empty_like(order)
    inverse[order] = np.arange(len(order))
    best.update(order=order, inverse=inverse)
    return best


# ---------------------------------------------------------------------------------------------
# differentiable all-gather of visibility blocks
# ---------------------------------------------------------------------------------------------
class _GatherHandle:
    """an all-gather in flight: buffers + the backend's work object(s)"""
    def __init__(self, x, counts, dim, group, async_op):
        world = len(counts)
        self.counts, self.dim, self.group = tuple(counts), dim, group
        self.rank = dist.get_rank(group)
        self.shape_in = tuple(x.shape)
        nmax = max(counts)
        self.complex = x.is_complex()
        xd = x.detach()
        self.equal = (min(counts) == nmax and x.shape[dim] == nmax and not _SIMPLE)
        if self.equal:
            # equal blocks: one collective into a [world, ...] buffer, then a single re-layout (none at all
            # when every axis before `dim` has length 1, i.e. the baseline-sharded visibility tensor)
            xin = xd.contiguous()
            buf = torch.view_as_real(xin) if self.complex else xin
            self.out = torch.empty((world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
            self.keep = buf
            self.work = dist.all_gather_into_tensor(self.out.view(-1), buf.reshape(-1), group=group, async_op=async_op)
        else:
            pad = xd
            if x.shape[dim] < nmax:
                padshape = list(x.shape)
                padshape[dim] = nmax - x.shape[dim]
                pad = torch.cat([xd, xd.new_zeros(padshape)], dim=dim)
            pad = pad.contiguous()
            buf = torch.view_as_real(pad) if self.complex else pad
            self.parts = [torch.empty_like(buf) for _ in range(world)]
            self.keep = buf
            self.work = dist.all_gather(self.parts, buf, group=group, async_op=async_op)

    def result(self):
        """wait for the collective (the current stream waits for the backend's) and lay the blocks out"""
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.equal:
            out = torch.view_as_complex(self.out) if self.complex else self.out
            shape = list(self.shape_in)
            shape[self.dim] = len(self.counts) * self.counts[0]
            return out.movedim(0, self.dim).reshape(shape)
        parts = [torch.view_as_complex(p) for p in self.parts] if self.complex else self.parts
        return torch.cat([p.narrow(self.dim, 0, c) for p, c in zip(parts, self.counts)], dim=self.dim)


class _GatherWait(torch.autograd.Function):
    """autograd node of an all-gather: forward = wait + layout; backward = this rank's slice of the
    upstream gradient (no communication)"""
    @staticmethod
    def forward(ctx, x, handle):
        ctx.counts, ctx.rank, ctx.dim = handle.counts, handle.rank, handle.dim
        return handle.result()

    @staticmethod
    def backward(ctx, g):
        s = sum(ctx.counts[:ctx.rank])
        return g.narrow(ctx.dim, s, ctx.counts[ctx.rank]).contiguous(), None


class PendingGather:
    def __init__(self, x, handle, inverse=None):
        self.x, self.handle, self.inverse = x, handle, inverse

    def wait(self):
        """gathered tensor, connected to the autograd graph of the local block"""
        full = _GatherWait.apply(self.x, self.handle)
        if self.inverse is not None:
            full = full.index_select(self.handle.dim, self.inverse)       # tile shards: back to the original order
        return full


def _counts(vis_local, counts, group, dim):
    if counts is not None:
        return tuple(int(c) for c in counts)
    world = dist.get_world_size(group)
    n = torch.tensor([vis_local.shape[dim]], device=vis_local.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    return tuple(int(x.item()) for x in ns)


def all_gather_vis_start(vis_local, counts=None, group=None, dim=2, inverse=None):
    """
    Start the all-gather of per-rank visibility blocks (async on backends that support it: the
    collective waits for the kernels already enqueued on the current stream, later kernels overlap
    it).  `.wait()` returns the full tensor, rank blocks in rank order along `dim` (2 = baseline
    shards, 4 = channel shards), re-ordered by `inverse` (int64 index tensor) when the shards are not
    contiguous blocks of the original baseline order (tile shards).
    """
    counts = _counts(vis_local, counts, group, dim)
    return PendingGather(vis_local, _GatherHandle(vis_local, counts, dim, group, async_op=True), inverse)


def all_gather_vis(vis_local, counts=None, group=None, dim=2, inverse=None):
    """
    Gather per-rank visibility blocks into the full (Npol, Npol, Nbl, Nt, Nf) tensor on every
    rank (blocking form of all_gather_vis_start).  `counts`: block sizes per rank (default:
    exchanged with an all_gather of the local size).
    """
    counts = _counts(vis_local, counts, group, dim)
    return PendingGather(vis_local, _GatherHandle(vis_local, counts, dim, group, async_op=False), inverse).wait()


class _ReduceWait(torch.autograd.Function):
    """autograd node of an all-reduce (sum) of partial visibilities: forward = wait; backward = the upstream gradient as it
    is (d sum / d part = 1 on every rank: no communication)"""
    @staticmethod
    def forward(ctx, x, pending):
        return pending._finish()

    @staticmethod
    def backward(ctx, g):
        return g, None


class PendingReduce:
    """an all-reduce (sum) of this rank's PARTIAL visibilities in flight (pixel partition, SURVEY 8e: every rank contracts its
    slice of the sky pixels for ALL baselines, times and channels; the visibility is the sum of the slices)"""
    def __init__(self, x, group, async_op):
        self.x = x
        self.complex = x.is_complex()
        buf = x.detach().contiguous().clone()
        self.buf = torch.view_as_real(buf) if self.complex else buf
        self.work = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)

    def _finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return torch.view_as_complex(self.buf) if self.complex else self.buf

    def wait(self):
        """summed tensor, connected to the autograd graph of the local partial sum"""
        return _ReduceWait.apply(self.x, self)


def all_reduce_vis_start(vis_partial, group=None):
    """
    Pixel partition: start the (async) sum of the ranks' partial visibilities -- the same tensor shape on every rank;
    `.wait()` returns the full visibilities; differentiable (backward: the upstream gradient, unchanged).  Deterministic for
    a given backend and world size (the backend's reduction order), not bit-identical to the unsharded sum.
    """
    return PendingReduce(vis_partial, group, async_op=True)


def all_reduce_vis(vis_partial, group=None):
    return PendingReduce(vis_partial, group, async_op=False).wait()


# ---------------------------------------------------------------------------------------------
# gradient exchange
# ---------------------------------------------------------------------------------------------
def _block_gather_inplace_ok(g, dim, bounds):
    nmax = max(b - a for a, b in bounds)
    return (not _SIMPLE and min(b - a for a, b in bounds) == nmax and g.is_contiguous() and not g.is_complex()
            and all(n == 1 for n in g.shape[:dim]) and bounds[0][0] == 0 and bounds[-1][1] == g.shape[dim])


def all_gather_block_grads(param, dim, bounds, group=None, async_op=False):
    """
    Gradient exchange for a replicated parameter whose slices along `dim` are each used by exactly
    one rank (channel sharding of per-channel sky / beam parameters): the local .grad is non-zero
    only inside this rank's (start, stop) block, so an all-gather of the blocks rebuilds the full
    gradient with half the bytes of an all-reduce.  Returns the work object when async_op (equal
    blocks only; ragged blocks complete before returning).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    s, e = bounds[rank]
    if param.grad is None:
        param.grad = torch.zeros_like(param)
    nmax = max(b - a for a, b in bounds)
    g = param.grad
    if _block_gather_inplace_ok(g, dim, bounds):
        # equal blocks that are contiguous runs of the gradient in rank order: in-place all-gather
        # (the send block is a copy: an input that aliases the output is legal for RCCL but made a 4-rank gloo
        # rehearsal on a shared GPU crawl; 1/world of the gradient is cheap to copy)
        flat = g.view(-1)
        n = flat.numel() // world
        return dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n].clone(), group=group, async_op=async_op)
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
    return None


BUCKET_BYTES = 1 << 20        # gradients smaller than this share one flat bucket per dtype; larger ones are reduced in place


def _reduce_one(p, group, average, world, async_op):
    """in-place all-reduce of one gradient (through a contiguous copy when .grad is not contiguous)"""
    g = p.grad
    if g.is_contiguous():
        buf = torch.view_as_real(g) if g.is_complex() else g
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return work, ((lambda: buf.div_(world)) if average else None)
    c = g.contiguous()
    buf = torch.view_as_real(c) if c.is_complex() else c
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)

    def finish():
        if average:
            buf.div_(world)
        g.copy_(c)                       # write back THROUGH the parameter's gradient, whatever its strides
    return work, finish


def all_reduce_grads(params, group=None, average=False):
    """
    Sum .grad of the given parameters over ranks, in place.  Large gradients are reduced where they
    are (no staging copy); gradients below BUCKET_BYTES share one flat bucket per dtype so the ring
    runs over few messages.  Non-contiguous gradients go through a contiguous copy and are written
    back through `p.grad`.  Parameters without a .grad on this rank contribute zeros.
    """
    params = [p for p in params if p is not None]
    world = dist.get_world_size(group)
    small = {}
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        if p.grad.numel() * p.grad.element_size() < BUCKET_BYTES:
            small.setdefault((p.grad.dtype, p.grad.device), []).append(p)
            continue
        _, finish = _reduce_one(p, group, average, world, False)
        if finish is not None:
            finish()
    for ps in small.values():
        real = [torch.view_as_real(p.grad.contiguous()) if p.grad.is_complex() else p.grad.contiguous() for p in ps]
        flat = torch.cat([r.reshape(-1) for r in real])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        o = 0
        for p, r in zip(ps, real):
            n = r.numel()
            piece = flat[o:o + n].view(r.shape)
            p.grad.copy_(torch.view_as_complex(piece) if p.grad.is_complex() else piece)
            o += n


class GradSync:
    """
    Gradient exchange started from autograd: when `arm()`ed, the post-accumulate hook of every
    registered parameter marks its gradient final, and the parameter's collective is launched
    asynchronously as soon as every parameter BEFORE it in the registration order has been launched, so the
    exchange overlaps the rest of the backward pass; `finish()` waits for all of them.
      shared parameters (used by every rank)            -> in-place all-reduce (sum)
      block parameters  [(param, axis)] + `bounds`      -> in-place all-gather of the per-rank blocks
    The registration order (shared, then blocks) is the order of the collectives on EVERY rank whatever order
    the hooks fire in: RCCL matches collectives by issue order, and ranks of a baseline-tile partition run
    graphs of different shapes.  That order must therefore be the SAME on every rank: the constructor proves it
    (`verify`: one all-gather of a digest of the planned sequence -- kind, shape, dtype, axis and block bounds of
    every entry -- and a RuntimeError on EVERY rank when the digests differ; under RCCL a mismatch would otherwise be
    a hang or a silent sum of unrelated buffers).  After the first step the order in which the hooks actually fired
    replaces it, but only if all ranks report the same one (`adapted`, again an explicit exchange).  Un-armed backward
    passes (earlier time chunks of a pipelined step) only accumulate.
    """
    def __init__(self, shared=(), blocks=(), bounds=None, group=None, verify=True):
        self.group, self.bounds = group, bounds
        self.shared = [p for p in shared if p is not None]
        self.blocks = list(blocks)
        self.entries = [(p, None) for p in self.shared] + [(p, ax) for p, ax in self.blocks]
        self.index = {id(p): i for i, (p, _) in enumerate(self.entries)}
        self.armed = False
        self.pending, self.ready, self.next = [], set(), 0
        self.fired = []                     # entry indices in the order the hooks fired in the last armed backward
        self.adapted = None                 # None: not decided yet; True / False after the first finish()
        if verify:
            self.verify()
        self.hooks = [p.register_post_accumulate_grad_hook(self._hook) for p, _ in self.entries]

    def plan_digest(self):
        """64-bit digest of the planned collective sequence (what RCCL will be asked to do, in order)"""
        import hashlib
        desc = repr([('reduce' if ax is None else 'gather%d' % ax, tuple(p.shape), str(p.dtype)) for p, ax in self.entries]
                    + [None if self.bounds is None else [tuple(b) for b in self.bounds]])
        return int.from_bytes(hashlib.sha256(desc.encode()).digest()[:8], 'big') >> 1        # fits int64

    def verify(self):
        """every rank must plan the same collectives in the same order; raises on ALL ranks otherwise"""
        world = dist.get_world_size(self.group)
        if world == 1 or not self.entries:
            return
        dev = self.entries[0][0].device
        mine = torch.tensor([self.plan_digest(), len(self.entries)], dtype=torch.int64, device=dev)
        every = torch.empty(world * 2, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(every, mine, group=self.group)
        every = every.view(world, 2).cpu()
        if not bool((every == every[0]).all()):
            raise RuntimeError('GradSync: the ranks plan different gradient collectives (digest, entries per rank: %s); '
                               'register the parameters in the same order on every rank' % every.tolist())

    def _launch(self, i):
        p, ax = self.entries[i]
        if ax is None:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            self.pending.append(_reduce_one(p, self.group, False, 1, True))
        else:
            self.pending.append((all_gather_block_grads(p, ax, self.bounds, self.group, async_op=True), None))

    def _drain(self, everything=False):
        while self.next < len(self.entries) and (everything or self.next in self.ready):
            self._launch(self.next)
            self.next += 1

    def _hook(self, p):
        if self.armed:
            self.ready.add(self.index[id(p)])
            self.fired.append(self.index[id(p)])
            self._drain()

    def arm(self):
        self.armed, self.pending, self.ready, self.next, self.fired = True, [], set(), 0, []

    def finish(self):
        """launch what the backward never reached (or touched out of order), wait for every collective"""
        self.armed = False
        self._drain(everything=True)
        for work, fin in self.pending:
            if work is not None:
                work.wait()
            if fin is not None:
                fin()
        self.pending, self.ready, self.next = [], set(), 0
        if self.adapted is None:
            self._adopt_fired_order()

    def _adopt_fired_order(self):
        """after the first armed backward: if EVERY rank saw the hooks fire in the same order, make that order the
        collective order from now on (a gradient then leaves the moment it is final instead of waiting for entries
        registered before it); one tiny MIN / MAX all-reduce pair, once"""
        n = len(self.entries)
        order = list(dict.fromkeys(self.fired)) + [i for i in range(n) if i not in self.fired]
        self.adapted = False
        if n < 2 or not self.entries:
            return
        dev = self.entries[0][0].device
        lo = torch.tensor(order, dtype=torch.int64, device=dev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if torch.equal(lo, hi) and order != list(range(n)):
            self.entries = [self.entries[i] for i in order]
            self.index = {id(p): i for i, (p, _) in enumerate(self.entries)}
            self.adapted = True

    def remove(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []


def reduce_scalar(x, group=None):
    """sum of a scalar tensor (e.g. a shard-local chi^2) over ranks"""
    y = x.detach().clone()
    dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
    return y


def pipelined_step(forward_chunk, nchunks, loss_fn, gather_start, grad_sync=None, trace=None):
    """
    One forward + backward over `nchunks` time chunks with the collectives overlapped:
        chunk k:  vis_k = forward_chunk(k)                 kernels enqueued
                  gather_start(vis_k)                      async all-gather, behind those kernels
                  loss + backward of chunk k-1             overlaps the gather of chunk k
    and the gradient collectives of `grad_sync` (a GradSync) start inside the LAST chunk's backward.
    loss_fn(full_vis_k, k) -> scalar.  Returns the summed loss (detached).  Gradients accumulate in
    .grad over the chunks (the loss must be a sum over chunks, as chi^2 / sum |V|^2 are).
    trace: optional callable(str) called at the stage boundaries (debugging aid; it may synchronise).
    """
    total = None

    def finish(pending, k, last):
        nonlocal total
        full = pending.wait()
        if trace:
            trace('chunk %d: gathered' % k)
        loss = loss_fn(full, k)
        if last and grad_sync is not None:
            grad_sync.arm()
        loss.backward()
        if trace:
            trace('chunk %d: backward done' % k)
        total = loss.detach() if total is None else total + loss.detach()

    pending = None
    for k in range(nchunks):
        vis = forward_chunk(k)
        if trace:
            trace('chunk %d: forward done' % k)
        nxt = gather_start(vis)
        if pending is not None:
            finish(pending, k - 1, False)
        pending = nxt
    finish(pending, nchunks - 1, True)
    if grad_sync is not None:
        grad_sync.finish()
        if trace:
            trace('gradient collectives finished')
    return total


class ShardedRIME:
    """
    Baseline-sharded driver around a RIME factory.  `make_rime(sim_bls)` must build a RIME for the
    given baseline list with this rank's replica of the sky / beam models.  forward() returns
    this rank's VisData block, or the gathered visibilities with gather=True.
    tiles: deal whole blocks of the antenna pair matrix to the ranks (plan_tile_shards) instead of
    contiguous baseline blocks -- shards the matrix-core work; `array` (ArrayModel) and optionally the
    beam's `ant2beam` are needed to index the antennas.
    """
    def __init__(self, make_rime, all_bls, group=None, tiles=False, array=None, ant2beam=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.all_bls = list(all_bls)
        self.inverse = None
        self.plan = None
        if tiles:
            idx = array._ant_idx
            bl_ants = [(idx[a], idx[b]) for a, b in self.all_bls]
            ant_model = bl_mp = None
            if ant2beam is not None and len(set(ant2beam.values())) > 1:
                ant_model = [ant2beam[a] for a in array.ants]
                uniq = sorted({(ant2beam[a], ant2beam[b]) for a, b in self.all_bls})
                bl_mp = [uniq.index((ant2beam[a], ant2beam[b])) for a, b in self.all_bls]
            self.plan = plan_tile_shards(bl_ants, len(array.ants), self.world, bl_mp, ant_model)
        if self.plan is not None:
            self.counts = [len(b) for b in self.plan['rank_bls']]
            self.local_bls = [self.all_bls[i] for i in self.plan['rank_bls'][self.rank]]
            self.inverse = torch.as_tensor(self.plan['inverse'])
        else:
            self.bounds = shard_bounds(len(self.all_bls), self.world)
            self.counts = [e - s for s, e in self.bounds]
            s, e = self.bounds[self.rank]
            self.local_bls = self.all_bls[s:e]
        self.rime = make_rime(self.local_bls)
        if self.plan is not None:
            self.rime.mfma_group = self.plan['group']        # the geometry's blocks = the plan's blocks
            self.rime.mfma_mode = True

    def gather_start(self, vis):
        inv = None if self.inverse is None else self.inverse.to(vis.device)
        return all_gather_vis_start(vis, self.counts, self.group, dim=2, inverse=inv)

    def forward(self, gather=False, **kw):
        vd = self.rime(**kw)
        if gather:
            vd.data = self.gather_start(vd.data).wait()
            vd._set_bls(self.all_bls)
        return vd

    __call__ = forward

    def sync_grads(self, params=None):
        if params is None:
            params = [p for p in self.rime.parameters()]
        all_reduce_grads(params, self.group)


class DistributedLogProb:
    """
    Data-parallel posterior: ONE optim.LogProb per rank (= per GPU), each predicting and comparing against its own
    shard of the target data (baselines, channels or times) from a replicated main-parameter tensor.  The
    reference's DistributedLogProb (optim.py:1391-1629) holds a list of LogProbs on several devices inside one
    process, sums their losses and adds their main_params gradients on a master device by hand (:1539-1566); here
    the main-parameter tensor itself is the flat bucket: closure() runs the local closure and finishes with ONE
    in-place all-reduce of main_params.grad over RCCL plus a scalar all-reduce of the loss.  As there, the prior
    is evaluated on one rank only ('post' becomes 'like' on the others), and main_params starts identical on
    every rank (broadcast from rank 0).
    """
    def __init__(self, prob, group=None):
        self.prob, self.group = prob, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        assert prob.main_params is not None and prob.main_params.is_leaf, "run LogProb.set_main_params() first"
        src = dist.get_global_rank(group, 0) if group is not None else 0
        with torch.no_grad():
            dist.broadcast(prob.main_params.data, src=src, group=group)
        prob.send_main_params()
        if prob.compute == 'post' and self.rank != 0:
            prob.compute = 'like'                      # the prior is counted once

    @property
    def main_params(self):
        return self.prob.main_params

    @property
    def Nbatch(self):
        return self.prob.Nbatch

    @property
    def closure_eval(self):
        return self.prob.closure_eval

    def closure(self, **kwargs):
        """summed loss over ranks; main_params.grad = the sum of the ranks' gradients, on every rank"""
        mp = self.prob.main_params
        if self.prob.compute == 'prior' and self.rank != 0:
            mp.grad = None
            loss = torch.zeros(1, device=mp.device, dtype=torch.float64)
        else:
            loss = self.prob.closure(**kwargs)
        if torch.is_grad_enabled():
            if mp.grad is None:
                mp.grad = torch.zeros_like(mp)
            dist.all_reduce(mp.grad, op=dist.ReduceOp.SUM, group=self.group)
        # float64 on every rank: a (1,)-shaped float32 prior term demotes a 0-dim float64 likelihood on the prior's rank
        return reduce_scalar(loss.reshape(-1)[:1].to(mp.device, torch.float64), self.group)


# ---------------------------------------------------------------------------------------------
# the same two collectives behind the library's C ABI (include/rime_hip.h: rime_comm_*)
# ---------------------------------------------------------------------------------------------
class RcclComm:
    """
    An RCCL communicator owned through the C ABI (rime_comm_init), for hosts that do not want
    torch.distributed on the data path: `allgather_vis` / `reduce_grads` enqueue ncclAllGather /
    ncclAllReduce on the CURRENT torch stream with raw device pointers.  One rank creates the 128-byte
    unique id (`RcclComm.unique_id()`), every rank passes it to the constructor; `from_group` does that
    exchange over an existing torch.distributed group (any backend, e.g. gloo: control plane only).
    """
    def __init__(self, nranks, rank, unique_id):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        assert len(unique_id) == 128
        self.nranks, self.rank = int(nranks), int(rank)
        self._comm = ctypes.c_void_p()
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(_lib.lib.rime_comm_init(ctypes.byref(self._comm), self.nranks, self.rank,
                                           ctypes.cast(buf, ctypes.c_void_p)), 'rime_comm_init')

    @staticmethod
    def unique_id():
        import ctypes
        from . import _lib
        buf = (ctypes.c_char * 128)()
        _lib.check(_lib.lib.rime_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), 'rime_comm_unique_id')
        return bytes(buf.raw)

    @classmethod
    def from_group(cls, group=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(world, rank, box[0])

    def _stream(self):
        return self._ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def allgather_vis(self, vis_local):
        """equal blocks: returns a (nranks,) + vis_local.shape complex tensor, blocks in rank order"""
        v = vis_local.detach().contiguous()
        assert v.is_cuda and v.is_complex()
        out = torch.empty((self.nranks,) + tuple(v.shape), dtype=v.dtype, device=v.device)
        code = 0 if v.dtype == torch.complex64 else 1
        self._lib.check(self._lib.lib.rime_comm_allgather_vis(self._comm, code, self._ct.c_void_p(v.data_ptr()),
                                                             self._ct.c_void_p(out.data_ptr()), v.numel(), self._stream()),
                        'rime_comm_allgather_vis')
        return out

    def reduce_grads(self, tensors):
        """in-place sum over ranks of contiguous real or complex float32 / float64 tensors"""
        for t in tensors:
            assert t.is_cuda and t.is_contiguous()
            r = torch.view_as_real(t) if t.is_complex() else t
            code = 0 if r.dtype == torch.float32 else 1
            self._lib.check(self._lib.lib.rime_comm_reduce_grads(self._comm, code, self._ct.c_void_p(r.data_ptr()), r.numel(),
                                                                self._stream()), 'rime_comm_reduce_grads')

    def close(self):
        if self._comm:
            self._lib.check(self._lib.lib.rime_comm_destroy(self._comm), 'rime_comm_destroy')
            self._comm = self._ct.c_void_p()
