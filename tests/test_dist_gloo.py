"""
Multi-process CPU tests (gloo, world_size 2) of the N > 1 path: baseline sharding, the
differentiable all-gather of visibility blocks and the bucketed gradient all-reduce.  The local
"simulation" here is the CPU oracle (tests may use it as the checker); on the GPU the same
collectives wrap the HIP path (bench.py --gpus N).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayeslim_amd import dist as rdist
from oracle import rime_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(even=False):
    rng = np.random.default_rng(0)
    Nbl, Nf, P, Nt = 7, 5, 40, 2            # 7 baselines over 2 ranks: ragged shards (4 + 3)
    if even:
        Nbl, Nf = 6, 4                      # equal blocks: the single-buffer all-gather paths
    T = lambda x: torch.as_tensor(x, dtype=torch.float64)
    blvecs = T(rng.normal(0, 30, (Nbl, 3)))
    freqs = T(np.linspace(120e6, 180e6, Nf))
    zen = T(np.rad2deg(np.arccos(rng.uniform(0, 1, (Nt, P)))))
    az = T(rng.uniform(0, 360, (Nt, P)))
    sky = T(rng.normal(size=(1, 1, Nf, P)))
    beam = T(np.abs(rng.normal(size=(1, 1, 1, Nf, P))))
    return blvecs, freqs, zen, az, sky, beam


def _simulate(blvecs, freqs, zen, az, sky, beam):
    out = []
    for t in range(zen.shape[0]):
        psky = orc.apply_beam(beam, sky, [(0, 0)] * len(blvecs), True)
        out.append(orc.prod_and_sum(psky, blvecs, zen[t], az[t], freqs))
    return torch.stack(out, dim=3)           # (1, 1, Nbl, Nt, Nf)


def _worker(rank, world, port, q, even=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        blvecs, freqs, zen, az, sky, beam = _problem(even)
        sky = sky.clone().requires_grad_(True)
        beam = beam.clone().requires_grad_(True)
        bounds = rdist.shard_bounds(len(blvecs), world)
        s, e = bounds[rank]
        assert rdist.shard_baselines(list(range(len(blvecs)))) == list(range(s, e))
        local = _simulate(blvecs[s:e], freqs, zen, az, sky, beam)
        full = rdist.all_gather_vis(local)                       # counts exchanged
        full2 = rdist.all_gather_vis(local, [b - a for a, b in bounds])
        assert torch.equal(full, full2)
        w = torch.as_tensor(np.random.default_rng(9).normal(size=tuple(full.shape)))
        loss = (w * (full.real ** 2 + full.imag ** 2)).sum()
        loss.backward()
        rdist.all_reduce_grads([sky, beam, None])
        tot = rdist.reduce_scalar(torch.tensor(float(rank + 1)))
        q.put((rank, full.detach().numpy(), sky.grad.numpy(), beam.grad.numpy(), float(tot)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('even', [False, True])
def test_sharded_forward_backward_equals_single_process(even):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, even)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference
    blvecs, freqs, zen, az, sky, beam = _problem(even)
    sky = sky.clone().requires_grad_(True)
    beam = beam.clone().requires_grad_(True)
    full = _simulate(blvecs, freqs, zen, az, sky, beam)
    w = torch.as_tensor(np.random.default_rng(9).normal(size=tuple(full.shape)))
    (w * (full.real ** 2 + full.imag ** 2)).sum().backward()
    for rank, v, gs, gb, tot in res:
        assert np.abs(v - full.detach().numpy()).max() < 1e-12          # gathered layout == single-GPU layout
        assert np.abs(gs - sky.grad.numpy()).max() < 1e-9 * np.abs(sky.grad.numpy()).max()
        assert np.abs(gb - beam.grad.numpy()).max() < 1e-9 * np.abs(beam.grad.numpy()).max()
        assert tot == 3.0


def _worker_freq(rank, world, port, q, even=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        blvecs, freqs, zen, az, sky, beam = _problem(even)
        sky = sky.clone().requires_grad_(True)          # replicated full parameters
        beam = beam.clone().requires_grad_(True)
        bounds = rdist.shard_bounds(len(freqs), world)  # 5 channels over 2 ranks: 3 + 2
        s, e = bounds[rank]
        local = _simulate(blvecs, freqs[s:e], zen, az, sky[:, :, s:e], beam[:, :, :, s:e])
        full = rdist.all_gather_vis(local, [b - a for a, b in bounds], dim=4)
        w = torch.as_tensor(np.random.default_rng(9).normal(size=tuple(full.shape)))
        (w * (full.real ** 2 + full.imag ** 2)).sum().backward()
        rdist.all_gather_block_grads(sky, 2, bounds)
        rdist.all_gather_block_grads(beam, 3, bounds)
        q.put((rank, full.detach().numpy(), sky.grad.numpy(), beam.grad.numpy(), 3.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('even', [False, True])
def test_frequency_sharded_forward_backward_equals_single_process(even):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_freq, args=(r, world, port, q, even)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blvecs, freqs, zen, az, sky, beam = _problem(even)
    sky = sky.clone().requires_grad_(True)
    beam = beam.clone().requires_grad_(True)
    full = _simulate(blvecs, freqs, zen, az, sky, beam)
    w = torch.as_tensor(np.random.default_rng(9).normal(size=tuple(full.shape)))
    (w * (full.real ** 2 + full.imag ** 2)).sum().backward()
    for rank, v, gs, gb, _ in res:
        assert np.abs(v - full.detach().numpy()).max() < 1e-12
        assert np.abs(gs - sky.grad.numpy()).max() < 1e-9 * np.abs(sky.grad.numpy()).max()
        assert np.abs(gb - beam.grad.numpy()).max() < 1e-9 * np.abs(beam.grad.numpy()).max()


def _worker_pipeline(rank, world, port, q, tiles):
    """pipelined step: async gathers of time chunks + GradSync hooks, contiguous or tile shards"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ant, pairs, freqs, zen, az, sky, beam = _ant_problem()
        sky = sky.clone().requires_grad_(True)
        beam = beam.clone().requires_grad_(True)
        blvecs = torch.stack([ant[b] - ant[a] for a, b in pairs])
        if tiles:
            plan = rdist.plan_tile_shards(pairs, len(ant), world)
            mine = plan['rank_bls'][rank]
            counts = [len(b) for b in plan['rank_bls']]
            inverse = torch.as_tensor(plan['inverse'])
        else:
            s, e = rdist.shard_bounds(len(pairs), world)[rank]
            mine, counts, inverse = list(range(s, e)), None, None
        Nt = zen.shape[0]
        w = torch.as_tensor(np.random.default_rng(9).normal(size=(1, 1, len(pairs), Nt, len(freqs))))
        sync = rdist.GradSync(shared=[sky, beam])

        def forward_chunk(k):
            return _simulate(blvecs[mine], freqs, zen[k:k + 1], az[k:k + 1], sky, beam)

        def gather_start(v):
            return rdist.all_gather_vis_start(v, counts, dim=2, inverse=inverse)

        def loss_fn(full, k):
            return (w[:, :, :, k:k + 1] * (full.real ** 2 + full.imag ** 2)).sum()

        tot = rdist.pipelined_step(forward_chunk, Nt, loss_fn, gather_start, sync)
        sync.remove()
        q.put((rank, float(tot), sky.grad.numpy(), beam.grad.numpy()))
    finally:
        dist.destroy_process_group()


def _ant_problem():
    rng = np.random.default_rng(3)
    Nant, Nf, P, Nt = 70, 3, 30, 3
    T = lambda x: torch.as_tensor(x, dtype=torch.float64)
    ant = T(rng.normal(0, 40, (Nant, 3)))
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    freqs = T(np.linspace(120e6, 180e6, Nf))
    zen = T(np.rad2deg(np.arccos(rng.uniform(0, 1, (Nt, P)))))
    az = T(rng.uniform(0, 360, (Nt, P)))
    sky = T(rng.normal(size=(1, 1, Nf, P)))
    beam = T(np.abs(rng.normal(size=(1, 1, 1, Nf, P))))
    return ant, pairs, freqs, zen, az, sky, beam


@pytest.mark.parametrize('tiles, world', [(False, 2), (True, 2), (True, 4), (False, 3)])
def test_pipelined_step_overlapped_collectives_equal_single_process(tiles, world):
    # world 4 with tile shards (uneven blocks of the pair matrix, inverse permutation over four ranks) and world 3 with
    # contiguous shards (2415 baselines do not divide): rank counts beyond the pair the other tests use
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, world, port, q, tiles)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ant, pairs, freqs, zen, az, sky, beam = _ant_problem()
    sky = sky.clone().requires_grad_(True)
    beam = beam.clone().requires_grad_(True)
    blvecs = torch.stack([ant[b] - ant[a] for a, b in pairs])
    full = _simulate(blvecs, freqs, zen, az, sky, beam)
    w = torch.as_tensor(np.random.default_rng(9).normal(size=tuple(full.shape)))
    loss = (w * (full.real ** 2 + full.imag ** 2)).sum()
    loss.backward()
    for rank, tot, gs, gb in res:
        assert abs(tot - float(loss.detach())) < 1e-9 * abs(float(loss.detach()))
        assert np.abs(gs - sky.grad.numpy()).max() < 1e-9 * np.abs(sky.grad.numpy()).max()
        assert np.abs(gb - beam.grad.numpy()).max() < 1e-9 * np.abs(beam.grad.numpy()).max()


def _worker_noncontig(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        big = torch.zeros(600, 700, dtype=torch.float64, requires_grad=True)        # > BUCKET_BYTES: reduced in place
        small = torch.zeros(5, 7, dtype=torch.complex128, requires_grad=True)        # shares the flat bucket
        gb = torch.arange(600 * 700, dtype=torch.float64).reshape(700, 600).t() * (rank + 1)     # transposed strides
        gs = (torch.arange(35, dtype=torch.float64).reshape(7, 5).t() * (1 + 2j)) * (rank + 1)
        big.grad, small.grad = gb, gs
        assert not big.grad.is_contiguous() and not small.grad.is_contiguous()
        rdist.all_reduce_grads([big, small])
        q.put((rank, big.grad.numpy().copy(), small.grad.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _worker_pix(rank, world, port, q):
    """pixel partition: every rank contracts every world-th sky pixel for ALL baselines, times and channels;
    forward = differentiable all-reduce of the partial visibilities (async, per time chunk), backward = all-reduce of the
    gradients (full-size leaves, each rank's sky gradient non-zero on its own pixels only)"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        blvecs, freqs, zen, az, sky, beam = _problem()
        sky = sky.clone().requires_grad_(True)
        beam = beam.clone().requires_grad_(True)
        sel = slice(rank, None, world)
        Nt = zen.shape[0]
        w = torch.as_tensor(np.random.default_rng(9).normal(size=(1, 1, len(blvecs), Nt, len(freqs))))
        sync = rdist.GradSync(shared=[sky, beam])

        def forward_chunk(k):
            return _simulate(blvecs, freqs, zen[k:k + 1, sel], az[k:k + 1, sel], sky[..., sel], beam[..., sel])

        def loss_fn(full, k):
            return (w[:, :, :, k:k + 1] * (full.real ** 2 + full.imag ** 2)).sum()

        tot = rdist.pipelined_step(forward_chunk, Nt, loss_fn, rdist.all_reduce_vis_start, sync)
        sync.remove()
        blocking = rdist.all_reduce_vis(forward_chunk(0).detach())
        q.put((rank, float(tot), sky.grad.numpy(), beam.grad.numpy(), blocking.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_pixel_partition_equals_single_process(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pix, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blvecs, freqs, zen, az, sky, beam = _problem()
    sky = sky.clone().requires_grad_(True)
    beam = beam.clone().requires_grad_(True)
    full = _simulate(blvecs, freqs, zen, az, sky, beam)
    w = torch.as_tensor(np.random.default_rng(9).normal(size=tuple(full.shape)))
    loss = (w * (full.real ** 2 + full.imag ** 2)).sum()
    loss.backward()
    for rank, tot, gs, gb, v0 in res:
        assert abs(tot - float(loss)) < 1e-9 * abs(float(loss))
        assert np.abs(v0 - full.detach().numpy()[:, :, :, :1]).max() < 1e-12 * np.abs(full.detach().numpy()).max()
        assert np.abs(gs - sky.grad.numpy()).max() < 1e-9 * np.abs(sky.grad.numpy()).max()
        assert np.abs(gb - beam.grad.numpy()).max() < 1e-9 * np.abs(beam.grad.numpy()).max()


def test_all_reduce_grads_writes_back_through_noncontiguous_grads():
    """a transposed .grad (custom backwards return such views) must receive the reduced values"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_noncontig, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eb = np.arange(600 * 700, dtype=np.float64).reshape(700, 600).T * 3
    es = np.arange(35, dtype=np.float64).reshape(7, 5).T * (1 + 2j) * 3
    for rank, b, s_ in res:
        assert np.array_equal(b, eb) and np.array_equal(s_, es)


def _worker_hook_order(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        a = torch.full((300, 400), 1.0, dtype=torch.float64, requires_grad=True)     # different sizes: a mismatched
        b = torch.full((50,), 2.0, dtype=torch.float64, requires_grad=True)          # pair of all-reduces would fail
        c = torch.zeros(4, 6, dtype=torch.float64, requires_grad=True)               # never touched by the loss
        sync = rdist.GradSync(shared=[a, b, c])
        # the two ranks build their graphs in opposite orders, so the post-accumulate hooks of a and b fire in
        # opposite orders (autograd runs the most recently created branch first)
        if rank == 0:
            loss = (a * 3.0).sum() + (b * 5.0).sum()
        else:
            loss = (b * 5.0).sum() + (a * 3.0).sum()
        sync.arm()
        loss.backward()
        fired = list(sync.fired)
        sync.finish()
        assert sync.adapted is False                     # the ranks disagree: registration order stays
        out = (rank, fired, a.grad.numpy().copy(), b.grad.numpy().copy(), c.grad.numpy().copy())
        # same graph on both ranks, hooks fire b before a: that order is adopted after the first step
        sync.remove()
        for p in (a, b, c):
            p.grad = None
        sync2 = rdist.GradSync(shared=[a, b, c])
        for step in range(2):
            for p in (a, b, c):
                p.grad = None
            sync2.arm()
            ((a * 3.0).sum() + (b * 5.0).sum()).backward()
            sync2.finish()
            assert sync2.adapted is True and [p is q_ for (p, _), q_ in zip(sync2.entries, (b, a, c))] == [True] * 3
            assert torch.equal(a.grad, torch.full((300, 400), 6.0, dtype=torch.float64))
            assert torch.equal(b.grad, torch.full((50,), 10.0, dtype=torch.float64))
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_grad_sync_collective_order_is_rank_invariant():
    """hooks that fire in different orders on different ranks must still issue the collectives in one order"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hook_order, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] != res[1][1], 'the test did not produce different hook orders: %r' % (res[0][1],)
    for rank, fired, ga, gb, gc in res:
        assert np.array_equal(ga, np.full((300, 400), 6.0)) and np.array_equal(gb, np.full((50,), 10.0))
        assert np.array_equal(gc, np.zeros((4, 6)))


def _worker_misordered(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        a = torch.nn.Parameter(torch.zeros(300, 400, dtype=torch.float64))
        b = torch.nn.Parameter(torch.zeros(50, dtype=torch.float64))
        order = [a, b] if rank == 0 else [b, a]              # rank 1 registers the collectives the other way round
        try:
            rdist.GradSync(shared=order)
            q.put((rank, 'no error'))
        except RuntimeError as e:
            q.put((rank, 'raised: ' + str(e)))
        # same order, but rank 1 plans different block bounds for a block parameter
        c = torch.nn.Parameter(torch.zeros(1, 1, 8, 5, dtype=torch.float64))
        bounds = [(0, 4), (4, 8)] if rank == 0 else [(0, 3), (3, 8)]
        try:
            rdist.GradSync(shared=[b], blocks=[(c, 2)], bounds=bounds)
            q.put((rank, 'no error'))
        except RuntimeError as e:
            q.put((rank, 'raised: ' + str(e)))
        # and an identical plan passes
        rdist.GradSync(shared=[a, b], blocks=[(c, 2)], bounds=[(0, 4), (4, 8)]).remove()
        q.put((rank, 'ok'))
    finally:
        dist.destroy_process_group()


def test_grad_sync_misordered_rank_raises_instead_of_hanging():
    """a rank that registers its gradient collectives in another order (or with other block bounds) makes the
    constructor raise on EVERY rank; under RCCL the mismatch would be a hang"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_misordered, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(3 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        mine = [m for r, m in res if r == rank]
        assert mine[0].startswith('raised: GradSync: the ranks plan different') and mine[1].startswith('raised:'), mine
        assert mine[2] == 'ok'


def test_tile_shard_plan_falls_back_when_a_rank_would_be_empty():
    """hex-37 over 8 ranks has fewer blocks than ranks: no plan (callers use contiguous baseline blocks)"""
    pairs = [(i, j) for i in range(37) for j in range(i + 1, 37)]
    assert rdist.plan_tile_shards(pairs, 37, 8) is None
    assert rdist.plan_tile_shards(pairs, 37, 2) is not None


def _toy_logprob(shard, nshard):
    """LogProb over a shard of a 6-visibility toy problem with two minibatches (shard None: the whole problem)"""
    from bayeslim_amd import optim, dataset, utils
    w = torch.tensor([1.0, 2.0, 3.0, -1.0, 0.5, 2.5], dtype=torch.float64)
    d = torch.tensor([0.5, 3.0, 10.0, -2.0, 1.0, 4.0], dtype=torch.float64)
    ic = torch.tensor([1.0, 0.5, 2.0, 1.5, 0.7, 3.0], dtype=torch.float64)
    sl = slice(None) if shard is None else slice(shard * 6 // nshard, (shard + 1) * 6 // nshard)

    class Toy(utils.Module):
        def __init__(self):
            super().__init__(name='toy')
            self.params = torch.nn.Parameter(torch.tensor([[1.0, 2.0, 3.0], [0.5, -1.0, 2.0]], dtype=torch.float64))
            self.Nbatch, self.batch_idx = 2, 0

        def forward(self, inp=None, prior_cache=None, **kw):
            self.eval_prior(prior_cache)
            p = self.params[self.batch_idx]
            td = dataset.TensorData()
            td.data = (torch.cat([p, p ** 2]) * w)[sl]                # every parameter reaches every shard
            return td

    toy = Toy()
    toy.set_priors(priors_inp_params=[optim.LogGaussPrior(torch.zeros(2, 3, dtype=torch.float64),
                                                          torch.full((2, 3), 4.0, dtype=torch.float64))])
    tds = []
    for i in range(2):
        td = dataset.TensorData()
        td.data = (d * (i + 1))[sl]
        td.set_cov(None, None, icov=ic[sl])
        tds.append(td)
    prob = optim.LogProb(toy, dataset.Dataset(tds), complex_circular=False)
    prob.set_main_params(['params'])
    return prob


def _worker_dist_logprob(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_default_dtype(torch.float64)       # LogProb's zero-initialised sums take the default dtype, as the reference's
        prob = _toy_logprob(rank, world)
        if rank == 1:
            with torch.no_grad():
                prob.main_params += 5.0                  # must be overwritten by rank 0's values
        dprob = rdist.DistributedLogProb(prob)
        assert prob.compute == ('post' if rank == 0 else 'like')
        loss = dprob.closure()
        q.put((rank, float(loss), dprob.main_params.detach().numpy().copy(), dprob.main_params.grad.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_distributed_logprob_equals_single_process():
    """one LogProb per rank on its shard of the data: summed loss and gradient of the main-parameter tensor equal the
    unsharded LogProb's (the pattern of optim.py:1539-1566), the prior counted once"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dist_logprob, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = _toy_logprob(None, 1)
        loss = float(ref.closure())
    finally:
        torch.set_default_dtype(old)
    # the two shards each carry half of the likelihood normalisation constants: the summed loss is the same
    for rank, l, mpv, g in res:
        assert abs(l - loss) < 1e-10 * abs(loss), (rank, l, loss)
        assert np.allclose(mpv, ref.main_params.detach().numpy())
        assert np.allclose(g, ref.main_params.grad.numpy(), rtol=1e-12, atol=1e-12)


def test_tile_shard_plan():
    """whole blocks of the antenna pair matrix per rank: a partition of the baselines, balanced cost,
    and the inverse permutation restores the original order"""
    for Nant, world in [(128, 8), (128, 4), (128, 2), (512, 8), (70, 4), (37, 2)]:
        pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
        plan = rdist.plan_tile_shards(pairs, Nant, world)
        allb = sorted(i for bl in plan['rank_bls'] for i in bl)
        assert allb == list(range(len(pairs)))
        cat = np.concatenate([np.asarray(b, dtype=np.int64) for b in plan['rank_bls']])
        assert np.array_equal(cat[plan['inverse']], np.arange(len(pairs)))
        load = np.asarray(plan['load'])
        assert load.min() > 0
        if Nant >= 128:
            assert load.max() / load.mean() < 1.25, (Nant, world, load)
    # 128 antennas over 8 ranks: one-tile blocks (groups of 32), six cross tiles + two ranks with two diagonal tiles
    plan = rdist.plan_tile_shards([(i, j) for i in range(128) for j in range(i + 1, 128)], 128, 8)
    assert plan['group'] == 32 and sorted(plan['nblocks']) == [1] * 6 + [2] * 2
    assert sorted(len(b) for b in plan['rank_bls']) == [992] * 2 + [1024] * 6
    assert rdist.plan_tile_shards([(0, 1), (0, 1)], 4, 2) is None


def test_shard_bounds():
    assert rdist.shard_bounds(8128, 8) == [(i * 1016, (i + 1) * 1016) for i in range(8)]
    b = rdist.shard_bounds(171, 4)
    assert b[0] == (0, 43) and b[-1][1] == 171 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert sorted(e - s for s, e in b) == [42, 43, 43, 43]
    assert rdist.shard_bounds(3, 8)[3:] == [(3, 3)] * 5
