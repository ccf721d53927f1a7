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


def test_shard_bounds():
    assert rdist.shard_bounds(8128, 8) == [(i * 1016, (i + 1) * 1016) for i in range(8)]
    b = rdist.shard_bounds(171, 4)
    assert b[0] == (0, 43) and b[-1][1] == 171 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert sorted(e - s for s, e in b) == [42, 43, 43, 43]
    assert rdist.shard_bounds(3, 8)[3:] == [(3, 3)] * 5
