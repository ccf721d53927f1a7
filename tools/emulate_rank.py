"""Per-rank COMPUTE time of the sharded C4 step for world sizes 1, 2, 4, 8 on ONE GPU, no collectives
(those are measured by the driver's multi-GPU runs): the slowest rank's shard of
  freq : contiguous channel blocks (rank 0's block; all blocks are equal)
  bl   : baseline-tile shards (dist.plan_tile_shards): the rank with the largest planned load
  pix  : every world-th sky pixel / point source (round 5, SURVEY 8e's third axis): rank 0's share (all shares are equal)
python tools/emulate_rank.py [workload] [nt] [nf]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bayeslim_amd import dist as rdist

wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
cfg = bench.WORKLOADS[wl]
if len(sys.argv) > 3:                                   # third argument: number of channels (e.g. c5 with 32 of its 512)
    cfg = dict(cfg, Nf=int(sys.argv[3]))
    bench.WORKLOADS[wl] = cfg
nt = int(sys.argv[2]) if len(sys.argv) > 2 else cfg['nt']
dev = torch.device('cuda', 0)
inp = bench.build_inputs(wl, nt)
bls = bench.all_baselines(inp)
idx = {a: i for i, a in enumerate(inp['ants'])}
bl_ants = [(idx[a], idx[b]) for a, b in bls]


def timed(rime, params, attach):
    def step():
        for p in params:
            p.grad = None
        attach()
        v = rime().data
        (v.real ** 2 + v.imag ** 2).sum().backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    t0 = time.perf_counter(); step(); enq = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
    return ms, enq


base = {}
for mode in ('freq', 'bl', 'pix'):
    for world in (1, 2, 4, 8):
        if mode == 'pix':
            rime, params, attach, _ = bench.build_model(inp, dev, bls, pblock=(0, world))
            what = 'every %d-th sky pixel / point source' % world
        elif mode == 'freq':
            fblock = rdist.shard_bounds(cfg['Nf'], world)[0]
            rime, params, attach, _ = bench.build_model(inp, dev, bls, fblock=fblock)
            what = 'channels %3d per rank' % (fblock[1] - fblock[0])
        else:
            plan = rdist.plan_tile_shards(bl_ants, len(inp['ants']), world)
            if plan is None:                                # fewer blocks than ranks (HERA-19): contiguous baseline blocks
                s_, e_ = rdist.shard_bounds(len(bls), world)[0]
                rime, params, attach, _ = bench.build_model(inp, dev, bls[s_:e_])
                what = 'contiguous block of %d baselines' % (e_ - s_)
            else:
                r = int(np.argmax(plan['load']))
                rime, params, attach, _ = bench.build_model(inp, dev, [bls[i] for i in plan['rank_bls'][r]])
                rime.mfma_group, rime.mfma_mode = plan['group'], True
                what = 'rank %d of the tile plan: groups of %d antennas, %d block(s), %d baselines, planned load %.0f of %.0f' % (
                    r, plan['group'], plan['nblocks'][r], len(plan['rank_bls'][r]), plan['load'][r], sum(plan['load']))
        ms, enq = timed(rime, params, attach)
        base.setdefault(mode, ms)
        print('%-4s world %d: %s, %.2f ms/step (host enqueue %.2f ms), compute-only speedup %.2fx' % (
            mode, world, what, ms, enq, base[mode] / ms))
        del rime, params, attach
        torch.cuda.empty_cache()
