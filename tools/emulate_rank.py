"""Per-rank compute time of the channel-sharded C4 step for world sizes 1, 2, 4, 8 on ONE GPU:
rank 0's shard, no collectives (those are measured by the driver's multi-GPU runs).
python tools/emulate_rank.py [workload] [nt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bayeslim_amd import dist as rdist

wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
cfg = bench.WORKLOADS[wl]
nt = int(sys.argv[2]) if len(sys.argv) > 2 else cfg['nt']
dev = torch.device('cuda', 0)
inp = bench.build_inputs(wl, nt)
bls = bench.all_baselines(inp)
base = None
for world in (1, 2, 4, 8):
    fblock = rdist.shard_bounds(cfg['Nf'], world)[0]
    rime, params, attach, _ = bench.build_model(inp, dev, bls, fblock=fblock)

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
    base = base or ms
    print('world %d: channels %3d per rank, %.2f ms/step (host enqueue %.2f ms), compute-only speedup %.2fx' % (
        world, fblock[1] - fblock[0], ms, enq, base / ms))
    del rime, params
    torch.cuda.empty_cache()
