"""Where the HOST time of one rank step goes (4.1 of 12.5 ms at world 8): cProfile of the enqueue of
one forward + backward of the channel-sharded C4 rank share (32 of 256 channels) on one GPU, no collectives.
python tools/host_profile.py [workload] [world] [nt]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bayeslim_amd import dist as rdist

wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = bench.WORKLOADS[wl]
nt = int(sys.argv[3]) if len(sys.argv) > 3 else cfg['nt']
dev = torch.device('cuda', 0)
inp = bench.build_inputs(wl, nt)
bls = bench.all_baselines(inp)
fblock = rdist.shard_bounds(cfg['Nf'], world)[0]
rime, params, attach, _ = bench.build_model(inp, dev, bls, fblock=fblock)


def step():
    for p in params:
        p.grad = None
    attach()
    v = rime().data
    (v.real ** 2 + v.imag ** 2).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
enq = []
for _ in range(5):
    t0 = time.perf_counter(); step(); enq.append((time.perf_counter() - t0) * 1e3); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print('world %d rank share: %.2f ms/step, host enqueue %s ms (min %.2f)' % (world, (time.perf_counter() - t0) / 5 * 1e3,
                                                                          ' '.join('%.2f' % e for e in enq), min(enq)))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(30)
