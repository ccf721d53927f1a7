"""Where does a bench step spend its time?  Host wall-clock per section (with syncs) + kernel gaps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bayeslim_amd import ops

wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
nt = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[wl]['nt']
dev = torch.device('cuda', 0)
inp = bench.build_inputs(wl, nt)
bls = bench.all_baselines(inp)
rime, params, attach, per_channel = bench.build_model(inp, dev, bls)


def step():
    for p in params:
        p.grad = None
    attach()
    vd = rime()
    v = vd.data
    loss = (v.real ** 2 + v.imag ** 2).sum()
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print('step wall %.2f ms' % ((time.perf_counter() - t0) / 3 * 1e3))
# host time only (no sync inside): how long does Python take to enqueue one step?
t0 = time.perf_counter()
step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms, then drain %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=20, max_name_column_width=60))
