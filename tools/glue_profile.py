"""Which torch ops (not HIP-library kernels) does a bench step launch, with shapes and the python line that issued them?
python tools/glue_profile.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bayeslim_amd import ops
from torch.profiler import profile, ProfilerActivity

wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
if len(sys.argv) > 2:                                   # second argument: number of channels (c5 64 = one rank's share)
    bench.WORKLOADS[wl] = dict(bench.WORKLOADS[wl], Nf=int(sys.argv[2]))
dev = torch.device('cuda', 0)
inp = bench.build_inputs(wl, bench.WORKLOADS[wl]['nt'])
rime, params, attach, _ = bench.build_model(inp, dev, bench.all_baselines(inp))


def step():
    for p in params:
        p.grad = None
    attach()
    ops.chisq(rime().data).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
rows = []
for e in ka:
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(e, 'self_cuda_time_total', 0)
    if e.key.startswith('aten::') and dt > 0:
        stack = [s_ for s_ in (e.stack or []) if 'bayeslim_amd' in s_ or 'bench.py' in s_]
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:60], (stack[0] if stack else (e.stack[0] if e.stack else '?'))[-100:]))
rows.sort(reverse=True)
print('%d aten op groups with device time, %.1f us in total' % (len(rows), sum(r[0] for r in rows)))
for r in rows[:60]:
    print('%8.1f us x%d %-20s %-60s %s' % r)
