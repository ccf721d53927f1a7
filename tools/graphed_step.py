"""hipGraph capture of a whole RIME step (forward + backward) with torch.cuda.make_graphed_callables:
the drop-in modules only launch kernels on the current stream and allocate through torch, so the
launch-bound small workloads (C2: ~120 launches per step) replay as two graphs.
python tools/graphed_step.py [workload] [nt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
nt = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[wl]['nt']
dev = torch.device('cuda', 0)
inp = bench.build_inputs(wl, nt)
bls = bench.all_baselines(inp)
rime, params, attach, _ = bench.build_model(inp, dev, bls)
sky, beam = rime.sky, rime.beam


def fwd(*ps):
    # the reference's parameter protocol: params are re-set before every forward
    models = list(sky.models.values()) if hasattr(sky, 'models') else [sky]
    for m, q in zip(models + [beam], ps):
        if hasattr(m, 'params'):
            delattr(m, 'params')
        m.params = q
    v = rime().data
    return torch.view_as_real(v)


def eager_step():
    for p in params:
        p.grad = None
    out = fwd(*params)
    (out ** 2).sum().backward()


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('params:', [tuple(p.shape) for p in params])
t_eager = timeit(eager_step)
g_ref = [p.grad.clone() for p in params]
sample = tuple(p.detach().clone().requires_grad_(True) for p in params)
gfwd = torch.cuda.make_graphed_callables(fwd, sample)


def graph_step():
    for p in params:
        p.grad = None
    out = gfwd(*params)
    (out ** 2).sum().backward()


t_graph = timeit(graph_step)
err = max(float((p.grad - g).abs().max() / g.abs().max()) for p, g in zip(params, g_ref))
print('%s nt=%d: eager %.3f ms/step, graphed %.3f ms/step (x%.2f); max grad difference %.1e' % (
    wl, nt, t_eager, t_graph, t_eager / t_graph, err))
