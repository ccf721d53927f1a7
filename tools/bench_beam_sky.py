"""Time the fused psky builder (rime_beam_sky_fwd / _bwd + the interpolation adjoint) in isolation on the arguments
the bench model hands it (real stencil / cut / pos arrays of the workload: the locality of the beam-node gathers is what
these kernels live on).  python tools/bench_beam_sky.py [c4] [nt]"""
import os, sys
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

calls = []
orig = ops.beam_sky_product


def spy(bmap, sky, stencil, cut, pos, Nt, Ps):
    calls.append((bmap.detach().clone(), sky.detach().clone(), stencil, cut, pos, int(Nt), int(Ps)))
    return orig(bmap, sky, stencil, cut, pos, Nt, Ps)


ops.beam_sky_product = spy
import bayeslim_amd.rime_model as rm
import bayeslim_amd.beam_model as bm
for mod in (rm, bm):
    if hasattr(mod, 'beam_sky_product'):
        mod.beam_sky_product = spy
attach()
with torch.no_grad():
    rime()
torch.cuda.synchronize()
ops.beam_sky_product = orig
print('%d builder calls per forward' % len(calls))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


for (bmap, sky, st, cut, pos, Nt, Ps) in calls:
    R, Npb = bmap.shape
    Npix = sky.shape[1]
    Q = Nt * Ps
    esz = bmap.element_size()
    print('R=%d Npb=%d Npix=%d Nt=%d Ps=%d Nnn=%d %s' % (R, Npb, Npix, Nt, Ps, st.Nnn, bmap.dtype))
    b = bmap.clone().requires_grad_(True)
    k = sky.clone().requires_grad_(True)
    out = orig(b, k, st, cut, pos, Nt, Ps)
    g = torch.randn_like(out)
    tf = timeit(lambda: orig(bmap, sky, st, cut, pos, Nt, Ps))

    def fb():
        b.grad = None
        k.grad = None
        o = orig(b, k, st, cut, pos, Nt, Ps)
        o.backward(g)
    tfb = timeit(fb)
    alg_f = esz * R * (2.0 * Q + Npb) + 8.0 * st.Nnn * Q          # psky write + sky gather + map + stencil
    alg_b = esz * R * (3.0 * Q + 2.0 * Npb + Npix) + 8.0 * st.Nnn * Q   # gpsky twice + sky gather + T1 ... (lower bound)
    print('  forward %.3f ms (%.0f GB/s of %.2f GB algorithmic)   backward %.3f ms (%.0f GB/s of %.2f GB)' % (
        tf, alg_f / tf / 1e6, alg_f / 1e9, tfb - tf, alg_b / (tfb - tf) / 1e6, alg_b / 1e9))
