"""Where is the crossover between the vector-ALU fringe kernels and the antenna-factored matrix-core kernels for small
arrays (all pairs of Nant antennas)?  forward + backward, float32.  python tools/bench_small_arrays.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslim_amd import ops


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def run(Nant, Nt, Nf, P):
    dev = 'cuda'
    rng = np.random.default_rng(Nant)
    antpos = torch.as_tensor(rng.normal(0, 60.0, (Nant, 3)) * [1, 1, 0.01], device=dev)
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    blv = torch.stack([antpos[j] - antpos[i] for i, j in pairs])
    Ps = ops.pad_to_tile(P)
    cz = torch.rand(Nt, Ps, device=dev, dtype=torch.float64)
    az = torch.rand(Nt, Ps, device=dev, dtype=torch.float64) * 2 * np.pi
    sz = torch.sqrt(1 - cz ** 2)
    sdir = torch.stack([sz * torch.sin(az), sz * torch.cos(az), cz], dim=1)
    freqs = torch.linspace(120e6, 180e6, Nf, dtype=torch.float64)
    out = []
    psky = torch.rand(Nt, 1, 1, Nf, Ps, device=dev).requires_grad_(True)
    for mode in (False, True):
        geom = ops.FringeGeometry(blv, sdir, freqs, antpos=antpos, bl_ants=pairs, mfma=mode)
        assert (geom.ant is not None) == mode
        vis = ops.fringe_sum(psky, geom)
        g = torch.randn_like(vis)
        tf = timeit(lambda: ops.fringe_sum(psky.detach(), geom))

        def fb():
            psky.grad = None
            ops.fringe_sum(psky, geom).backward(g)
        out.append((tf, timeit(fb) - tf, vis.detach()))
    err = float((out[0][2] - out[1][2]).abs().max() / out[0][2].abs().max())
    print('Nant %3d (%4d bl) Nt %d Nf %d P %6d:  vector-ALU fwd %.3f bwd %.3f   matrix-core fwd %.3f bwd %.3f ms   (sum %.3f vs %.3f; max diff %.1e)' % (
        Nant, len(pairs), Nt, Nf, P, out[0][0], out[0][1], out[1][0], out[1][1], out[0][0] + out[0][1], out[1][0] + out[1][1], err))


def run_fullpol(Nant, Nt, Nf, P):
    """4-pol complex psky (the full-polarisation layout) on the matrix-core kernels: forward and backward times"""
    dev = 'cuda'
    rng = np.random.default_rng(Nant)
    antpos = torch.as_tensor(rng.normal(0, 60.0, (Nant, 3)) * [1, 1, 0.01], device=dev)
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    blv = torch.stack([antpos[j] - antpos[i] for i, j in pairs])
    Ps = ops.pad_to_tile(P)
    cz = torch.rand(Nt, Ps, device=dev, dtype=torch.float64)
    az = torch.rand(Nt, Ps, device=dev, dtype=torch.float64) * 2 * np.pi
    sz = torch.sqrt(1 - cz ** 2)
    sdir = torch.stack([sz * torch.sin(az), sz * torch.cos(az), cz], dim=1)
    freqs = torch.linspace(120e6, 180e6, Nf, dtype=torch.float64)
    geom = ops.FringeGeometry(blv, sdir, freqs, antpos=antpos, bl_ants=pairs, mfma=True)
    psky = torch.complex(torch.randn(Nt, 1, 4, Nf, Ps, device=dev), torch.randn(Nt, 1, 4, Nf, Ps, device=dev)).requires_grad_(True)
    vis = ops.fringe_sum(psky, geom)
    g = torch.randn_like(vis)
    tf = timeit(lambda: ops.fringe_sum(psky.detach(), geom))

    def fb():
        psky.grad = None
        ops.fringe_sum(psky, geom).backward(g)
    print('4-pol complex psky, Nant %3d (%4d bl) Nt %d Nf %d P %6d:  forward %.3f ms  backward %.3f ms' % (
        Nant, len(pairs), Nt, Nf, P, tf, timeit(fb) - tf))


if len(sys.argv) > 1 and sys.argv[1] == 'fullpol':
    for Nant in (37, 64, 96, 128):
        run_fullpol(Nant, 2, 64, 49152)
    sys.exit(0)
for Nant in (7, 10, 13, 16, 19, 24, 28, 32):
    run(Nant, 8, 64, 6144)
for Nant in (10, 16, 24):
    run(Nant, 4, 128, 49152)
