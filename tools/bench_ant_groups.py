"""Antenna-factored (matrix-core) vs baseline-formulation fringe kernels for arrays of more than
128 antennas: python tools/bench_ant_groups.py [Nant] [Nf] [P] [Nt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayeslim_amd import ops

Nant = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
P = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
Nt = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(0)
ant = rng.normal(0, 150.0, (Nant, 3)); ant[:, 2] *= 0.02
pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
blvecs = torch.as_tensor(np.stack([ant[b] - ant[a] for a, b in pairs])).cuda()
cz = rng.uniform(0, 1, (Nt, P)); az = rng.uniform(0, 2 * np.pi, (Nt, P)); sz = np.sqrt(1 - cz ** 2)
sdir = torch.as_tensor(np.stack([sz * np.sin(az), sz * np.cos(az), cz], axis=1)).cuda()
freqs = torch.linspace(120e6, 180e6, Nf, dtype=torch.float64)
psky = (torch.randn(Nt, 1, 1, Nf, P, device='cuda') * 1e-3).requires_grad_(True)
gm = ops.FringeGeometry(blvecs, sdir, freqs, antpos=torch.as_tensor(ant).cuda(), bl_ants=pairs, mfma=True)
gv = ops.FringeGeometry(blvecs, sdir, freqs)
print('Nant %d Nbl %d Nf %d P %d Nt %d: %d blocks, %.3g executed MFMA flop/pass' % (
    Nant, len(pairs), Nf, P, Nt, len(gm.ant['blocks']), gm.ant['mfma_flops_fwd']))


def timeit(geom, reps=3):
    out = []
    for r in range(reps + 1):
        psky.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vis = ops.fringe_sum(psky, geom)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (vis.real ** 2 + vis.imag ** 2).sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if r:
            out.append((t1 - t0, t2 - t1))
    return min(o[0] for o in out) * 1e3, min(o[1] for o in out) * 1e3, vis.detach(), psky.grad.clone()


fm, bm, vm, gmg = timeit(gm)
fv, bv, vv, gvg = timeit(gv)
print('matrix cores : fwd %.2f ms  bwd %.2f ms  (%.0f TFLOP/s executed fwd)' % (fm, bm, gm.ant['mfma_flops_fwd'] / fm / 1e9))
print('vector ALU   : fwd %.2f ms  bwd %.2f ms' % (fv, bv))
print('agreement    : vis %.2e  grad %.2e (of max)' % (float((vm - vv).abs().max() / vv.abs().max()),
                                                      float((gmg - gvg).abs().max() / gvg.abs().max())))
