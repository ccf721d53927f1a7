#!/usr/bin/env python3
"""Fringe sum with SEVERAL antenna beam models (Nmp > 1, beam_model.py:303-327): antenna-factored matrix-core
kernels (one block per (group pair, model pair)) against the baseline-formulation vector-ALU kernels round 1 fell
back to.  HERA-128-like layout, all pairs, float32.   python tools/bench_beam_models.py [Nant] [Nmodels] [Nf] [P]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslim_amd import ops

Nant = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Nmod = int(sys.argv[2]) if len(sys.argv) > 2 else 2
Nf = int(sys.argv[3]) if len(sys.argv) > 3 else 64
P = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
Nt = 1
rng = np.random.default_rng(0)
ant = torch.as_tensor(np.c_[rng.uniform(-120, 120, (Nant, 2)), rng.normal(0, 0.5, Nant)])
pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
model = [a % Nmod for a in range(Nant)]
uniq = sorted({(model[a], model[b]) for a, b in pairs})
bl_mp = [uniq.index((model[a], model[b])) for a, b in pairs]
blvecs = torch.stack([ant[b] - ant[a] for a, b in pairs]).cuda()
cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P)
sdir = torch.as_tensor(np.stack([np.sqrt(1 - cz ** 2) * np.sin(az), np.sqrt(1 - cz ** 2) * np.cos(az), cz])[None]).cuda()
freqs = np.linspace(120e6, 180e6, Nf)
psky = torch.rand(Nt, len(uniq), 1, Nf, P, device='cuda', requires_grad=True)
print('%d antennas, %d beam models -> %d model pairs, %d baselines, %d channels, %d pixels' % (Nant, Nmod, len(uniq), len(pairs), Nf, P))
res = {}
for name, kw in (('matrix cores (blocks per model pair)', dict(antpos=ant.cuda(), bl_ants=pairs, mfma=True, mp_pairs=uniq)),
                 ('vector ALU (baseline formulation)', dict())):
    geom = ops.FringeGeometry(blvecs, sdir, freqs, bl_mp=bl_mp, Nmp=len(uniq), **kw)
    def step():
        psky.grad = None
        v = ops.fringe_sum(psky, geom)
        (v.real ** 2 + v.imag ** 2).sum().backward()
        return v
    v = step(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    res[name] = (float(np.median(ts)), v.detach())
    nb = len(geom.ant['blocks']) if geom.ant is not None else 0
    print('  %-40s %8.3f ms fwd+bwd   (%d blocks)' % (name, res[name][0], nb))
a, b = list(res.values())
print('  speed-up %.2fx; max |difference| / max |V| = %.2e' % (b[0] / a[0], float((a[1] - b[1]).abs().max() / b[1].abs().max())))
