"""Randomised cross-checks of the GPU ops against each other (development tool): the antenna-factored
(matrix-core) fringe kernels vs the baseline-formulation kernels, forward and backward, over random
antenna counts, pair subsets and orientations, polarisation layouts, channel / time / pixel counts, numbers of
beam models (blocks per model pair) and block group sizes (128 / 64 / 32: every kernel shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayeslim_amd import ops

ntrial = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
worst = [0.0, 0.0]
for trial in range(ntrial):
    rng = np.random.default_rng(seed0 + trial)
    Nant = int(rng.choice([3, 17, 33, 37, 40, 45, 48, 64, 65, 97, 128, 129, 160, 257]))
    Nt, Nf = int(rng.integers(1, 4)), int(rng.integers(1, 40))
    P = int(rng.choice([1, 30, 64, 100, 700, 3000, 9000, 20000]))
    Npp, cplx = [(1, False), (2, False), (1, True), (4, True), (4, False)][int(rng.integers(0, 5))]
    frac = float(rng.choice([1.0, 0.6, 0.15]))
    ant = rng.normal(0, 70.0, (Nant, 3)); ant[:, 2] *= 0.03
    # round 5: half of the trials on arrays WITH point symmetry (mirror pairs: conjugate phasors) -- a random number of
    # mirrored pairs about a centre off the origin, the rest without a partner, shuffled
    sym = 'none'
    if rng.random() < 0.5 and Nant >= 4:
        npair = int(rng.integers(2, Nant // 2 + 1))
        h = ant[:npair]
        ant = np.vstack([h, -h, ant[2 * npair:]])[rng.permutation(Nant)] + rng.normal(0, 30.0, 3)
        sym = '%d pairs' % npair
    pairs = [(i, j) for i in range(Nant) for j in range(i, Nant) if rng.random() < frac]
    if not pairs:
        pairs = [(0, Nant - 1)]
    pairs = [p if rng.random() < 0.5 else p[::-1] for p in pairs]
    pairs = [pairs[k] for k in rng.permutation(len(pairs))]
    blv = torch.as_tensor(np.stack([ant[b] - ant[a] for a, b in pairs])).cuda()
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    cz, az = rng.uniform(0, 1, (Nt, P)), rng.uniform(0, 2 * np.pi, (Nt, P))
    sz = np.sqrt(1 - cz ** 2)
    sdir[:, :, :P] = torch.as_tensor(np.stack([sz * np.sin(az), sz * np.cos(az), cz], axis=1))
    uniform = rng.random() < 0.7
    freqs = torch.as_tensor(np.linspace(110e6, 190e6, Nf) if uniform else np.sort(rng.uniform(100e6, 200e6, Nf)))
    conj = bool(rng.random() < 0.5)
    orient = rng.choice(['mixed', 'up', 'down'])
    if orient != 'mixed':
        pairs = [(min(p), max(p)) if orient == 'up' else (max(p), min(p)) for p in pairs]
        blv = torch.as_tensor(np.stack([ant[b] - ant[a] for a, b in pairs])).cuda()
    Nmod = int(rng.choice([1, 1, 2, 3]))
    model = [int(rng.integers(0, Nmod)) for _ in range(Nant)]
    uniq = sorted({(model[a], model[b]) for a, b in pairs})
    bl_mp = [uniq.index((model[a], model[b])) for a, b in pairs]
    Nmp = len(uniq)
    group = int(rng.choice([128, 128, 64, 32]))
    gm = ops.FringeGeometry(blv, sdir.cuda(), freqs, conj=conj, antpos=torch.as_tensor(ant).cuda(), bl_ants=pairs, mfma=True,
                            bl_mp=bl_mp, Nmp=Nmp, mp_pairs=uniq, group=group)
    gv = ops.FringeGeometry(blv, sdir.cuda(), freqs, conj=conj, mfma=False, bl_mp=bl_mp, Nmp=Nmp)
    assert gm.ant is not None, 'antenna path refused: Nant %d' % Nant
    x = rng.normal(size=(Nt, Nmp, Npp, Nf, Ps)) * np.exp(-8 * rng.uniform(size=(Nt, Nmp, Npp, Nf, Ps)))
    if cplx:
        x = x + 1j * rng.normal(size=x.shape) * np.exp(-8 * rng.uniform(size=x.shape))
    x[..., P:] = 0
    x = torch.as_tensor(x).to(torch.complex64 if cplx else torch.float32).cuda()
    res = []
    for geom in (gm, gv):
        xx = x.clone().requires_grad_(True)
        v = ops.fringe_sum(xx, geom)
        g = torch.as_tensor(np.random.default_rng(1).normal(size=tuple(v.shape)) + 1j * np.random.default_rng(2).normal(size=tuple(v.shape))).to(torch.complex64).cuda()
        (v * g.conj()).real.sum().backward()
        res.append((v.detach(), xx.grad.detach()))
    ev = float((res[0][0] - res[1][0]).abs().max() / res[1][0].abs().max().clamp_min(1e-30))
    eg = float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max().clamp_min(1e-30))
    worst = [max(worst[0], ev), max(worst[1], eg)]
    flag = '' if (ev < 1e-5 and eg < 1e-4) else '   <-- FAIL'
    mg = gm.ant.get('mirror_groups', [])
    pb = gm.ant.get('pair_blocks', [])                        # conjugate-pair form: (pairs, rows, hub) per block (real psky passes)
    print('trial %3d: Nant %3d Nbl %5d Nt %d Nf %2d P %4d Npp %d cplx %d conj %d uniform %d orient %-5s models %d (%d pairs) group %3d blocks %3d  sym %-9s mirror groups %-14s pair form %-16s vis %.1e grad %.1e%s' % (
        trial, Nant, len(pairs), Nt, Nf, P, Npp, cplx, conj, uniform, orient, Nmod, Nmp, group, len(gm.ant['blocks']), sym,
        '+'.join('%d/%d' % g for g in mg) or '-', '+'.join('%d/%d/%d' % g for g in pb) or '-', ev, eg, flag), flush=True)
print('worst: vis %.2e grad %.2e' % tuple(worst))
