"""Soak of the matrix-core fringe kernels with MANY blocks in flight (several blocks per CU, most of them dispatched while
others stream MFMAs): three runs of forward and backward on the same inputs must be bit-identical and agree with the
vector-ALU kernels.  Round 5: this is the regime in which packed-f32 arithmetic in one block went wrong beside another block's
MFMA stream (csrc/fringe_mfma.hip, keep_scalar); tools/soak_fullsize.py's cases have too few blocks to reach it.
usage: python tools/soak_coresident.py [P] [Nf] [Nt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayeslim_amd import ops, utils

P = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Nt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T64 = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)
worst, bad = 0.0, 0
cases = [('rand24', 24), ('rand40', 40), ('rand60', 60), ('rand90', 90), ('rand128', 128), ('rand150', 150),
         ('hex37', -4), ('hex61', -5), ('hex91', -6), ('hex127', -7), ('hex127+1', -7.5)]
for name, n in cases:
    rng = np.random.default_rng(abs(int(n * 2)))
    if n < 0:
        ant = np.asarray(utils._make_hex(int(-n), D=14.6)[1])
        if n != int(n):
            ant = np.vstack([ant, [[250.0, 0.0, 0.0]]])
    else:
        ant = rng.normal(0, 150.0, (n, 3)) * [1, 1, 0.01]
    Nant = len(ant)
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    antp = T64(ant).cuda()
    blvecs = antp[torch.as_tensor([b for _, b in pairs], device='cuda')] - antp[torch.as_tensor([a for a, _ in pairs], device='cuda')]
    cz, az = rng.uniform(0, 1, (Nt, P)), rng.uniform(0, 2 * np.pi, (Nt, P))
    sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz], axis=1)).cuda()
    freqs = torch.linspace(140e6, 160e6, Nf, dtype=torch.float64)
    gm = ops.FringeGeometry(blvecs, sdir, freqs, antpos=antp, bl_ants=pairs, mfma=True)
    gv = ops.FringeGeometry(blvecs, sdir, freqs, mfma=False)
    gen = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(Nt, 1, 1, Nf, P, device='cuda', generator=gen)
    g = torch.complex(torch.randn(1, len(pairs), Nt, Nf, device='cuda', generator=gen), torch.randn(1, len(pairs), Nt, Nf, device='cuda', generator=gen))
    rv, rg = ops.fringe_sum(x, gv), ops.fringe_adjoint(g, gv)
    fw = [ops.fringe_sum(x, gm).clone() for _ in range(3)]
    bw = [ops.fringe_adjoint(g, gm).clone() for _ in range(3)]
    same = all(torch.equal(fw[0], v) for v in fw[1:]) and all(torch.equal(bw[0], v) for v in bw[1:])
    ev = float((fw[0] - rv).abs().max() / rv.abs().max())
    eg = float((bw[0] - rg).abs().max() / rg.abs().max())
    worst = max(worst, ev, eg)
    ok = same and ev < 1e-5 and eg < 1e-4
    bad += not ok
    print('%-9s %3d antennas %5d baselines, %d x %d x %d: %s, vs vector-ALU vis %.1e grad %.1e  pair form %s mirror %s%s' % (
        name, Nant, len(pairs), Nt, Nf, P, 'bit-identical runs' if same else 'RUNS DIFFER', ev, eg,
        gm.ant.get('pair_blocks') or '-', gm.ant.get('mirror_groups') or '-', '' if ok else '   <-- FAIL'), flush=True)
    del gm, gv, x, g, rv, rg, fw, bw
    torch.cuda.empty_cache()
print('worst %.2e, failures %d' % (worst, bad))
sys.exit(1 if bad else 0)
