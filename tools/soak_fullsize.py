"""Soak: repeated forward + backward of the matrix-core fringe kernels at full pixel counts, real and complex psky,
checking run-to-run bit identity and agreement with the vector-ALU kernels every iteration (a timing-dependent defect
shows up as non-identical runs).  usage: python tools/soak_fullsize.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayeslim_amd import ops

niter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T64 = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)
worst = 0.0
# (round 5) the first two cases are arrays WITH mirror pairs: the benchmark's HERA-128 (127-antenna hexagon + outrigger) and HERA-37
for Nant, P, Nf, group in ((-128, 98304, 32, 128), (-37, 24576, 64, 128), (128, 98304, 32, 128), (512, 393216, 2, 128),
                           (128, 393216, 4, 32), (37, 24576, 64, 128), (96, 196608, 4, 128)):
    rng = np.random.default_rng(abs(Nant))
    if Nant < 0:
        from bayeslim_amd import utils
        ant = utils._make_hex({128: 7, 37: 4}[-Nant], D=14.6)[1]
        if Nant == -128:
            ant = np.vstack([ant, [[250.0, 0.0, 0.0]]])
        Nant = len(ant)
    else:
        ant = rng.normal(0, 200.0, (Nant, 3)); ant[:, 2] = 0.0
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    antp = T64(ant).cuda()
    blvecs = antp[torch.as_tensor([b for _, b in pairs], device='cuda')] - antp[torch.as_tensor([a for a, _ in pairs], device='cuda')]
    cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P); sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz])[None]).cuda()
    freqs = torch.linspace(150e6, 151e6, Nf, dtype=torch.float64)
    gm = ops.FringeGeometry(blvecs, sdir, freqs, antpos=antp, bl_ants=pairs, mfma=True, group=group)
    gv = ops.FringeGeometry(blvecs, sdir, freqs, mfma=False)
    gen = torch.Generator(device='cuda').manual_seed(1)
    xc = torch.complex(torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen), torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen))
    for cplx in (True, False):
        xin = xc if cplx else xc.real.contiguous()
        ref = first = None
        for it in range(niter + 1):
            g_ = gv if it == 0 else gm
            xx = xin.clone().requires_grad_(True)
            vv = ops.fringe_sum(xx, g_)
            if it == 0:
                G = torch.complex(torch.randn(vv.shape, device='cuda', generator=gen), torch.randn(vv.shape, device='cuda', generator=gen))
            (vv * G.conj()).real.sum().backward()
            cur = (vv.detach(), xx.grad.detach())
            if it == 0:
                ref = cur
            elif first is None:
                first = cur
                ev = float((cur[0] - ref[0]).abs().max() / ref[0].abs().max())
                eg = float((cur[1] - ref[1]).abs().max() / ref[1].abs().max())
                worst = max(worst, ev, eg)
                assert ev < 1e-5 and eg < 1e-5, (Nant, P, cplx, ev, eg)
            else:
                assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]), ('run-to-run', Nant, P, cplx, it)
        print('Nant %3d P %6d Nf %2d group %3d cplx %d: %d identical runs, vs vector-ALU vis %.1e grad %.1e'
              % (Nant, P, Nf, group, cplx, niter, ev, eg), flush=True)
print('worst %.2e' % worst)
