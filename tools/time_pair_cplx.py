import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from bayeslim_amd import ops, utils
ant = np.vstack([utils._make_hex(7, D=14.6)[1], [[250.0, 0.0, 0.0]]])
n = len(ant); pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
rng = np.random.default_rng(0)
Nt, Nf, P = 2, 64, 98304
blvecs = torch.as_tensor(np.stack([ant[b] - ant[a] for a, b in pairs])).cuda()
freqs = torch.linspace(120e6, 180e6, Nf, dtype=torch.float64)
s = rng.normal(size=(Nt, 3, P)); s /= np.linalg.norm(s, axis=1, keepdims=True); s[:, 2] = np.abs(s[:, 2])
sdir = torch.as_tensor(s).cuda()
x = torch.complex(torch.randn(Nt, 1, 1, Nf, P, device='cuda'), torch.randn(Nt, 1, 1, Nf, P, device='cuda'))
g = torch.complex(torch.randn(1, len(pairs), Nt, Nf, device='cuda'), torch.randn(1, len(pairs), Nt, Nf, device='cuda'))
for on in (True, False):
    ops.PAIR_CPLX = on
    geom = ops.FringeGeometry(blvecs, sdir, freqs, antpos=torch.as_tensor(ant).cuda(), bl_ants=pairs, mfma=True)
    xx = x.clone().requires_grad_(True)
    for it in range(3):
        v = ops.fringe_sum(xx, geom); (v * g.conj()).real.sum().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(5):
        v = ops.fringe_sum(xx, geom)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for it in range(5):
        v = ops.fringe_sum(xx, geom); (v * g.conj()).real.sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('complex psky, HERA-128 hex + outrigger, %d x %d x %d: PAIR_CPLX=%d  forward %.2f ms  forward + backward %.2f ms' % (Nt, Nf, P, on, (t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3))
