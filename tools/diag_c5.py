import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bayeslim_amd import ops
T64 = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)
def run(Nant, P, Nf=4, seed=12):
    rng = np.random.default_rng(seed)
    ant = rng.normal(0, 300.0, (Nant, 3)); ant[:, 2] = 0.0
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    antp = T64(ant).cuda()
    i1 = torch.as_tensor([a for a, _ in pairs], device='cuda'); i2 = torch.as_tensor([b for _, b in pairs], device='cuda')
    blvecs = antp[i2] - antp[i1]
    cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P); sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz])[None]).cuda()
    freqs = torch.linspace(120e6, 121e6, Nf, dtype=torch.float64)
    gen = torch.Generator(device='cuda').manual_seed(2)
    env = torch.exp(-9.0 * torch.rand(1, 1, 1, Nf, P, device='cuda', generator=gen))
    x1 = torch.complex(torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen), torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen)) * env
    res = {}
    for name, kw in (('mfma', dict(antpos=antp, bl_ants=pairs)), ('valu', dict(mfma=False))):
        geom = ops.FringeGeometry(blvecs, sdir, freqs, **kw)
        x = x1.clone().requires_grad_(True)
        v = ops.fringe_sum(x, geom)
        g2 = torch.Generator(device='cuda').manual_seed(5)
        G = torch.complex(torch.randn(v.shape, device='cuda', generator=g2), torch.randn(v.shape, device='cuda', generator=g2))
        lhs = (v.detach() * G.conj()).real.double().sum()
        (v * G.conj()).real.sum().backward()
        rhs = (x.grad.conj() * x1).real.double().sum()
        res[name] = (v.detach(), x.grad.detach())
        print(Nant, P, name, 'adjoint rel err %.2e' % (abs(float(lhs - rhs)) / abs(float(lhs))), flush=True)
    dv = (res['mfma'][0] - res['valu'][0]).abs().max() / res['valu'][0].abs().max()
    dg = (res['mfma'][1] - res['valu'][1]).abs().max() / res['valu'][1].abs().max()
    print('   mfma vs valu: vis %.2e grad %.2e' % (float(dv), float(dg)))
    # which psky plane / region of gradient differs?
    d = (res['mfma'][1] - res['valu'][1])[0, 0, 0]
    print('   grad diff re %.2e im %.2e' % (float(d.real.abs().max()), float(d.imag.abs().max())), 'ref max', float(res['valu'][1].abs().max()))
for Nant, P in ((512, 393216), (512, 32768), (256, 393216), (300, 65536)):
    run(Nant, P)

def pattern(Nant, P, Nf=4, seed=12):
    rng = np.random.default_rng(seed)
    ant = rng.normal(0, 300.0, (Nant, 3)); ant[:, 2] = 0.0
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    antp = T64(ant).cuda()
    i1 = torch.as_tensor([a for a, _ in pairs], device='cuda'); i2 = torch.as_tensor([b for _, b in pairs], device='cuda')
    blvecs = antp[i2] - antp[i1]
    cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P); sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz])[None]).cuda()
    freqs = torch.linspace(120e6, 121e6, Nf, dtype=torch.float64)
    gen = torch.Generator(device='cuda').manual_seed(2)
    x1 = torch.complex(torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen), torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen))
    gr = {}
    for name, kw in (('mfma', dict(antpos=antp, bl_ants=pairs)), ('valu', dict(mfma=False))):
        geom = ops.FringeGeometry(blvecs, sdir, freqs, **kw)
        x = x1.clone().requires_grad_(True)
        v = ops.fringe_sum(x, geom)
        g2 = torch.Generator(device='cuda').manual_seed(5)
        G = torch.complex(torch.randn(v.shape, device='cuda', generator=g2), torch.randn(v.shape, device='cuda', generator=g2))
        (v * G.conj()).real.sum().backward()
        gr[name] = x.grad.detach()[0, 0, 0]
    d = (gr['mfma'] - gr['valu']).imag.abs()
    ref = gr['valu'].abs().max()
    bad = d > 1e-3 * ref
    print('Nant', Nant, 'P', P, 'bad entries', int(bad.sum()), 'of', bad.numel(), 'per channel', bad.sum(1).tolist())
    idx = torch.nonzero(bad[0])[:, 0]
    if len(idx):
        print('  channel 0 bad pixel range', int(idx.min()), int(idx.max()), 'first', idx[:12].tolist(), 'mod 32 set', sorted(set((idx % 32).tolist()))[:40])
        ratio = (gr['mfma'].imag[0][idx] / gr['valu'].imag[0][idx])[:8]
        print('  ratio mfma/valu imag', ratio.tolist())
        print('  tile ids', sorted(set((idx // 32).tolist()))[:20], '... count', len(set((idx // 32).tolist())))

print('--- pattern')
pattern(256, 393216)
pattern(256, 131072)
pattern(128, 393216)
