"""Debug: determinism and correctness of the conjugate-pair backward / forward at the headline size against the mirror-pair kernels.
usage: python tools/debug_pair_bwd.py [P] [Nf] [Nt]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslim_amd import ops, utils

P = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Nt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ant = np.vstack([utils._make_hex(7, D=14.6)[1], [[250.0, 0.0, 0.0]]])
if os.environ.get('DEBUG_NO_OUTRIGGER'):
    ant = ant[:-1]
n = len(ant)
pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
rng = np.random.default_rng(0)
blvecs = torch.as_tensor(np.stack([ant[b] - ant[a] for a, b in pairs])).cuda()
freqs = torch.linspace(120e6, 180e6, Nf, dtype=torch.float64)
Ps = ops.pad_to_tile(P)
s = rng.normal(size=(Nt, 3, Ps)); s /= np.linalg.norm(s, axis=1, keepdims=True); s[:, 2] = np.abs(s[:, 2])
sdir = torch.as_tensor(s).cuda()
g = torch.as_tensor(rng.normal(size=(1, len(pairs), Nt, Nf)) + 1j * rng.normal(size=(1, len(pairs), Nt, Nf))).to(torch.complex64).cuda()
psky = torch.as_tensor(rng.normal(size=(Nt, 1, 1, Nf, Ps))).float().cuda()
out = {}
for pair in (True, False):
    ops.PAIR = pair
    geom = ops.FringeGeometry(blvecs, sdir, freqs, antpos=torch.as_tensor(ant).cuda(), bl_ants=pairs, mfma=True)
    print('pair', pair, geom.ant.get('pair_blocks'), geom.ant.get('mirror_groups'))
    runs = [ops.fringe_adjoint(g, geom).clone() for _ in range(3)]
    vis = [ops.fringe_sum(psky, geom).clone() for _ in range(3)]
    torch.cuda.synchronize()
    for k in (1, 2):
        d = (runs[k] - runs[0]).abs()
        print('  bwd run', k, 'vs 0: max diff %.3e' % float(d.max()), 'differing px', int((d > 0).sum()))
        if float(d.max()) > 0:
            idx = torch.nonzero(d > 0)
            print('   first differing (t, mp, pp, f, p):', idx[:5].tolist(), ' px tiles:', sorted({int(i[4]) // 32 for i in idx})[:20])
        dv = (vis[k] - vis[0]).abs()
        print('  fwd run', k, 'vs 0: max diff %.3e' % float(dv.max()))
    out[pair] = (runs[0], vis[0])
d = (out[True][0] - out[False][0]).abs()
print('bwd pair vs mirror: max %.3e / max|ref| %.3e' % (float(d.max()), float(out[False][0].abs().max())))
rel = d / out[False][0].abs().max()
bad = torch.nonzero(rel > 1e-4)
print('  entries above 1e-4:', len(bad), bad[:8].tolist())
if len(bad):
    S = Ps // 8192
    blk = sorted({(int(i[0]) * S + int(i[4]) // 8192) * Nf + int(i[3]) for i in bad.cpu()})
    print('  wrong blocks (blockIdx):', len(blk), 'of', Nt * S * Nf, blk[:40], '...', blk[-10:])
    relb = rel[:, 0, 0].reshape(Nt, Nf, S, 8192).amax(-1)         # (t, f, split)
    print('  per-block max rel err, t=0:', relb[0].max().item(), ' t=1 min/max:', relb[1].min().item(), relb[1].max().item())
    print('  px tiles', sorted({int(i[4]) // 32 for i in bad})[:30], ' channels', sorted({int(i[3]) for i in bad}), ' times', sorted({int(i[0]) for i in bad}))
dv = (out[True][1] - out[False][1]).abs()
print('fwd pair vs mirror: max %.3e / max|ref| %.3e' % (float(dv.max()), float(out[False][1].abs().max())))
