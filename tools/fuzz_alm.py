"""Randomised check of the float32 alm2pix paths (f16-split MFMA and exact-f32 MFMA) against the
float64 VALU kernels, forward and backward, over random row / coefficient / pixel counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bayeslim_amd import ops

ntrial = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for trial in range(ntrial):
    rng = np.random.default_rng(trial)
    R = int(rng.choice([1, 3, 4, 31, 32, 33, 64, 65, 100, 128, 129, 200]))
    Nc = int(rng.choice([1, 5, 8, 31, 32, 33, 127, 128, 129, 500, 1000]))
    Npix = int(rng.choice([1, 2, 7, 30, 31, 32, 33, 34, 100, 1110, 1111, 4096, 5001, 5002]))
    amp = 10.0 ** rng.uniform(-6, 6)
    yamp = 10.0 ** rng.uniform(-4, 4)
    a = torch.as_tensor((rng.normal(size=(R, Nc)) + 1j * rng.normal(size=(R, Nc))) * amp * np.exp(-6 * rng.uniform(size=(R, 1))))
    Y = torch.as_tensor((rng.normal(size=(Nc, Npix)) + 1j * rng.normal(size=(Nc, Npix))) * yamp)
    g = torch.as_tensor(rng.normal(size=(R, Npix)))
    ref_in = a.cuda().requires_grad_(True)
    ref = ops.alm2pix(ref_in, Y.cuda())                                   # float64 VALU kernels
    (ref * g.cuda()).sum().backward()
    errs = []
    for split in (True, False):
        ops.ALM_SPLIT_F16 = split
        x = a.to(torch.complex64).cuda().requires_grad_(True)
        y = ops.alm2pix(x, Y.to(torch.complex64).cuda())
        (y * g.float().cuda()).sum().backward()
        ev = float((y.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300))
        eg = float((x.grad.to(torch.complex128) - ref_in.grad).abs().max() / ref_in.grad.abs().max().clamp_min(1e-300))
        errs += [ev, eg]
    ops.ALM_SPLIT_F16 = True
    worst = max(worst, *errs)
    flag = '' if max(errs) < 1e-5 else '   <-- FAIL'
    print('trial %2d: R %3d Nc %4d Npix %4d amp %.0e yamp %.0e | split fwd %.1e bwd %.1e | exact fwd %.1e bwd %.1e%s' % (
        trial, R, Nc, Npix, amp, yamp, *errs, flag))
print('worst %.2e' % worst)
