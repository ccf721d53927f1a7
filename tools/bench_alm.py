#!/usr/bin/env python3
"""Kernel-level timing of alm2pix fwd/bwd (C3-like: R=128 rows, lmax=128 -> 8385 coeffs, nside-64 pixels)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslim_amd import ops

def run(R, Nc, Npix, reps=5):
    a = torch.randn(R, Nc, dtype=torch.complex64, device='cuda', requires_grad=True)
    Y = torch.randn(Nc, Npix, dtype=torch.complex64, device='cuda')
    flop = 4.0 * R * Nc * Npix
    byts = 8.0 * Nc * Npix + 4.0 * R * Npix
    for name in ('fwd', 'bwd'):
        ts = []
        for _ in range(reps + 1):
            out = ops.alm2pix(a, Y)
            g = torch.ones_like(out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == 'fwd':
                out = ops.alm2pix(a, Y)
            else:
                out.backward(g)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1)); a.grad = None
        ms = float(np.median(ts[1:]))
        print('  %s R=%d Nc=%d Npix=%d: %.3f ms  %.1f TFLOP/s  %.0f GB/s (Ylm stream)' % (name, R, Nc, Npix, ms, flop / ms * 1e-9, byts / ms * 1e-6), flush=True)

print(ops._lib.version())
for packed in (False, True):
    ops.ALM_PACKED = packed
    print('packed Ylm (cached pre-split f16 copies in fragment order):' if packed else 'Ylm split inside the kernels:', flush=True)
    run(128, 8385, 49152)
    run(128, 2145, 196608)
    run(4, 8385, 49152)
    if len(sys.argv) > 1:
        run(128, 8385, 44279)            # the C3 bench's own pixel count class (odd)
