#!/usr/bin/env python3
"""Kernel-level timing of the fused fringe sum (fwd + bwd) at benchmark shapes.  Development
tool: prints ms, elements/s and the SURVEY.md section 8(d) flop rate (10 flop / element / pass)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslim_amd import ops  # noqa: E402


def run(Nbl, Nt, Nf, P, Npp=1, cplx=False, blen=150.0, reps=5, dtype=torch.float32):
    dev = 'cuda'
    rng = np.random.default_rng(0)
    blvecs = torch.as_tensor(rng.normal(0, blen, (Nbl, 3)), device=dev)
    Ps = ops.pad_to_tile(P)
    cz = torch.rand(Nt, Ps, device=dev, dtype=torch.float64)
    az = torch.rand(Nt, Ps, device=dev, dtype=torch.float64) * 2 * np.pi
    sz = torch.sqrt(1 - cz ** 2)
    sdir = torch.stack([sz * torch.sin(az), sz * torch.cos(az), cz], dim=1)
    freqs = torch.linspace(120e6, 180e6, Nf, dtype=torch.float64)
    geom = ops.FringeGeometry(blvecs, sdir, freqs)
    psky = torch.randn(Nt, 1, Npp, Nf, Ps, device=dev, dtype=dtype)
    if cplx:
        psky = torch.complex(psky, torch.randn_like(psky))
    psky.requires_grad_(True)
    E = Nbl * Nf * P * Nt
    flop = E * (6 + (8 if cplx else 4) * Npp)
    res = {}
    for name in ('fwd', 'bwd'):
        ts = []
        for _ in range(reps + 1):
            vis = ops.fringe_sum(psky, geom)
            g = torch.ones_like(vis)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if name == 'fwd':
                e0.record()
                vis = ops.fringe_sum(psky, geom)
                e1.record()
            else:
                e0.record()
                vis.backward(g)
                e1.record()
                psky.grad = None
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts[1:]))
        res[name] = ms
        print('  %s: %8.3f ms   %.3e elem/s   %.1f TFLOP/s (%.1f%% of 157.3)' % (
            name, ms, E / ms * 1e3, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3 * 100), flush=True)
    return res


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='c2,c4')
    a = ap.parse_args()
    print(ops._lib.version(), torch.cuda.get_device_name(0))
    if 'c2' in a.cases:
        print('C2-like: 171 bl x 30 t x 64 f x 6144 pix, 1-pol real, lmode lift')
        run(171, 30, 64, 6144, blen=30.0)
    if 'c4' in a.cases:
        print('C4-like: 8128 bl x 2 t x 256 f x 108000 pix, 1-pol real')
        run(8128, 2, 256, 108000, blen=100.0, reps=3)
    if 'c4rot' in a.cases:
        print('C4-like, km baselines (standard rotation path)')
        run(8128, 1, 256, 108000, blen=3000.0, reps=3)
    if 'c5' in a.cases:
        print('C5-like slice: 8192 bl x 1 t x 128 f x 100000 pix, 4-pol complex')
        run(8192, 1, 128, 100000, Npp=4, cplx=True, blen=300.0, reps=3)
    if 'f64' in a.cases:
        print('C2-like float64')
        run(171, 4, 64, 6144, blen=30.0, dtype=torch.float64)
