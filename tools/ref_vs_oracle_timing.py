#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE (build container only: imports the reference from /root/reference).
SURVEY.md section 8(d): the CPU oracle (oracle/rime_oracle.py) stands in for the reference as the CPU baseline of
bench.py (`cpu_baseline.kind = "port"`), so its equivalence to the imported reference is established here --
values (float64, <= 1e-12) and wall time (float32, forward + backward, same threads) on the same inputs:
a C2-like shape (hex-19, 171 baselines, diffuse pixel sky, rect-linear PixelBeam).
    python tools/ref_vs_oracle_timing.py            -> prints the table recorded in BASELINE.md
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as mg                       # noqa: E402  (the import recipe of SURVEY.md appendix A)
from oracle import rime_oracle as orc          # noqa: E402


def build(ba, dtype, Nf, Npix, Nt):
    torch.set_default_dtype(dtype)
    freqs = torch.linspace(120e6, 180e6, Nf)
    times = 2459861.0 + np.arange(Nt) * 10.0 / 1440
    arr = mg.hex_array(ba, 3, freqs)
    arr.push(dtype)                               # antenna vectors in the run's dtype (the reference's own push)
    tel = ba.telescope_model.TelescopeModel((21.42827, mg.LAT))
    ra, dec = mg.fib_sky(Npix)
    px_area = 4 * np.pi / Npix
    rng = np.random.default_rng(0)
    sp = torch.as_tensor(rng.normal(size=(1, 1, Nf, len(ra))), dtype=dtype)
    sky = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra, dec]), px_area,
                                R=ba.sky_model.PixelSkyResponse(freqs, cosmo=object()), parameter=True, name='pixsky')
    beam, tg, pg = mg.airy_pixbeam(ba, freqs, dtheta=1.0, dphi=1.0, parameter=True)
    bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    rime = ba.rime_model.RIME(sky, tel, beam, arr, bls, times, freqs)
    zenaz = mg.fill_eq2top(tel, sky.name, ra, dec, times)
    for k in list(tel.conv_cache):
        tel.conv_cache[k] = tel.conv_cache[k].to(dtype)
    ants = arr.ants
    av = mg.npy(arr.antvecs)
    blvecs = torch.as_tensor(np.stack([av[ants.index(j)] - av[ants.index(i)] for i, j in bls]), dtype=dtype)
    return rime, sky, beam, sp, tg, pg, torch.as_tensor(zenaz, dtype=dtype), blvecs, freqs, px_area, len(bls)


def main():
    nthreads = int(os.environ.get('REF_THREADS', os.cpu_count() or 1))
    torch.set_num_threads(nthreads)
    ba = mg.bootstrap_reference()
    # ---- values, float64
    rime, sky, beam, sp, tg, pg, zenaz, blvecs, freqs, px_area, nbl = build(ba, torch.float64, 8, 3000, 2)
    vref = rime().data.detach()
    bp = beam.params.detach().clone()

    def beam_fn(z, a, bp=bp):
        inds, w = orc.rect_interp_weights(tg, pg, z, a, 'linear')
        return orc.interp(orc.pixel_response_forward(bp), inds, w)

    vorc = orc.rime_forward(sp * px_area, zenaz, beam_fn, blvecs, [(0, 0)] * nbl, freqs)
    err = float((vorc - vref).abs().max() / vref.abs().max())
    print('values (float64): max |oracle - reference| / max |reference| = %.2e' % err)
    # ---- wall time, float32, forward + backward
    Nf, Npix, Nt = 64, 12288, 2
    rime, sky, beam, sp, tg, pg, zenaz, blvecs, freqs, px_area, nbl = build(ba, torch.float32, Nf, Npix, Nt)

    def run_ref():
        sky.params.grad = None
        beam.params.grad = None
        v = rime().data
        (v.real ** 2 + v.imag ** 2).sum().backward()

    spo = sp.clone().requires_grad_(True)
    bpo = beam.params.detach().clone().requires_grad_(True)
    cache = {}

    def beam_fn32(z, a):
        key = z.shape[0]
        if key not in cache:                                   # the reference caches its interpolation stencil per time too
            inds, w = orc.rect_interp_weights(tg, pg, z, a, 'linear')
            cache[key] = (inds, w.to(torch.float32))
        inds, w = cache[key]
        return orc.interp(orc.pixel_response_forward(bpo), inds, w)

    def run_orc():
        spo.grad = None
        bpo.grad = None
        v = orc.rime_forward(spo * px_area, zenaz, beam_fn32, blvecs, [(0, 0)] * nbl, freqs)
        (v.real ** 2 + v.imag ** 2).sum().backward()

    res = {}
    for name, fn in (('reference', run_ref), ('oracle', run_orc), ('reference', run_ref), ('oracle', run_orc)):
        fn()                                                    # warm-up (caches, allocator)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        res.setdefault(name, []).append(min(ts))
    tr, to = min(res['reference']), min(res['oracle'])
    nvis = nbl * Nt * Nf
    print('wall time (float32, fwd + bwd, %d threads; hex-19 = %d bl, %d ch, %d times, %d sky px): reference %.3f s '
          '(%.3e vis/s), oracle %.3f s (%.3e vis/s), oracle / reference = %.3f'
          % (nthreads, nbl, Nf, Nt, len(sky.angs[0]), tr, nvis / tr, to, nvis / to, to / tr))


if __name__ == '__main__':
    main()
