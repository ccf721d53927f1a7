"""
imaging.VisMapper.make_map at the headline workload's shape (C4: HERA-128, 8128 baselines, 256 channels, nside-128
pixels, Airy PixelBeam; Nt time steps): dirty maps + PSF diagonal + the 'A2w' normalisation on the fused fringe
kernels (A, 8128 x 256 x 98k complex = 1.6 TB per time step, is never built), timed with HIP events; beside it the
reference's arithmetic (materialised A, einsum) on a bounded sample on the host cores (the CPU oracle).
usage: python tools/bench_imaging.py [nt] [method]
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bayeslim_amd import imaging, dataset, utils, telescope_model, beam_model, ops


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    method = sys.argv[2] if len(sys.argv) > 2 else 'A2w'
    dev = torch.device('cuda:0')
    inp = bench.build_inputs('c4', nt)
    cfg = inp['cfg']
    f32 = torch.float32
    freqs = torch.as_tensor(inp['freqs'], dtype=f32, device=dev)
    antpos = utils.AntposDict(inp['ants'], torch.as_tensor(inp['antvecs']))
    bls = bench.all_baselines(inp)
    tel = telescope_model.TelescopeModel((bench.LON, bench.LAT))
    tg, pg = torch.as_tensor(inp['theta_grid'], device=dev), torch.as_tensor(inp['phi_grid'], device=dev)
    b_phi, b_theta = torch.meshgrid(pg, tg, indexing='xy')
    airy = beam_model.airy_disk(b_theta.ravel() * utils.D2R, b_phi.ravel() * utils.D2R, 14.0, freqs.double(), square=True).to(f32)
    R = beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta_grid=tg, phi_grid=pg, freq_mode='channel',
                                 powerbeam=True, device=dev)
    beam = beam_model.PixelBeam(airy[None, None, None].contiguous(), freqs, R=R, pol='e', powerbeam=True, fov=180,
                                parameter=False)
    gen = torch.Generator(device='cpu').manual_seed(0)
    Nbl, Nf, Npix = len(bls), cfg['Nf'], len(inp['ra'])
    data = torch.complex(torch.randn(1, 1, Nbl, nt, Nf, generator=gen), torch.randn(1, 1, Nbl, nt, Nf, generator=gen)).to(dev)
    icov = (torch.rand(1, 1, Nbl, nt, Nf, generator=gen) + 0.5).to(dev)
    vd = dataset.VisData()
    vd.setup_meta(tel, antpos)
    vd.setup_data(bls, torch.as_tensor(inp['times']), freqs, pol='ee', data=data, icov=icov)
    vm = imaging.VisMapper(vd, inp['ra'], inp['dec'], beam=beam, fov=180, cache_A=True)
    for t, za in zip(inp['times'], inp['zenaz']):
        vm.telescope.conv_cache[(float(t), Npix)] = torch.as_tensor(za, dtype=torch.float64)
    vm.set_normalization(method)

    def run():
        prof = []
        ops.PROFILE = prof
        maps, P = vm.make_map(return_P=True, contract='diag')
        ops.PROFILE = None
        return maps, P, prof

    run()                                                      # geometry / interpolation caches
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        maps, P, prof = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    Pvis = int((inp['zenaz'][:, 0] < 90).sum() / nt)
    elems = Nbl * Nf * Pvis * nt
    kern = {}
    for k in prof:
        kern[k[0]] = kern.get(k[0], 0) + 1
    print('VisMapper.make_map (%s): %d baselines x %d channels x %d times, %d map pixels (~%d above the horizon per time)'
          % (method, Nbl, Nf, nt, Npix, Pvis))
    print('  GPU: %.1f ms per call = %.1f ms per time step; %.3g elements of A per second; kernels %s'
          % (ms, ms / nt, elems / (ms * 1e-3), kern))

    # the reference arithmetic (materialised A, einsum; oracle restatement) on a bounded sample, host cores
    from oracle import rime_oracle as orc
    nb, nf, npx = 1016, 8, 6000
    idx = {a: i for i, a in enumerate(inp['ants'])}
    av = torch.as_tensor(inp['antvecs'], dtype=torch.float64)
    blv = torch.stack([av[idx[b]] - av[idx[a]] for a, b in bls[::Nbl // nb][:nb]])
    za = torch.as_tensor(inp['zenaz'][:1, :, ::max(1, Npix // npx)][:, :, :npx], dtype=torch.float64)
    fq = torch.as_tensor(inp['freqs'][:nf], dtype=torch.float64)
    bmap = airy[:nf].double().cpu()
    tgc, pgc = tg.double().cpu(), pg.double().cpu()

    def beam_fn(zen, az):
        inds, wgts = orc.rect_interp_weights(tgc, pgc, zen, az, 'linear')
        return orc.interp(bmap, inds, wgts)

    vis = data[0, 0, ::Nbl // nb][:nb, :1, :nf].to(torch.complex128).cpu()
    w = icov[0, 0, ::Nbl // nb][:nb, :1, :nf].double().cpu()
    orc.vismapper_make_map(blv[:64], za, fq, vis[:64], w[:64], beam_fn, method=method)        # thread pool warm-up
    t0 = time.perf_counter()
    orc.vismapper_make_map(blv, za, fq, vis, w, beam_fn, method=method)
    dt = time.perf_counter() - t0
    pv = int((za[0, 0] < 90).sum())
    ce = nb * nf * pv
    print('  CPU (oracle, fp64, %d threads): %d baselines x %d channels x %d pixels in %.2f s = %.3g elements/s; GPU/CPU %.0fx'
          % (torch.get_num_threads(), nb, nf, pv, dt, ce / dt, elems / (ms * 1e-3) / (ce / dt)))


if __name__ == '__main__':
    main()
