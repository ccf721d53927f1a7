#!/usr/bin/env python3
"""
Golden-vector generator for the RIME hot path.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where the reference
lives (read-only) at /root/reference; it imports the reference's Python on the
CPU in float64, feeds it seeded synthetic inputs and writes inputs + outputs
(+ gradients) as small .npz files next to this script.  Only those .npz data
files travel to the GPU box; nothing here is imported by the product, the GPU
tests, smoke() or bench.py.

Import recipe: astropy / h5py are absent, so their
top-level imports are satisfied with MagicMock modules; bayeslim/__init__.py is
bypassed (calibration.py needs py>=3.11 syntax); eq2top never reaches astropy
because TelescopeModel.conv_cache is pre-populated with our own (zen, az).

Usage:  python tests/golden/make_golden.py            # regenerates every file
"""
import os
import sys
import types
import importlib
import importlib.machinery
import warnings
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/bayeslim'


def bootstrap_reference():
    sys.dont_write_bytecode = True
    for n in ['astropy', 'astropy.units', 'astropy.constants', 'astropy.coordinates',
              'astropy.time', 'astropy.cosmology', 'h5py']:
        sys.modules[n] = MagicMock(name=n)
    pkg = types.ModuleType('bayeslim')
    pkg.__path__ = [REF]
    pkg.__spec__ = importlib.machinery.ModuleSpec('bayeslim', None, is_package=True)
    sys.modules['bayeslim'] = pkg
    sys.modules['bayeslim.calibration'] = MagicMock()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        mods = {m: importlib.import_module('bayeslim.' + m) for m in
                ['version', 'utils', 'special', 'linalg', 'fft', 'linear_model', 'paramdict',
                 'dataset', 'cosmology', 'sph_harm', 'telescope_model', 'beam_model',
                 'sky_model', 'io', 'rime_model']}
    return types.SimpleNamespace(**mods)


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print('wrote %-28s %7.1f kB' % (name + '.npz', os.path.getsize(path) / 1e3))


# ---------------------------------------------------------------------------
# synthetic inputs (numpy only -- independent of both the reference and the build)
# ---------------------------------------------------------------------------
def random_dirs(rng, n, zen_max=120.0):
    """uniform-on-sphere directions down to zen_max [deg]; returns zen, az in deg"""
    cz = rng.uniform(np.cos(np.deg2rad(zen_max)), 1.0, n)
    zen = np.rad2deg(np.arccos(cz))
    az = rng.uniform(0.0, 360.0, n)
    return zen, az


def radec_to_zenaz(ra, dec, lst, lat):
    """plain LST rotation (degrees in / out); az East of North"""
    H = np.deg2rad(lst - ra)
    d = np.deg2rad(dec)
    p = np.deg2rad(lat)
    x = -np.cos(d) * np.sin(H)
    y = np.sin(d) * np.cos(p) - np.cos(d) * np.sin(p) * np.cos(H)
    z = np.sin(d) * np.sin(p) + np.cos(d) * np.cos(p) * np.cos(H)
    zen = np.rad2deg(np.arccos(np.clip(z, -1, 1)))
    az = np.rad2deg(np.arctan2(x, y)) % 360.0
    return zen, az


def lst_of(jd):
    return (100.0 + 360.0 * 1.00273790935 * (jd - 2459861.0)) % 360.0


LAT = -30.72148


# ---------------------------------------------------------------------------
def gen_fringe_cases(ba):
    rng = np.random.default_rng(1)
    zen, az = random_dirs(rng, 150)
    zen[0], az[0] = 0.0, 0.0                      # zenith: fringe == 1
    blvecs = rng.normal(0, 60, (6, 3))
    blvecs[:, 2] *= 0.02
    blvecs[0] = 0.0                                # auto-correlation
    blvecs[1] = [14.6, 0, 0]
    for tag, freqs in [('uniform', np.linspace(120e6, 180e6, 7)),
                       ('ragged', np.array([100e6, 101e6, 125.5e6, 180e6, 181.25e6]))]:
        antpos = ba.utils.AntposDict([0, 1], np.array([[0., 0, 0], [14.6, 0, 0]]))
        arr = ba.telescope_model.ArrayModel(antpos, freqs=torch.as_tensor(freqs), cache_s=False)
        f1 = arr.gen_fringe(torch.as_tensor(blvecs), torch.as_tensor(zen), torch.as_tensor(az))
        f2 = arr.gen_fringe(torch.as_tensor(blvecs), torch.as_tensor(zen), torch.as_tensor(az),
                            conj=True)
        save('fringe_' + tag, blvecs=blvecs, zen=zen, az=az, freqs=freqs, fringe=f1,
             fringe_conj=f2)


def gen_apply_beam_cases(ba):
    rng = np.random.default_rng(2)
    Nf, P = 4, 37
    freqs = torch.linspace(120e6, 130e6, Nf)
    bls = [(0, 1), (0, 2), (1, 2), (2, 2), (1, 0)]

    def rnd(*shape, comp=False):
        x = rng.normal(size=shape)
        if comp:
            x = x + 1j * rng.normal(size=shape)
        return torch.as_tensor(x)

    cases = {}
    # 1pol power beam
    cases['1pol_power'] = dict(beam=rnd(1, 1, 1, Nf, P).abs(), sky=rnd(1, 1, Nf, P),
                               powerbeam=True, ant2beam=None)
    # 1pol non-power, 3 models, complex antenna beams
    cases['1pol_nonpower_3model'] = dict(beam=rnd(1, 1, 3, Nf, P, comp=True), sky=rnd(1, 1, Nf, P),
                                         powerbeam=False, ant2beam={0: 0, 1: 1, 2: 2})
    # 1pol non-power, 1 model, real beam
    cases['1pol_nonpower_1model'] = dict(beam=rnd(1, 1, 1, Nf, P), sky=rnd(1, 1, Nf, P),
                                         powerbeam=False, ant2beam=None)
    # 1pol full stokes (Nvec = 2)
    cases['1pol_nvec2'] = dict(beam=rnd(1, 2, 2, Nf, P, comp=True), sky=rnd(2, 2, Nf, P, comp=True),
                               powerbeam=False, ant2beam={0: 0, 1: 1, 2: 0})
    # 2pol power
    cases['2pol_power'] = dict(beam=rnd(2, 1, 1, Nf, P).abs(), sky=rnd(1, 1, Nf, P),
                               powerbeam=True, ant2beam=None)
    # 4pol, 1 model and 3 models
    cases['4pol_1model'] = dict(beam=rnd(2, 2, 1, Nf, P, comp=True), sky=rnd(2, 2, Nf, P, comp=True),
                                powerbeam=False, ant2beam=None)
    cases['4pol_3model'] = dict(beam=rnd(2, 2, 3, Nf, P, comp=True), sky=rnd(2, 2, Nf, P, comp=True),
                                powerbeam=False, ant2beam={0: 0, 1: 1, 2: 2})
    out = {}
    for k, c in cases.items():
        b = c['beam']
        pb = ba.beam_model.PixelBeam(b.clone(), freqs, R=None if b.shape[2] == 1 else None,
                                     ant2beam=c['ant2beam'] if c['ant2beam'] is None else None,
                                     parameter=False, powerbeam=c['powerbeam'], pol='e') \
            if c['ant2beam'] is None else None
        if pb is None:
            # ant2beam given: the constructor only installs the default SimpleIndex when
            # ant2beam is None (beam_model.py:153-155), so set the attribute ourselves
            pb = ba.beam_model.PixelBeam.__new__(ba.beam_model.PixelBeam)
            ba.utils.Module.__init__(pb)
            pb.params = b.clone()
            pb.p0 = None
            pb.device = b.device
            pb.powerbeam = c['powerbeam']
            pb.Npol, pb.Nvec, pb.Nmodel = b.shape[:3]
            pb.freqs = freqs
            pb.ant2beam = c['ant2beam']
        psky = pb.apply_beam(b, bls, c['sky'])
        out[k + '__beam'] = b
        out[k + '__sky'] = c['sky']
        out[k + '__psky'] = psky.contiguous()
        out[k + '__powerbeam'] = np.array(c['powerbeam'])
        a2b = c['ant2beam']
        out[k + '__ant2beam'] = np.array([-1]) if a2b is None else np.array([a2b[a] for a in range(3)])
    out['bls'] = np.array(bls)
    save('apply_beam', **out)


def gen_interp_cases(ba):
    rng = np.random.default_rng(3)
    theta_grid = torch.arange(0, 90.1, 1.0)
    phi_grid = torch.arange(0, 360, 1.0)
    # random samples kept clear of exact grid half-way ties (undefined argsort order)
    zen = rng.uniform(0.02, 89.9, 120)
    az = rng.uniform(0.0, 359.999, 120)
    frac = (zen % 1.0)
    zen = np.where(np.abs(frac - 0.5) < 0.02, zen + 0.1, zen)
    frac = (az % 1.0)
    az = np.where(np.abs(frac - 0.5) < 0.02, az + 0.1, az)
    # edge samples: az wrap, on-node, beyond last zenith node, near zenith, first az cell
    edge_zen = np.array([30.2, 45.0, 90.4, 0.3, 10.3, 89.7, 12.25, 0.0])
    edge_az = np.array([359.6, 100.0, 12.3, 200.2, 0.2, 358.8, 0.0, 0.0])
    zen = np.concatenate([zen, edge_zen])
    az = np.concatenate([az, edge_az])
    Npb = len(theta_grid) * len(phi_grid)
    m = torch.as_tensor(rng.normal(size=(2, 3, Npb)))
    out = dict(theta_grid=theta_grid, phi_grid=phi_grid, zen=zen, az=az, m=m)
    for mode in ['nearest', 'linear', 'quadratic', 'cubic', 'linear,quadratic']:
        zz, aa = zen, az
        if mode in ('nearest', 'quadratic', 'linear,quadratic'):
            pass
        PI = ba.utils.PixInterp('rect', interp_mode=mode, theta_grid=theta_grid,
                                phi_grid=phi_grid)
        tz, ta = torch.as_tensor(zz), torch.as_tensor(aa)
        inds, wgts = PI.get_interp(tz, ta)
        mm = m.clone().requires_grad_(True)
        y = PI.interp(mm, tz, ta)
        g = torch.as_tensor(np.random.default_rng(33).normal(size=tuple(y.shape)))
        (y * g).sum().backward()
        key = mode.replace(',', '_')
        out[key + '__inds'] = inds
        out[key + '__wgts'] = wgts
        out[key + '__out'] = y
        out[key + '__gout'] = g
        out[key + '__gm_nnz_idx'] = torch.nonzero(mm.grad.reshape(-1)).reshape(-1)
        out[key + '__gm_nnz_val'] = mm.grad.reshape(-1)[torch.nonzero(mm.grad.reshape(-1)).reshape(-1)]
    # coarse grid as well (2.5 deg), linear only
    tg2, pg2 = torch.arange(0, 92.5, 2.5), torch.arange(0, 360, 5.0)
    PI = ba.utils.PixInterp('rect', interp_mode='linear', theta_grid=tg2, phi_grid=pg2)
    inds, wgts = PI.get_interp(torch.as_tensor(zen), torch.as_tensor(az))
    out['coarse__theta_grid'] = tg2
    out['coarse__phi_grid'] = pg2
    out['coarse__inds'] = inds
    out['coarse__wgts'] = wgts
    # drop the dense m from the file: regenerate from seed in the test instead
    del out['m']
    out['m_seed'] = np.array([3])
    save('interp_rect', **out)


def gen_sph_cases(ba):
    rng = np.random.default_rng(4)
    lmax = 8
    l, m = ba.sph_harm.gen_lm(lmax)
    theta = np.rad2deg(np.arccos(rng.uniform(-1, 1, 50)))
    phi = rng.uniform(0, 360, 50)
    out = dict(lmax=np.array(lmax), l=l, m=m, theta=theta, phi=phi)
    for real in (False, True):
        Y, norm, mult = ba.sph_harm.gen_sph2pix(theta * ba.utils.D2R, phi * ba.utils.D2R, l, m,
                                                high_prec=False, real=real)
        k = 'real' if real else 'comp'
        out['Ylm_' + k] = Y
        out['alm_mult_' + k] = mult
    # separable grid
    tg = np.linspace(1.0, 89.0, 9)
    pg = np.linspace(0.0, 350.0, 12)
    (T, Ph), _, mult = ba.sph_harm.gen_sph2pix(tg * ba.utils.D2R, pg * ba.utils.D2R, l, m,
                                               separable=True, high_prec=False)
    out.update(theta_grid=tg, phi_grid=pg, Theta=T, Phi=Ph, alm_mult_sep=mult)
    # forward_alm, full: complex params (3, Ncoeff)
    a = torch.as_tensor(rng.normal(size=(3, len(l))) + 1j * rng.normal(size=(3, len(l))))
    a[:, m == 0] = a[:, m == 0].real + 0j
    A = ba.sph_harm.AlmModel(l, m, real_output=True)
    A.setup_Ylm(theta, phi, Ylm=out['Ylm_comp'], alm_mult=out['alm_mult_comp'])
    ar = torch.view_as_real(a).clone().requires_grad_(True)
    y = A.forward_alm(ar)
    g = torch.as_tensor(rng.normal(size=tuple(y.shape)))
    (y * g).sum().backward()
    out.update(alm=a, fwd_full=y, gout_full=g, galm_full=ar.grad)
    # separable
    A2 = ba.sph_harm.AlmModel(l, m, real_output=True)
    A2.setup_Ylm(tg, pg, Ylm=(T, Ph), alm_mult=mult, separable=True)
    ar2 = torch.view_as_real(a).clone().requires_grad_(True)
    y2 = A2.forward_alm(ar2)
    g2 = torch.as_tensor(rng.normal(size=tuple(y2.shape)))
    (y2 * g2).sum().backward()
    out.update(fwd_sep=y2, gout_sep=g2, galm_sep=ar2.grad)
    # complex output (real_output False)
    A3 = ba.sph_harm.AlmModel(l, m, real_output=False)
    A3.setup_Ylm(theta, phi, Ylm=out['Ylm_comp'], alm_mult=out['alm_mult_comp'])
    out['fwd_full_complex'] = A3.forward_alm(a)
    save('sph_harm', **out)


def gen_sky_beam_response_cases(ba):
    rng = np.random.default_rng(5)
    freqs = torch.linspace(120e6, 180e6, 6)
    out = dict(freqs=freqs)
    # PointSkyResponse powerlaw
    params = torch.as_tensor(np.stack([rng.uniform(0.5, 3, 9), rng.uniform(-3, -1, 9)])[None, None])
    R = ba.sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs[0])
    out['point_params'] = params
    out['point_powerlaw'] = R(params)
    Rl = ba.sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs[2], log=True)
    out['point_powerlaw_log'] = Rl(params)
    out['point_f0_log'] = freqs[2]
    # Airy / Gauss beams
    zen, az = random_dirs(rng, 40, zen_max=100)
    out.update(zen=zen, az=az)
    tz, ta = torch.as_tensor(zen), torch.as_tensor(az)
    RA = ba.beam_model.AiryResponse(powerbeam=True)
    out['airy_D14'] = RA(torch.ones(1, 1, 1, 1, 1) * 14.0, tz, ta, freqs)
    RA2 = ba.beam_model.AiryResponse(powerbeam=False)
    out['airy_asym'] = RA2(torch.as_tensor([12.0, 15.0]).reshape(1, 1, 1, 1, 2), tz, ta, freqs)
    RG = ba.beam_model.GaussResponse(powerbeam=True)
    gp = torch.as_tensor(rng.uniform(0.2, 0.5, (1, 1, 1, 6, 2)))
    out['gauss_params'] = gp
    out['gauss'] = RG(gp, tz, ta, freqs)
    # PixelResponse.forward variants on a tiny rect grid
    tg, pg = torch.arange(0, 91, 10.0), torch.arange(0, 360, 30.0)
    Npb = len(tg) * len(pg)
    p = torch.as_tensor(rng.normal(size=(1, 1, 1, 6, Npb)))
    b0 = torch.as_tensor(rng.uniform(0, 1, size=(1, 1, 1, 6, Npb)))
    out.update(pr_theta_grid=tg, pr_phi_grid=pg, pr_params=p, pr_beam0=b0)
    for tag, kw in [('abs', dict()), ('log', dict(log=True)), ('beam0', dict(beam0=b0.clone())),
                    ('normpix', dict(norm_pix=3)), ('nonpower', dict(powerbeam=False))]:
        Rp = ba.beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta_grid=tg,
                                         phi_grid=pg, **kw)
        out['pr_fwd_' + tag] = Rp.forward(p.clone())
    save('responses', **out)


# ---------------------------------------------------------------------------
# full RIME cases
# ---------------------------------------------------------------------------
def hex_array(ba, N, freqs, D=14.6, extra=None):
    ants, vecs = ba.utils._make_hex(N, D=D)
    if extra is not None:
        ants = list(ants) + [len(ants)]
        vecs = np.vstack([vecs, extra])
    antpos = ba.utils.AntposDict(ants, vecs)
    arr = ba.telescope_model.ArrayModel(antpos, freqs=freqs, cache_s=True, redtol=1.0)
    return arr


def airy_pixbeam(ba, freqs, D=14.0, dtheta=5.0, dphi=10.0, interp_mode='linear', parameter=True):
    theta = torch.arange(0, 90 + dtheta / 2, dtheta)
    phi = torch.arange(0, 360, dphi)
    b_phi, b_theta = torch.meshgrid(phi, theta, indexing='xy')
    b_phi, b_theta = b_phi.ravel(), b_theta.ravel()
    airy = ba.beam_model.airy_disk(b_theta * ba.utils.D2R, b_phi * ba.utils.D2R, D, freqs,
                                   square=True)
    R = ba.beam_model.PixelResponse(freqs, 'rect', interp_mode=interp_mode,
                                    theta=b_theta, phi=b_phi, theta_grid=theta, phi_grid=phi,
                                    freq_mode='channel', powerbeam=True, realbeam=True)
    p = torch.as_tensor(airy[None, None, None, :, :]).clone()
    beam = ba.beam_model.PixelBeam(p, freqs, R=R, pol='e', powerbeam=True, fov=180,
                                   parameter=parameter)
    return beam, theta, phi


def fill_eq2top(telescope, sky_name, ra, dec, times):
    zas = []
    for t in times:
        zen, az = radec_to_zenaz(npy(ra), npy(dec), lst_of(float(t)), LAT)
        telescope.conv_cache[(sky_name, len(ra), float(t))] = torch.stack(
            [torch.as_tensor(zen), torch.as_tensor(az)])
        zas.append(np.stack([zen, az]))
    return np.stack(zas)


def run_rime(ba, rime, params):
    """forward + backward of loss = sum(w * |V|^2)-like functional with fixed random weights"""
    vd = rime()
    V = vd.data
    rng = np.random.default_rng(77)
    gw = torch.as_tensor(rng.normal(size=tuple(V.shape)) + 1j * rng.normal(size=tuple(V.shape)))
    loss = (V * gw.conj()).real.sum()
    for p in params:
        if p.grad is not None:
            p.grad = None
    loss.backward()
    return V.detach(), gw, [p.grad.detach().clone() for p in params]


def gen_rime_c1(ba):
    """config 1: 7-antenna hex, 10 point sources, 8 freqs, 2 times (CPU plumbing case)"""
    freqs = torch.linspace(120e6, 130e6, 8)
    times = np.array([2459861.0, 2459861.0 + 10.0 / 1440])
    arr = hex_array(ba, 2, freqs)
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    Nsrc = 10
    R = ba.sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs[0])
    rng = np.random.default_rng(10)
    params = torch.ones(1, 1, 2, Nsrc)
    params[..., 0, :] = torch.as_tensor(rng.uniform(0.5, 2.0, Nsrc))
    params[..., 1, :] = -2.2
    ra = torch.as_tensor(lst_of(times[0]) + rng.uniform(-40, 40, Nsrc))
    dec = torch.as_tensor(LAT + rng.uniform(-35, 35, Nsrc))
    angs = torch.stack([ra, dec])
    sky = ba.sky_model.PointSky(params.clone(), angs, R=R, parameter=True, name='ptsky')
    beam = ba.beam_model.PixelBeam(torch.ones(1, 1, 1, 1, 1) * 14.0, freqs,
                                   R=ba.beam_model.AiryResponse(powerbeam=True), pol='e',
                                   powerbeam=True, fov=180, parameter=False)
    sim_bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
    zenaz = fill_eq2top(tel, sky.name, ra, dec, times)
    V, gw, grads = run_rime(ba, rime, [sky.params])
    save('rime_c1', freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
         sim_bls=np.array(sim_bls), sky_params=params, ra=ra, dec=dec, zenaz=zenaz,
         vis=V, gvis=gw, g_sky_params=grads[0], airy_D=np.array(14.0))


def gen_rime_c2_mini(ba):
    """mini config 2: hex-19 (171 bl), diffuse pixel sky, rect-linear interpolated PixelBeam,
    fwd + grads wrt sky pixels and beam map; time-minibatched == unbatched; data_bls inflation"""
    Nf = 8
    freqs = torch.linspace(120e6, 180e6, Nf)
    times = 2459861.0 + np.arange(3) * 10.0 / 1440
    arr = hex_array(ba, 3, freqs)
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    rng = np.random.default_rng(20)
    Npix = 600
    # quasi-uniform sky directions (Fibonacci lattice) cut at dec < 59.27852 like tests/test_sky.py
    k = np.arange(Npix) + 0.5
    dec = np.rad2deg(np.arcsin(1 - 2 * k / Npix))
    ra = (k * 137.50776405) % 360.0
    keep = dec < 59.27852
    ra, dec = torch.as_tensor(ra[keep]), torch.as_tensor(dec[keep])
    px_area = 4 * np.pi / Npix
    Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
    sp = torch.as_tensor(rng.normal(size=(1, 1, Nf, len(ra))))
    sky = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra, dec]), px_area, R=Rs,
                                parameter=True, name='pixsky')
    beam, tg, pg = airy_pixbeam(ba, freqs, parameter=True)
    sim_bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
    zenaz = fill_eq2top(tel, sky.name, ra, dec, times)
    V, gw, grads = run_rime(ba, rime, [sky.params, beam.params])
    # minibatched over time: must equal unbatched (tests/test_rime.py:41-51)
    rime.setup_sim_times(ba.utils.split_into_groups(times, Nelem=2))
    with torch.no_grad():
        Vb = rime.run_batches().data
    assert (Vb - V).abs().max() < 1e-10
    # redundant inflation: simulate unique bls only, inflate to all data_bls grouped by redundancy
    uniq = arr.get_bls(uniq_bls=True, keep_autos=False)
    data_bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    rime2 = ba.rime_model.RIME(sky, tel, beam, arr, uniq, times, freqs, data_bls=data_bls)
    with torch.no_grad():
        V2 = rime2().data
    save('rime_c2_mini', freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
         sim_bls=np.array(sim_bls), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
         sky_params=sp, beam_params=beam.params.detach(), theta_grid=tg, phi_grid=pg,
         vis=V, gvis=gw, g_sky_params=grads[0], g_beam_params=grads[1],
         uniq_bls=np.array(uniq), data_bls=np.array(rime2.data_bls),
         sim2data=rime2._sim2data[0], vis_inflated=V2)


def gen_rime_c3_mini(ba):
    """mini config 3: a_lm sky (AlmModel, lmax 6) + YlmResponse beam (interpolate mode, rect grid)"""
    Nf = 6
    freqs = torch.linspace(120e6, 180e6, Nf)
    times = 2459861.0 + np.arange(2) * 10.0 / 1440
    arr = hex_array(ba, 2, freqs)
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    rng = np.random.default_rng(30)
    Npix = 400
    k = np.arange(Npix) + 0.5
    dec = np.rad2deg(np.arcsin(1 - 2 * k / Npix))
    ra = (k * 137.50776405) % 360.0
    ra_t, dec_t = torch.as_tensor(ra), torch.as_tensor(dec)
    # sky a_lm
    lmax = 6
    l, m = ba.sph_harm.gen_lm(lmax)
    Ncoeff = len(l)
    colat = 90.0 - dec
    Ysky, _, mult = ba.sph_harm.gen_sph2pix(colat * ba.utils.D2R, ra * ba.utils.D2R, l, m,
                                            high_prec=False)
    A = ba.sph_harm.AlmModel(l, m, real_output=True)
    A.setup_Ylm(colat, ra, Ylm=Ysky, alm_mult=mult)
    a = rng.normal(size=(1, 1, Nf, Ncoeff)) + 1j * rng.normal(size=(1, 1, Nf, Ncoeff))
    a /= (1.0 + l)
    a[..., m == 0] = a[..., m == 0].real
    sp = torch.view_as_real(torch.as_tensor(a)).clone()
    Rs = ba.sky_model.PixelSkyResponse(freqs, spatial_mode='alm', spat_LM=A, comp_params=False,
                                       cosmo=object())
    px_area = 4 * np.pi / Npix
    sky = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra_t, dec_t]), px_area, R=Rs,
                                parameter=True, name='almsky')
    # beam a_lm on a rect grid, interpolate mode
    bl_, bm_ = ba.sph_harm.gen_lm(4)
    tg = torch.arange(0, 91, 5.0)
    pg = torch.arange(0, 360, 10.0)
    b_phi, b_theta = torch.meshgrid(pg, tg, indexing='xy')
    b_phi, b_theta = b_phi.ravel(), b_theta.ravel()
    Yb, _, bmult = ba.sph_harm.gen_sph2pix(npy(b_theta) * ba.utils.D2R, npy(b_phi) * ba.utils.D2R,
                                           bl_, bm_, high_prec=False)
    RB = ba.beam_model.YlmResponse(bl_, bm_, freqs, pixtype='rect', mode='interpolate',
                                   interp_mode='linear', theta=b_theta, phi=b_phi,
                                   theta_grid=tg, phi_grid=pg, powerbeam=True, comp_params=True)
    RB.set_Ylm(Yb, (b_theta, b_phi), alm_mult=bmult)
    bp = rng.normal(size=(1, 1, 1, Nf, len(bl_))) + 1j * rng.normal(size=(1, 1, 1, Nf, len(bl_)))
    bp /= (1.0 + bl_) ** 2
    bp[..., 0] += 3.0
    bp[..., bm_ == 0] = bp[..., bm_ == 0].real
    bpr = torch.view_as_real(torch.as_tensor(bp)).clone()
    beam = ba.beam_model.PixelBeam(bpr.clone(), freqs, R=RB, pol='e', powerbeam=True, fov=180,
                                   parameter=True)
    sim_bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
    zenaz = fill_eq2top(tel, sky.name, ra_t, dec_t, times)
    V, gw, grads = run_rime(ba, rime, [sky.params, beam.params])
    with torch.no_grad():
        skymap = sky().data
        RB.clear_beam_cache()
        bcache = RB.set_beam_cache(beam.params.detach())
    save('rime_c3_mini', freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
         sim_bls=np.array(sim_bls), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
         sky_l=l, sky_m=m, sky_params=sp, sky_map=skymap,
         beam_l=bl_, beam_m=bm_, beam_params=bpr, beam_cache=bcache, theta_grid=tg, phi_grid=pg,
         vis=V, gvis=gw, g_sky_params=grads[0], g_beam_params=grads[1])


def gen_rime_c5_mini(ba):
    """mini config 5: 4-pol (Npol = Nvec = 2) Jones PixelBeam (2 beam models), coherency sky from
    Stokes2Coherency of (I, 0.1 I, 0.05 I) (no V, so the coherency stays real).  Beam and sky
    are REAL-valued: under torch 2.10 the reference's PixInterp.interp raises on a complex map
    (einsum of complex `nearest` with real `wgts`, utils.py:841) and apply_beam's einsum raises
    on a real beam with a complex sky (beam_model.py:363), so the all-real case is the one the
    reference can run end to end.  Complex psky is pinned by apply_beam.npz instead."""
    Nf = 5
    freqs = torch.linspace(120e6, 180e6, Nf)
    times = 2459861.0 + np.arange(2) * 10.0 / 1440
    arr = hex_array(ba, 2, freqs, extra=np.array([[120.0, -35.0, 1.5]]))
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    rng = np.random.default_rng(50)
    Npix = 300
    k = np.arange(Npix) + 0.5
    dec = np.rad2deg(np.arcsin(1 - 2 * k / Npix))
    ra = (k * 137.50776405) % 360.0
    ra_t, dec_t = torch.as_tensor(ra), torch.as_tensor(dec)
    I = torch.as_tensor(np.abs(rng.normal(size=(1, 1, Nf, Npix))))
    S2C = ba.sky_model.Stokes2Coherency(
        params=torch.as_tensor(np.array([0.1, 0.05]).reshape(2, 1, 1, 1)) * torch.ones(2, 1, Nf, Npix))
    Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
    px_area = 4 * np.pi / Npix
    stokes_sky = ba.sky_model.PixelSky(I.clone(), torch.stack([ra_t, dec_t]), px_area, R=Rs,
                                       parameter=True, name='polsky')

    class CohSky(ba.utils.Module):
        """Stokes-I PixelSky followed by Stokes2Coherency, exposing the sky interface RIME uses"""
        def __init__(self, sky, s2c):
            super().__init__(name='cohsky')
            self.sky = sky
            self.s2c = s2c
            self.device = sky.device

        def forward(self, prior_cache=None, **kw):
            return self.s2c(self.sky(prior_cache=prior_cache))

    sky = CohSky(stokes_sky, S2C)
    # Jones beam: airy amplitude x small random phase, per (pol, vec, model)
    tg = torch.arange(0, 91, 5.0)
    pg = torch.arange(0, 360, 10.0)
    b_phi, b_theta = torch.meshgrid(pg, tg, indexing='xy')
    b_phi, b_theta = b_phi.ravel(), b_theta.ravel()
    airy = ba.beam_model.airy_disk(b_theta * ba.utils.D2R, b_phi * ba.utils.D2R, 14.0, freqs,
                                   square=False)
    amp = npy(airy)[None, None, None] * rng.uniform(0.8, 1.2, (2, 2, 2, 1, 1))
    amp[0, 1] *= 0.1
    amp[1, 0] *= 0.1
    ph = rng.normal(0, 0.2, (2, 2, 2, Nf, amp.shape[-1]))
    Jr = torch.as_tensor(amp * np.cos(ph)).clone()
    R = ba.beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta=b_theta, phi=b_phi,
                                    theta_grid=tg, phi_grid=pg, freq_mode='channel',
                                    powerbeam=False, realbeam=True, comp_params=False)
    ants = arr.ants
    ant2beam = {a: (i % 2) for i, a in enumerate(ants)}
    beam = ba.beam_model.PixelBeam(Jr.clone(), freqs, R=R, ant2beam=ant2beam, powerbeam=False,
                                   fov=180, parameter=True)
    beam.ant2beam = ant2beam
    sim_bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
    zenaz = fill_eq2top(tel, 'polsky', ra_t, dec_t, times)
    V, gw, grads = run_rime(ba, rime, [stokes_sky.params, beam.params])
    save('rime_c5_mini', freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
         sim_bls=np.array(sim_bls), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
         stokes_I=I, frac_pol=np.array([0.1, 0.05]), beam_params=Jr,
         ant2beam=np.array([ant2beam[a] for a in ants]), theta_grid=tg, phi_grid=pg,
         vis=V, gvis=gw, g_sky_params=grads[0], g_beam_params=grads[1])


def fib_sky(Npix, cut=True):
    """quasi-uniform (Fibonacci lattice) directions, optionally cut at dec < 59.27852 (tests/test_sky.py:18)"""
    k = np.arange(Npix) + 0.5
    dec = np.rad2deg(np.arcsin(1 - 2 * k / Npix))
    ra = (k * 137.50776405) % 360.0
    if cut:
        keep = dec < 59.27852
        ra, dec = ra[keep], dec[keep]
    return torch.as_tensor(ra), torch.as_tensor(dec)


def gen_rime_mfma_arrays(ba):
    """
    Arrays large enough (>= 33 antennas) that the build's antenna-factored matrix-core kernels serve
    them: hex-37 (666 baselines) and a 70-antenna random array (2415 baselines); diffuse pixel sky,
    rect-linear interpolated PixelBeam, visibilities + gradients w.r.t. sky pixels and beam map.
    """
    for tag, Nf, Npix, seed in [('hex37', 4, 500, 21), ('rand70', 3, 400, 22)]:
        freqs = torch.linspace(120e6, 180e6, Nf)
        times = 2459861.0 + np.arange(2) * 10.0 / 1440
        rng = np.random.default_rng(seed)
        if tag == 'hex37':
            arr = hex_array(ba, 4, freqs)
        else:
            vecs = np.stack([rng.uniform(-150, 150, 70), rng.uniform(-150, 150, 70), rng.normal(0, 0.5, 70)], 1)
            arr = ba.telescope_model.ArrayModel(ba.utils.AntposDict(list(range(70)), vecs), freqs=freqs,
                                                cache_s=True, redtol=1.0)
        tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
        ra, dec = fib_sky(Npix)
        px_area = 4 * np.pi / Npix
        Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
        sp = torch.as_tensor(rng.normal(size=(1, 1, Nf, len(ra))))
        sky = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra, dec]), px_area, R=Rs,
                                    parameter=True, name='pixsky')
        beam, tg, pg = airy_pixbeam(ba, freqs, parameter=True)
        ants = arr.ants
        sim_bls = [(ants[i], ants[j]) for i in range(len(ants)) for j in range(i + 1, len(ants))]
        rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
        zenaz = fill_eq2top(tel, sky.name, ra, dec, times)
        V, gw, grads = run_rime(ba, rime, [sky.params, beam.params])
        save('rime_%s_mini' % tag, freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
             sim_bls=np.array(sim_bls), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
             sky_params=sp, beam_params=beam.params.detach(), theta_grid=tg, phi_grid=pg,
             vis=V, gvis=gw, g_sky_params=grads[0], g_beam_params=grads[1])


def gen_rime_mfma_large(ba):
    """
    Reference outputs for the kernels of the HEADLINE configuration and of arrays beyond one antenna group: 128 random
    antennas (8128 baselines: the four-row-tile forward / backward kernels of config 4) and 150 random antennas (11 175
    baselines: group blocks, i.e. diagonal + cross kernels), and the headline array itself (127-antenna hexagon + outrigger:
    the mirror-pair variants of the same kernels); diffuse pixel sky (signed), rect-linear interpolated
    PixelBeam, visibilities + gradients w.r.t. sky pixels and beam map.  Small in every other dimension (2 channels,
    2 times, 600 / 400 directions) so that the files stay below 1.5 MB each.
    """
    for tag, Nant, Nf, Npix, seed in [('rand128', 128, 2, 600, 24), ('rand150', 150, 2, 400, 25), ('hex128', 128, 2, 600, 26)]:
        freqs = torch.linspace(130e6, 170e6, Nf)
        times = 2459861.0 + np.arange(2) * 10.0 / 1440
        rng = np.random.default_rng(seed)
        if tag == 'hex128':
            # the headline ARRAY itself (bench.py::hera_array('hera128')): 127-antenna hexagon + one outrigger, i.e. 63
            # point-symmetric antenna pairs + 2 singles -- the layout the mirror-pair kernels are built for
            arr = hex_array(ba, 7, freqs, extra=[[250.0, 0.0, 0.0]])
            assert len(arr.ants) == Nant
        else:
            vecs = np.stack([rng.uniform(-200, 200, Nant), rng.uniform(-200, 200, Nant), rng.normal(0, 0.5, Nant)], 1)
            arr = ba.telescope_model.ArrayModel(ba.utils.AntposDict(list(range(Nant)), vecs), freqs=freqs,
                                                cache_s=True, redtol=1.0)
        tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
        ra, dec = fib_sky(Npix)
        px_area = 4 * np.pi / Npix
        Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
        sp = torch.as_tensor(rng.normal(size=(1, 1, Nf, len(ra))))
        sky = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra, dec]), px_area, R=Rs,
                                    parameter=True, name='pixsky')
        beam, tg, pg = airy_pixbeam(ba, freqs, parameter=True)
        ants = arr.ants
        sim_bls = [(ants[i], ants[j]) for i in range(len(ants)) for j in range(i + 1, len(ants))]
        rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
        zenaz = fill_eq2top(tel, sky.name, ra, dec, times)
        V, gw, grads = run_rime(ba, rime, [sky.params, beam.params])
        save('rime_%s_mini' % tag, freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
             sim_bls=np.array(sim_bls, dtype=np.int16), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
             sky_params=sp, beam_params=beam.params.detach(), theta_grid=tg, phi_grid=pg,
             vis=V, gvis=gw, g_sky_params=grads[0], g_beam_params=grads[1])


def gen_rime_pol_mfma(ba):
    """
    4-pol on an array the build serves with its matrix-core kernels: 40 random antennas (780 baselines), (2,2) Jones
    PixelBeam with two beam models (four model pairs), coherency sky from Stokes2Coherency of (I, 0.1 I, 0.05 I) --
    gen_rime_c5_mini's model (same restrictions: all-real beam and sky, the case the reference runs end to end) on an
    array above the 16-antenna threshold.
    """
    Nf, Nant = 3, 40
    freqs = torch.linspace(120e6, 180e6, Nf)
    times = 2459861.0 + np.arange(2) * 10.0 / 1440
    rng = np.random.default_rng(51)
    vecs = np.stack([rng.uniform(-120, 120, Nant), rng.uniform(-120, 120, Nant), rng.normal(0, 0.5, Nant)], 1)
    arr = ba.telescope_model.ArrayModel(ba.utils.AntposDict(list(range(Nant)), vecs), freqs=freqs,
                                        cache_s=True, redtol=1.0)
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    Npix = 300
    k = np.arange(Npix) + 0.5
    dec = np.rad2deg(np.arcsin(1 - 2 * k / Npix))
    ra = (k * 137.50776405) % 360.0
    ra_t, dec_t = torch.as_tensor(ra), torch.as_tensor(dec)
    I = torch.as_tensor(np.abs(rng.normal(size=(1, 1, Nf, Npix))))
    S2C = ba.sky_model.Stokes2Coherency(
        params=torch.as_tensor(np.array([0.1, 0.05]).reshape(2, 1, 1, 1)) * torch.ones(2, 1, Nf, Npix))
    Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
    px_area = 4 * np.pi / Npix
    stokes_sky = ba.sky_model.PixelSky(I.clone(), torch.stack([ra_t, dec_t]), px_area, R=Rs,
                                       parameter=True, name='polsky')

    class CohSky(ba.utils.Module):
        def __init__(self, sky, s2c):
            super().__init__(name='cohsky')
            self.sky = sky
            self.s2c = s2c
            self.device = sky.device

        def forward(self, prior_cache=None, **kw):
            return self.s2c(self.sky(prior_cache=prior_cache))

    sky = CohSky(stokes_sky, S2C)
    tg = torch.arange(0, 91, 5.0)
    pg = torch.arange(0, 360, 10.0)
    b_phi, b_theta = torch.meshgrid(pg, tg, indexing='xy')
    b_phi, b_theta = b_phi.ravel(), b_theta.ravel()
    airy = ba.beam_model.airy_disk(b_theta * ba.utils.D2R, b_phi * ba.utils.D2R, 14.0, freqs, square=False)
    amp = npy(airy)[None, None, None] * rng.uniform(0.8, 1.2, (2, 2, 2, 1, 1))
    amp[0, 1] *= 0.1
    amp[1, 0] *= 0.1
    ph = rng.normal(0, 0.2, (2, 2, 2, Nf, amp.shape[-1]))
    Jr = torch.as_tensor(amp * np.cos(ph)).clone()
    R = ba.beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta=b_theta, phi=b_phi,
                                    theta_grid=tg, phi_grid=pg, freq_mode='channel',
                                    powerbeam=False, realbeam=True, comp_params=False)
    ants = arr.ants
    ant2beam = {a: (i % 2) for i, a in enumerate(ants)}
    beam = ba.beam_model.PixelBeam(Jr.clone(), freqs, R=R, ant2beam=ant2beam, powerbeam=False,
                                   fov=180, parameter=True)
    beam.ant2beam = ant2beam
    sim_bls = [(ants[i], ants[j]) for i in range(len(ants)) for j in range(i + 1, len(ants))]
    rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
    zenaz = fill_eq2top(tel, 'polsky', ra_t, dec_t, times)
    V, gw, grads = run_rime(ba, rime, [stokes_sky.params, beam.params])
    save('rime_pol40_mini', freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
         sim_bls=np.array(sim_bls), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
         stokes_I=I, frac_pol=np.array([0.1, 0.05]), beam_params=Jr,
         ant2beam=np.array([ant2beam[a] for a in ants]), theta_grid=tg, phi_grid=pg,
         vis=V, gvis=gw, g_sky_params=grads[0], g_beam_params=grads[1])


def gen_rime_composite(ba):
    """
    Diffuse pixel sky + point sources through ONE beam: the sky of the headline benchmark.  The
    reference's RIME.forward raises on a multi-component sky (torch.sum over a list,
    rime_model.py:377), so the golden is two reference RIMEs -- one per component, sharing the beam
    parameter -- with their visibilities added (SURVEY.md section 3.1 note); gradients flow through
    the sum.  Two array sizes: hex-7 (vector-ALU kernels in the build) and hex-37 (matrix-core kernels).
    """
    for tag, N in [('hex7', 2), ('hex37', 4)]:
        Nf = 4
        freqs = torch.linspace(120e6, 180e6, Nf)
        times = 2459861.0 + np.arange(2) * 10.0 / 1440
        rng = np.random.default_rng(23 + N)
        arr = hex_array(ba, N, freqs)
        tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
        Npix = 400
        ra, dec = fib_sky(Npix)
        px_area = 4 * np.pi / Npix
        Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
        sp = torch.as_tensor(np.abs(rng.normal(size=(1, 1, Nf, len(ra)))))
        diffuse = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra, dec]), px_area, R=Rs,
                                        parameter=True, name='diffuse')
        Nsrc = 25
        Rp = ba.sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs[0])
        pp = torch.ones(1, 1, 2, Nsrc)
        pp[..., 0, :] = torch.as_tensor(rng.uniform(0.01, 0.05, Nsrc))
        pp[..., 1, :] = -2.2
        pra = torch.as_tensor(lst_of(times[0]) + rng.uniform(-60, 60, Nsrc))
        pdec = torch.as_tensor(LAT + rng.uniform(-50, 50, Nsrc))
        points = ba.sky_model.PointSky(pp.clone(), torch.stack([pra, pdec]), R=Rp, parameter=True,
                                       name='points')
        beam, tg, pg = airy_pixbeam(ba, freqs, parameter=True)
        ants = arr.ants
        sim_bls = [(ants[i], ants[j]) for i in range(len(ants)) for j in range(i + 1, len(ants))]
        zenaz = fill_eq2top(tel, 'diffuse', ra, dec, times)
        pt_zenaz = fill_eq2top(tel, 'points', pra, pdec, times)
        r1 = ba.rime_model.RIME(diffuse, tel, beam, arr, sim_bls, times, freqs)
        r2 = ba.rime_model.RIME(points, tel, beam, arr, sim_bls, times, freqs)
        V = r1().data + r2().data
        g = np.random.default_rng(77)
        gw = torch.as_tensor(g.normal(size=tuple(V.shape)) + 1j * g.normal(size=tuple(V.shape)))
        loss = (V * gw.conj()).real.sum()
        params = [diffuse.params, points.params, beam.params]
        grads = torch.autograd.grad(loss, params)
        save('rime_composite_%s' % tag, freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
             sim_bls=np.array(sim_bls), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
             sky_params=sp, pt_params=pp, pt_ra=pra, pt_dec=pdec, pt_zenaz=pt_zenaz,
             beam_params=beam.params.detach(), theta_grid=tg, phi_grid=pg,
             vis=V.detach(), gvis=gw, g_sky_params=grads[0], g_pt_params=grads[1], g_beam_params=grads[2])


def gen_rime_two_models(ba):
    """
    hex-37 with TWO antenna beam models (1-pol, non-power real voltage beams; ant2beam alternates),
    i.e. four beam-model pairs (0,0) (0,1) (1,0) (1,1) (beam_model.py:303-327): the case the build
    must serve on its matrix-core kernels with per-antenna beams; visibilities + gradients.
    """
    Nf = 4
    freqs = torch.linspace(120e6, 180e6, Nf)
    times = 2459861.0 + np.arange(2) * 10.0 / 1440
    rng = np.random.default_rng(31)
    arr = hex_array(ba, 4, freqs)
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    Npix = 400
    ra, dec = fib_sky(Npix)
    px_area = 4 * np.pi / Npix
    Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
    sp = torch.as_tensor(rng.normal(size=(1, 1, Nf, len(ra))))
    sky = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra, dec]), px_area, R=Rs, parameter=True, name='pixsky')
    tg = torch.arange(0, 91, 5.0)
    pg = torch.arange(0, 360, 10.0)
    b_phi, b_theta = torch.meshgrid(pg, tg, indexing='xy')
    b_phi, b_theta = b_phi.ravel(), b_theta.ravel()
    airy = npy(ba.beam_model.airy_disk(b_theta * ba.utils.D2R, b_phi * ba.utils.D2R, 14.0, freqs, square=False))
    amp = airy[None, None, None] * np.array([1.0, 0.85]).reshape(1, 1, 2, 1, 1)
    amp = amp * (1.0 + 0.1 * rng.normal(size=(1, 1, 2, Nf, amp.shape[-1])))
    bp = torch.as_tensor(amp).clone()
    R = ba.beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta=b_theta, phi=b_phi,
                                    theta_grid=tg, phi_grid=pg, freq_mode='channel',
                                    powerbeam=False, realbeam=True, comp_params=False)
    ants = arr.ants
    ant2beam = {a: (i % 2) for i, a in enumerate(ants)}
    beam = ba.beam_model.PixelBeam(bp.clone(), freqs, R=R, ant2beam=ant2beam, pol='e', powerbeam=False,
                                   fov=180, parameter=True)
    beam.ant2beam = ant2beam
    sim_bls = [(ants[i], ants[j]) for i in range(len(ants)) for j in range(i + 1, len(ants))]
    rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, times, freqs)
    zenaz = fill_eq2top(tel, 'pixsky', ra, dec, times)
    V, gw, grads = run_rime(ba, rime, [sky.params, beam.params])
    save('rime_two_models_hex37', freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants),
         sim_bls=np.array(sim_bls), ra=ra, dec=dec, zenaz=zenaz, px_area=np.array(px_area),
         sky_params=sp, beam_params=bp, ant2beam=np.array([ant2beam[a] for a in ants]),
         theta_grid=tg, phi_grid=pg, vis=V, gvis=gw, g_sky_params=grads[0], g_beam_params=grads[1])


def gen_prod_and_sum(ba):
    """RIME._prod_and_sum in isolation (rime_model.py:391-440), with and without sim2data"""
    rng = np.random.default_rng(60)
    Nf, P = 5, 64
    freqs = torch.linspace(120e6, 180e6, Nf)
    arr = hex_array(ba, 2, freqs)
    arr.cache_s = False
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    bls = [(0, 1), (0, 2), (1, 3), (2, 6)]
    blvecs = arr.get_blvecs(bls)
    zen, az = random_dirs(rng, P, zen_max=89)
    beam_t = torch.as_tensor(np.abs(rng.normal(size=(1, 1, 1, Nf, P))))
    sky_t = torch.as_tensor(rng.normal(size=(1, 1, Nf, P)))
    pb = ba.beam_model.PixelBeam(torch.ones(1, 1, 1, Nf, 1), freqs, parameter=False, pol='e')
    dummy_sky = ba.sky_model.PointSky(torch.ones(1, 1, Nf, 1), torch.zeros(2, 1),
                                      R=ba.sky_model.PointSkyResponse(freqs, freq_mode='channel'),
                                      parameter=False)
    rime = ba.rime_model.RIME(dummy_sky, tel, pb, arr, bls, np.array([2459861.0]), freqs)
    vis = []
    rime._prod_and_sum(beam_t, sky_t, bls, blvecs, torch.as_tensor(zen), torch.as_tensor(az),
                       vis, None, 0)
    idx = torch.as_tensor([0, 0, 1, 2, 2, 2, 3])
    rime._prod_and_sum(beam_t, sky_t, bls, blvecs, torch.as_tensor(zen), torch.as_tensor(az),
                       vis, idx, 0)
    save('prod_and_sum', freqs=freqs, blvecs=blvecs, zen=zen, az=az, beam=beam_t, sky=sky_t,
         sum_sky=vis[0], sim2data_idx=idx, sum_sky_inflated=vis[1])


def gen_chisq(ba):
    """diagonal-inverse-covariance chi-square of a visibility residual: optim.apply_icov with
    cov_axis=None (optim.py:1836-1915) as LogProb.forward_chisq uses it (optim.py:1019-1027), value
    and gradient w.r.t. the prediction"""
    import importlib
    optim = importlib.import_module('bayeslim.optim')
    rng = np.random.default_rng(7)
    shape = (2, 2, 13, 3, 5)
    pred = torch.as_tensor(rng.normal(size=shape) + 1j * rng.normal(size=shape))
    data = torch.as_tensor(rng.normal(size=shape) + 1j * rng.normal(size=shape))
    icov = torch.as_tensor(rng.uniform(0.1, 3.0, size=shape))
    out = {}
    for tag, ic in (('icov', icov), ('noicov', None)):
        p = pred.clone().requires_grad_(True)
        res = p - data
        chisq = optim.apply_icov(res, ic, None)
        tot = torch.sum(chisq)
        tot = tot.real if torch.is_complex(tot) else tot
        tot.backward()
        out['chisq_' + tag] = chisq
        out['sum_' + tag] = tot
        out['gpred_' + tag] = p.grad
    save('chisq', pred=pred, data=data, icov=icov, **out)


def gen_imaging(ba):
    """map-making arithmetic of imaging.py on one time step: A = conj(fringe) * beam as VisMapper.build_A
    builds it (imaging.py:251-296), then make_map (:717-736), compute_Am (:755-774) and compute_Pm
    (:777-815) -- the adjoint / forward uses of the RIME fringe"""
    import importlib
    imaging = importlib.import_module('bayeslim.imaging')
    rng = np.random.default_rng(11)
    Nant, Nf, P = 9, 6, 140
    ant = rng.normal(0, 40.0, (Nant, 3)); ant[:, 2] *= 0.02
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    blvecs = np.stack([ant[j] - ant[i] for i, j in pairs])
    freqs = np.linspace(120e6, 180e6, Nf)
    zen, az = random_dirs(rng, P, zen_max=89.0)
    antpos = ba.utils.AntposDict(list(range(Nant)), ant)
    arr = ba.telescope_model.ArrayModel(antpos, freqs=torch.as_tensor(freqs), cache_s=False)
    beam = torch.as_tensor(np.exp(-0.5 * (zen / 25.0) ** 2)[None, :] * (1.0 + 0.05 * rng.normal(size=(Nf, P))))
    A = arr.gen_fringe(torch.as_tensor(blvecs), torch.as_tensor(zen), torch.as_tensor(az), conj=True) * beam
    Nbl = len(pairs)
    v = torch.as_tensor(rng.normal(size=(2, Nbl, Nf)) + 1j * rng.normal(size=(2, Nbl, Nf)))     # two maps
    w = torch.as_tensor(rng.uniform(0.2, 2.0, size=(Nbl, Nf)))
    m = torch.as_tensor(rng.normal(size=(2, Nf, P)))
    D = torch.as_tensor(rng.uniform(0.5, 1.5, size=(Nf, P)))
    save('imaging', antpos=ant, pairs=np.asarray(pairs), blvecs=blvecs, freqs=freqs, zen=zen, az=az, beam=beam,
         v=v, w=w, m=m, D=D, A_checksum=torch.stack([A.real.sum(), A.imag.sum(), (A.abs() ** 2).sum()]),
         dirty=imaging.make_map(v, w, A), Am=imaging.compute_Am(A, m.to(A.dtype)),   # (a real m raises in einsum)
         Pm=imaging.compute_Pm(A, w, m, D=D))


def gen_vismapper(ba):
    """imaging.VisMapper (imaging.py:12-714) end to end: hex-37 (666 baselines), 5 channels, 3 times, 160 map pixels
    of which part set below the horizon, an Airy PixelBeam, random visibilities and weights; dirty maps + PSF
    contractions for the three normalisations, a channel / time / baseline sub-selection, compute_Am / compute_Pm /
    compute_P.  The telescope conversion cache of the mapper's own (re-instantiated) telescope is filled with the
    LST-rotation angles, as for the RIME fixtures."""
    import importlib
    imaging = importlib.import_module('bayeslim.imaging')
    rng = np.random.default_rng(21)
    freqs = torch.linspace(120e6, 160e6, 5)
    times = np.array([2459861.0 + k * 40.0 / 1440 for k in range(3)])
    arr = hex_array(ba, 4, freqs)
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    Nbl, Npix = len(bls), 160
    ra = lst_of(times[1]) + rng.uniform(-75, 75, Npix)
    dec = LAT + rng.uniform(-70, 70, Npix)
    beam, theta_grid, phi_grid = airy_pixbeam(ba, freqs, parameter=False)
    data = torch.as_tensor(rng.normal(size=(1, 1, Nbl, 3, 5)) + 1j * rng.normal(size=(1, 1, Nbl, 3, 5)))
    icov = torch.as_tensor(rng.uniform(0.2, 2.0, size=(1, 1, Nbl, 3, 5)))
    vd = ba.dataset.VisData()
    vd.setup_meta(tel, arr.to_antpos())
    vd.setup_data(bls, torch.as_tensor(times), freqs, pol='ee', data=data, icov=icov)

    def mapper(beam_obj, **sel):
        vm = imaging.VisMapper(vd, ra, dec, beam=beam_obj, fov=180)
        for t in times:
            zen, az = radec_to_zenaz(ra, dec, lst_of(float(t)), LAT)
            vm.telescope.conv_cache[(float(t), Npix)] = torch.stack([torch.as_tensor(zen), torch.as_tensor(az)])
        return vm

    zenaz = np.stack([np.stack(radec_to_zenaz(ra, dec, lst_of(float(t)), LAT)) for t in times])
    out = dict(freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants), bls=np.array(bls), ra=ra, dec=dec,
               zenaz=zenaz, data=data, icov=icov, beam_params=beam.params.detach(), theta_grid=theta_grid, phi_grid=phi_grid)
    vm = mapper(beam)
    for method in ('A2w', 'Aw', 'w'):
        vm.set_normalization(method)
        maps, P = vm.make_map(return_P=True, contract='diag')
        out['maps_' + method], out['Pdiag_' + method], out['D_' + method] = maps, P, vm.D
    vm.set_normalization('A2w')
    _, out['Prowsum'] = vm.make_map(return_P=True, contract='rowsum')
    m_in = torch.as_tensor(rng.normal(size=(2, 5, Npix)))
    out['m_in'] = m_in
    out['Am'] = vm.compute_Am(m_in)
    out['Pm'] = vm.compute_Pm(m_in)
    out['P_cdiag'] = vm.compute_P(contract='diag')
    # a list of two VisData imaged together
    vd2 = vd.copy(copydata=True)
    vd2.data = vd2.data * (0.5 - 0.25j)
    out['maps_list'], _ = vm.make_map(vd=[vd, vd2], return_P=False)
    # sub-selection: channels 1..3, times 0 and 2, every third baseline, no beam, fov 120, full PSF matrix on a
    # 40-pixel map
    sub = rng.choice(Npix, 40, replace=False); sub.sort()
    vms = imaging.VisMapper(vd, ra[sub], dec[sub], beam=None, fov=120)
    for t in times:
        vms.telescope.conv_cache[(float(t), 40)] = torch.as_tensor(zenaz[list(times).index(t)][:, sub])
    vms.set_freq_inds(freq_inds=[1, 2, 3])
    vms.set_time_inds(time_inds=[0, 2])
    vms.set_bl_inds(bl_inds=list(range(0, Nbl, 3)))
    vms.set_normalization('A2w', icov=torch.ones_like(icov) * 0.7)
    out['sub_pix'] = sub
    out['sub_maps'], out['sub_Pfull'] = vms.make_map(return_P=True, contract=None)
    out['sub_Am'] = vms.compute_Am(m_in[0][1:4][:, sub])
    A0, cut0 = vm.build_A(times[0])
    out['A0_checksum'] = torch.stack([A0.real.sum(), A0.imag.sum(), (A0.abs() ** 2).sum()])
    out['cut0'] = cut0
    save('vismapper', **out)


def gen_logprob(ba):
    """optim.LogProb (optim.py:385-1389) driving RIME: hex-37 (666 baselines), diffuse pixel sky + Airy PixelBeam, 4
    channels, 4 times in two minibatches; target data with per-visibility inverse covariance; Gaussian priors attached to
    the sky (input params) and the beam; closure() loss and accumulated gradients (a) with the modules' own Parameters,
    (b) through set_main_params with an indexed sky piece + the whole beam (non-leaf graph tensors on the modules), and
    (c) grad_type 'stochastic' on batch 1, compute 'like' / 'prior', negate False, complex_circular False."""
    import importlib
    optim = importlib.import_module('bayeslim.optim')
    Nf, Npix = 4, 400
    freqs = torch.linspace(120e6, 180e6, Nf)
    times = 2459861.0 + np.arange(4) * 10.0 / 1440
    rng = np.random.default_rng(31)
    arr = hex_array(ba, 4, freqs)
    tel = ba.telescope_model.TelescopeModel((21.42827, LAT))
    ra, dec = fib_sky(Npix)
    px_area = 4 * np.pi / Npix
    Rs = ba.sky_model.PixelSkyResponse(freqs, cosmo=object())
    sp = torch.as_tensor(rng.normal(size=(1, 1, Nf, len(ra))))
    sky = ba.sky_model.PixelSky(sp.clone(), torch.stack([ra, dec]), px_area, R=Rs, parameter=True, name='pixsky')
    beam, tg, pg = airy_pixbeam(ba, freqs, parameter=True)
    bp = beam.params.detach().clone()
    ants = arr.ants
    sim_bls = [(ants[i], ants[j]) for i in range(len(ants)) for j in range(i + 1, len(ants))]
    groups = ba.utils.split_into_groups(times, Nelem=2)
    rime = ba.rime_model.RIME(sky, tel, beam, arr, sim_bls, groups, freqs)
    zenaz = fill_eq2top(tel, sky.name, ra, dec, times)
    # priors on the modules
    sky_mean, sky_var = torch.as_tensor(rng.normal(size=sp.shape) * 0.3), torch.as_tensor(rng.uniform(0.5, 2.0, size=sp.shape))
    beam_var = torch.as_tensor(rng.uniform(0.01, 0.05, size=bp.shape))
    sky.set_priors(priors_inp_params=[optim.LogGaussPrior(sky_mean, sky_var)])
    beam.set_priors(priors_inp_params=[optim.LogGaussPrior(bp - 0.01, beam_var, side='upper'),
                                       optim.LogUniformPrior(-1.0, 2.0, index=(0, 0, 0, slice(0, 2)))])
    model = ba.utils.Sequential(dict(rime=rime))
    # target: two VisData minibatches with noise-like offsets from the model and a random diagonal icov
    targets, dvis, dicov = [], [], []
    with torch.no_grad():
        for i in range(2):
            model.batch_idx = i
            v = model().data
            d = v + torch.as_tensor(rng.normal(size=tuple(v.shape)) + 1j * rng.normal(size=tuple(v.shape))) * 0.05 * v.abs().mean()
            ic = torch.as_tensor(rng.uniform(0.5, 2.0, size=tuple(v.shape))) / (0.05 * v.abs().mean()) ** 2
            vd = ba.dataset.VisData()
            vd.setup_meta(tel, arr.to_antpos())
            vd.setup_data(sim_bls, torch.as_tensor(groups[i]), freqs, pol='ee', data=d, icov=ic)
            targets.append(vd); dvis.append(d); dicov.append(ic)
    model.batch_idx = 0
    target = ba.dataset.Dataset(targets)
    out = dict(freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(arr.ants), sim_bls=np.array(sim_bls), ra=ra, dec=dec,
               zenaz=zenaz, px_area=np.array(px_area), sky_params=sp, beam_params=bp, theta_grid=tg, phi_grid=pg,
               data0=dvis[0], data1=dvis[1], icov0=dicov[0], icov1=dicov[1], sky_mean=sky_mean, sky_var=sky_var,
               beam_var=beam_var)
    prob = optim.LogProb(model, target)
    out['loss_a'] = prob.closure()
    out['g_sky_a'], out['g_beam_a'] = sky.params.grad.clone(), beam.params.grad.clone()
    chisq0, _ = prob.forward_chisq(0)
    out['chisq0'] = chisq0.detach()
    out['like1'] = prob.forward_like(1).detach()
    prob.batch_idx = 0
    prob.clear_prior_cache()
    prob.forward_chisq(0)                      # fills the prior cache as the forward pass does
    out['prior'] = prob.forward_prior().detach()
    # (b) main params: pixels 10..59 of channels 1 and 3 of the sky in two pieces + the whole beam
    pieces = [(0, 0, 1, range(10, 60)), (0, 0, 3, range(10, 60))]
    prob.set_main_params([('rime.sky.params', pieces, 'sky'), ('rime.beam.params', None, 'beam')])
    assert not model['rime.sky.params'].is_leaf
    out['main0'] = prob.main_params.detach().clone()
    out['loss_b'] = prob.closure()
    out['g_main_b'] = prob.main_params.grad.clone()
    # a step along -grad, then re-evaluate: the modules must see the new values
    with torch.no_grad():
        prob.main_params -= 1e-9 * prob.main_params.grad / prob.main_params.grad.abs().max() * 1e6
    out['main1'] = prob.main_params.detach().clone()
    out['loss_b2'] = prob.closure()
    prob.set_main_params(None)
    assert model['rime.sky.params'].is_leaf
    out['sky_after_main'] = model['rime.sky.params'].detach().clone()
    # (c) variants
    prob.grad_type = 'stochastic'
    prob.batch_idx = 1
    out['loss_c_stoch1'] = prob.closure()
    out['g_sky_c_stoch1'] = sky.params.grad.clone()
    prob.grad_type = 'accumulate'
    prob.compute = 'like'
    out['loss_c_like'] = prob.closure()
    prob.compute = 'prior'
    out['loss_c_prior'] = prob.closure()
    out['g_beam_c_prior'], out['g_sky_c_prior'] = beam.params.grad.clone(), sky.params.grad.clone()
    prob.compute, prob.negate, prob.complex_circular = 'post', False, False
    out['loss_c_pos_real'] = prob.closure()
    out['g_sky_c_pos_real'] = sky.params.grad.clone()
    save('logprob', **out)


def load_calibration_module(ba):
    """bayeslim.calibration does not parse under Python 3.10 (one py>=3.11 subscript, `A[*self.idx]`, at :2279 in a class
    unrelated to the path); the module is executed from its own source, read from the reference checkout at generation
    time, with that token spelled `A[tuple(self.idx)]`."""
    import importlib
    importlib.import_module('bayeslim.optim')
    src = open(os.path.join(REF, 'calibration.py')).read().replace('[*self.idx]', '[tuple(self.idx)]')
    mod = types.ModuleType('bayeslim.calibration')
    mod.__package__ = 'bayeslim'
    mod.__file__ = os.path.join(REF, 'calibration.py')
    sys.modules['bayeslim.calibration'] = mod
    exec(compile(src, mod.__file__, 'exec'), mod.__dict__)
    return mod


class AxisLM:
    """a linear basis along one axis (the protocol JonesResponse expects of freq_LM / time_LM: callable + push)"""
    def __init__(self, A, axis):
        self.A, self.axis = A, axis

    def __call__(self, p):
        return torch.movedim(torch.movedim(p, self.axis, -1) @ self.A.to(p.dtype).T, -1, self.axis)

    def push(self, device):
        self.A = self.A.to(device)


def gen_jones(ba):
    """calibration.JonesModel / JonesResponse (calibration.py:416-875) applied to a VisData: every gain type, the
    reference-antenna conventions, p0, 1 / 2 / 4-pol, single_ant, undo, a time minibatch through the index cache, linear
    bases over frequency and time; outputs and gradients w.r.t. the parameters"""
    cal = load_calibration_module(ba)
    rng = np.random.default_rng(41)
    freqs = torch.linspace(120e6, 180e6, 6)
    times = torch.as_tensor(2459861.0 + np.arange(4) * 10.0 / 1440)
    arr = hex_array(ba, 2, freqs)
    ants = arr.ants
    bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    Nant, Nbl, Nt, Nf = len(ants), len(bls), 4, 6
    antpos = arr.to_antpos()

    def rc(*shape):
        return torch.as_tensor(rng.normal(size=shape) + 1j * rng.normal(size=shape))

    def visdata(npol, tsel=slice(None)):
        vd = ba.dataset.VisData()
        vd.setup_meta(None, antpos)
        data = rc(npol, npol, Nbl, Nt, Nf)
        vd.setup_data(bls, times[tsel], freqs, pol='ee' if npol == 1 else None, data=data[:, :, :, tsel])
        return vd, data

    out = dict(freqs=freqs, times=times, antvecs=arr.antvecs, ants=np.array(ants), bls=np.array(bls))
    # the reference's time index cache needs the response to know the time axis (calibration.py:327-336)
    JR = lambda **kw: cal.JonesResponse(times=times, **kw)

    def run(tag, jm, vd, **fw):
        vout = jm(vd, **fw)
        cot = rc(*vout.data.shape) if vout.data.is_complex() else torch.as_tensor(rng.normal(size=tuple(vout.data.shape)))
        if jm.params.requires_grad:
            ((vout.data * cot.conj()).real.sum() if vout.data.is_complex() else (vout.data * cot).sum()).backward()
            out['g_' + tag] = jm.params.grad.clone()
            jm.params.grad = None
        out['vout_' + tag], out['cot_' + tag], out['params_after_' + tag] = vout.data.detach(), cot, jm.params.detach().clone()

    vd1, d1 = visdata(1)
    out['vis1'] = d1
    # complex gains, reference antenna (rephase mode), p0
    p = rc(1, 1, Nant, Nt, Nf); p0 = 0.1 * rc(1, 1, Nant, Nt, Nf)
    out['p_com'], out['p0_com'] = p.clone(), p0.clone()
    run('com', cal.JonesModel(p.clone(), ants, p0=p0.clone(), refant=ants[2], R=JR(param_type='com')), vd1)
    # real-view complex parameters
    pr = torch.view_as_real(rc(1, 1, Nant, Nt, Nf)).clone()
    out['p_comreal'] = pr.clone()
    run('comreal', cal.JonesModel(pr.clone(), ants, refant=ants[0], R=JR(param_type='com')), vd1)
    for ptype in ('amp', 'phs', 'real'):
        pp = torch.as_tensor(rng.normal(size=(1, 1, Nant, Nt, Nf)) * 0.3)
        out['p_' + ptype] = pp.clone()
        run(ptype, cal.JonesModel(pp.clone(), ants, refant=ants[1] if ptype == 'phs' else None,
                                  R=JR(param_type=ptype)), vd1)
    pap = torch.as_tensor(rng.normal(size=(1, 1, Nant, Nt, Nf, 2)) * 0.3)
    out['p_amp_phs'] = pap.clone()
    run('amp_phs', cal.JonesModel(pap.clone(), ants, refant=ants[3], R=JR(param_type='amp_phs')), vd1)
    # delays [ns] on complex visibilities, one delay per antenna and time
    pd = torch.as_tensor(rng.normal(size=(1, 1, Nant, Nt, 1)) * 5.0)
    out['p_dly'] = pd.clone()
    run('dly', cal.JonesModel(pd.clone(), ants, refant=ants[0], R=JR(param_type='dly', freqs=freqs)), vd1)
    # phase / delay gradients over the array: antenna axis = (EW, NS)
    for ptype, scale in (('phs_slope', 0.02), ('dly_slope', 0.05)):
        ps = torch.as_tensor(rng.normal(size=(1, 1, 2, Nt, 1)) * scale)
        out['p_' + ptype] = ps.clone()
        run(ptype, cal.JonesModel(ps.clone(), ants, R=JR(param_type=ptype, antpos=antpos, freqs=freqs)), vd1)
    # one gain for the whole array; undo
    pg = rc(1, 1, 1, Nt, Nf)
    out['p_single'] = pg.clone()
    run('single', cal.JonesModel(pg.clone(), ants, single_ant=True, R=JR()), vd1)
    run('undo', cal.JonesModel(p.clone(), ants, R=JR()), vd1, undo=True)
    out['p_undo'] = p.clone()
    # time minibatch: the response knows all times, the VisData holds times 1 and 3
    vdt, _ = visdata(1, tsel=[1, 3])
    out['vis_tsel'] = vdt.data.clone()
    run('tsel', cal.JonesModel(p.clone(), ants, R=JR(param_type='com')), vdt)
    # polynomial bases: 3 coefficients over frequency, 2 over time ('zero' reference-antenna mode)
    Af = torch.as_tensor(np.vander(np.linspace(-1, 1, Nf), 3, increasing=True))
    At = torch.as_tensor(np.vander(np.linspace(-1, 1, Nt), 2, increasing=True))
    out['Af'], out['At'] = Af, At
    pl = rc(1, 1, Nant, 2, 3)
    out['p_linear'] = pl.clone()
    run('linear', cal.JonesModel(pl.clone(), ants, refant=ants[2],
                                 R=JR(param_type='com', freq_mode='linear', time_mode='linear',
                                                     freq_LM=AxisLM(Af, -1), time_LM=AxisLM(At, -2))), vd1)
    # 2-pol (diagonal) and 4-pol
    vd2, d2 = visdata(2)
    out['vis2'] = d2
    p2 = rc(2, 2, Nant, Nt, Nf); p2[0, 1] = 0; p2[1, 0] = 0
    out['p_2pol'] = p2.clone()
    run('2pol', cal.JonesModel(p2.clone(), ants, polmode='2pol', refant=ants[0], R=JR()), vd2)
    p4 = rc(2, 2, Nant, Nt, Nf)
    out['p_4pol'] = p4.clone()
    run('4pol', cal.JonesModel(p4.clone(), ants, polmode='4pol', R=JR()), vd2)
    save('jones', **out)


def gen_apply_cal(ba):
    """gain application G_p V G_q^dagger of calibration._apply_cal (calibration.py:2412-2487), 'com'
    visibilities, 1-pol / 2-pol (diagonal) / 4-pol, with gradients w.r.t. visibilities and gains.
    calibration.py does not parse under Python 3.10 (a py>=3.11 construct at :2279), so the function
    is executed from its own source lines, read from the reference checkout at generation time."""
    src = open(os.path.join(REF, 'calibration.py')).read().split('\n')
    i0 = next(i for i, l in enumerate(src) if l.startswith('def _apply_cal('))
    i1 = next(i for i in range(i0 + 1, len(src)) if src[i].startswith('def '))
    ns = {'torch': torch, 'linalg': ba.linalg, 'np': np}
    exec(compile('\n'.join(src[i0:i1]), 'calibration.py:_apply_cal', 'exec'), ns)
    apply_cal = ns['_apply_cal']
    rng = np.random.default_rng(13)
    Nant, Nt, Nf = 6, 3, 4
    pairs = [(i, j) for i in range(Nant) for j in range(i, Nant)]
    pairs = pairs[::2] + [(3, 1), (5, 0)]
    g1_idx = torch.as_tensor([p[0] for p in pairs])
    g2_idx = torch.as_tensor([p[1] for p in pairs])
    Nbl = len(pairs)

    def rc(*shape):
        return torch.as_tensor(rng.normal(size=shape) + 1j * rng.normal(size=shape))

    out = dict(g1_idx=g1_idx, g2_idx=g2_idx)
    for tag, npol, two, gshape in [('1pol', 1, False, (1, 1, Nant, Nt, Nf)), ('1pol_bcast', 1, False, (1, 1, Nant, 1, Nf)),
                                    ('2pol', 2, True, (2, 2, Nant, Nt, Nf)), ('4pol', 2, False, (2, 2, Nant, Nt, Nf))]:
        vis = rc(npol, npol, Nbl, Nt, Nf).requires_grad_(True)
        gains = rc(*gshape)
        if two:                                   # diagonal gains: off-diagonals are dropped by diag_matmul
            gains[0, 1] = 0
            gains[1, 0] = 0
        gains.requires_grad_(True)
        vout, _ = apply_cal(vis, gains, g1_idx, g2_idx, cal_2pol=two)
        cot = rc(*vout.shape)
        (vout * cot.conj()).real.sum().backward()
        out.update({'vis_' + tag: vis, 'gains_' + tag: gains, 'vout_' + tag: vout, 'cot_' + tag: cot,
                    'gvis_' + tag: vis.grad, 'ggains_' + tag: gains.grad})
    # remaining branches of the function: undo (gain inversion), covariance propagation, delay-type visibilities
    for tag, npol, two in [('1pol', 1, False), ('2pol', 2, True)]:
        vis = rc(npol, npol, Nbl, Nt, Nf).requires_grad_(True)
        gains = rc(npol, npol, Nant, Nt, Nf)
        if two:
            gains[0, 1] = 0
            gains[1, 0] = 0
        gains.requires_grad_(True)
        cov = torch.as_tensor(np.abs(rng.normal(size=(npol, npol, Nbl, Nt, Nf))) + 0.1)
        vout, cout = apply_cal(vis, gains, g1_idx, g2_idx, cal_2pol=two, cov=cov, undo=True)
        cot = rc(*vout.shape)
        (vout * cot.conj()).real.sum().backward()
        out.update({'u_vis_' + tag: vis, 'u_gains_' + tag: gains, 'u_cov_' + tag: cov, 'u_vout_' + tag: vout,
                    'u_cout_' + tag: cout, 'u_cot_' + tag: cot, 'u_gvis_' + tag: vis.grad, 'u_ggains_' + tag: gains.grad})
    dvis = torch.as_tensor(rng.normal(size=(1, 1, Nbl, Nt, Nf))).requires_grad_(True)
    dg = torch.as_tensor(rng.normal(size=(1, 1, Nant, Nt, Nf))).requires_grad_(True)
    for undo in (False, True):
        dout, _ = apply_cal(dvis, dg, g1_idx, g2_idx, vis_type='dly', undo=undo)
        out['dly_vout_undo%d' % undo] = dout
    out.update(dly_vis=dvis, dly_gains=dg)
    save('apply_cal', **out)


def main():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(0)
    ba = bootstrap_reference()
    if len(sys.argv) > 1:                       # e.g. `make_golden.py gen_chisq`: regenerate selected files only
        for name in sys.argv[1:]:
            globals()[name](ba)
        return
    gen_chisq(ba)
    gen_imaging(ba)
    gen_vismapper(ba)
    gen_logprob(ba)
    gen_jones(ba)
    gen_apply_cal(ba)
    gen_fringe_cases(ba)
    gen_apply_beam_cases(ba)
    gen_interp_cases(ba)
    gen_sph_cases(ba)
    gen_sky_beam_response_cases(ba)
    gen_prod_and_sum(ba)
    gen_rime_c1(ba)
    gen_rime_c2_mini(ba)
    gen_rime_c3_mini(ba)
    gen_rime_c5_mini(ba)
    gen_rime_mfma_arrays(ba)
    gen_rime_mfma_large(ba)
    gen_rime_pol_mfma(ba)
    gen_rime_composite(ba)
    gen_rime_two_models(ba)


if __name__ == '__main__':
    main()
