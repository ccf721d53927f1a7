"""
CPU tests of the host-side mirror of the reference interface (no kernels are launched):
array / redundancy bookkeeping, minibatch protocol, interpolation stencils, HEALPix helpers,
Ylm generation, module attribute protocol.  Values are checked against the golden vectors or
against the assertions of the reference's own tests (cited).
"""
import copy
import pickle

import numpy as np
import pytest
import torch

from conftest import load_golden
import bayeslim_amd as ba
from bayeslim_amd import utils, telescope_model, beam_model, sky_model, sph_harm, healpix, rime_model
from oracle import rime_oracle as orc


@pytest.fixture(autouse=True)
def _f64():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def hex_array(N, freqs=None, D=15):
    ants, vecs = utils._make_hex(N, D=D)
    return telescope_model.ArrayModel(utils.AntposDict(ants, vecs), freqs=freqs, cache_s=False, redtol=1.0)


def test_make_hex_matches_reference_layout():
    for name, N in [('rime_c1', 2), ('rime_c2_mini', 3)]:
        g = load_golden(name)
        ants, vecs = utils._make_hex(N, D=14.6)
        assert ants == g['ants'].tolist()
        assert np.abs(vecs - g['antvecs']).max() < 1e-12


def test_array_redundancy_counts_and_blvecs():
    """tests/test_telescope.py:41-51, 78-83"""
    arr = hex_array(3)
    assert len(arr.ants) == 19 and len(arr.reds) == 31
    blv = arr.get_antpos(1) - arr.get_antpos(0)
    assert (blv - torch.tensor([15., 0, 0])).norm() < 1e-10
    sim = arr.get_bls(uniq_bls=True, keep_autos=True, min_len=1, max_len=29)
    assert (0, 0) not in sim and (0, 2) not in sim and (1, 2) not in sim
    assert len(arr.get_bls(uniq_bls=False, keep_autos=False)) == 171
    assert len(arr.get_bls(uniq_bls=True, keep_autos=False)) == 30
    g = load_golden('rime_c2_mini')
    assert [tuple(b) for b in g['sim_bls']] == hex_array(3, D=14.6).get_bls(uniq_bls=False, keep_autos=False)
    assert [tuple(b) for b in g['uniq_bls']] == hex_array(3, D=14.6).get_bls(uniq_bls=True, keep_autos=False)


class _DummySky(utils.Module):
    def __init__(self):
        super().__init__(name='dummy')
        self.params = torch.nn.Parameter(torch.ones(1, 1, 4, 3))
        self.device = 'cpu'


def _rime_cpu(times, sim_bls=None, data_bls=None, N=3):
    freqs = torch.linspace(120e6, 130e6, 4)
    arr = hex_array(N, freqs, D=14.6)
    tel = telescope_model.TelescopeModel((21.42827, -30.72148))
    beam = beam_model.PixelBeam(torch.ones(1, 1, 1, 4, 1), freqs, parameter=False, pol='e')
    if sim_bls is None:
        sim_bls = arr.get_bls(uniq_bls=True, keep_autos=False)
    return rime_model.RIME(_DummySky(), tel, beam, arr, sim_bls, times, freqs, data_bls=data_bls), arr


def test_rime_batch_protocol():
    """rime_model.py:253-289 and tests/test_rime.py:41-47"""
    times = torch.linspace(2459861, 2459862, 5)
    rime, arr = _rime_cpu(times)
    assert rime.Nbatch == 1 and rime.batch_idx == 0 and rime.Ntimes == 5
    groups = utils.split_into_groups(times, Nelem=2)
    rime.setup_sim_times(groups)
    assert rime.Nbatch == int(np.ceil(5 / 2))
    bls = arr.get_bls(uniq_bls=True, keep_autos=False)
    rime.setup_sim_bls([bls[:10], bls[10:]])
    assert rime.Nbatch == 6
    order = []
    for i in range(rime.Nbatch):
        rime.batch_idx = i
        order.append((rime.time_group_id, rime.bl_group_id))
        assert rime.batch_idx == i
    assert order == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)]    # baselines fastest
    assert rime.Nsim_bls == 20 and rime.Ntimes == 1
    with pytest.raises(AssertionError):
        rime.batch_idx = 6
    assert rime.Ntimes_all == 5 and rime.Nbls_all == 30


def test_rime_sim2data_matches_reference():
    g = load_golden('rime_c2_mini')
    uniq = [tuple(b) for b in g['uniq_bls']]
    rime, arr = _rime_cpu(np.array([2459861.0]), sim_bls=uniq,
                          data_bls=arr_all_bls(g))
    assert rime._sim2data[0].tolist() == g['sim2data'].tolist()
    assert rime.data_bls == [tuple(b) for b in g['data_bls']]
    assert rime.Ndata_bls == 171 and rime.Nsim_bls == 30


def arr_all_bls(g):
    return [tuple(b) for b in g['sim_bls']]


def test_rime_refuses_cpu_forward():
    rime, _ = _rime_cpu(np.array([2459861.0]))

    class Sky(utils.Module):
        device = 'cpu'
        name_ = 's'

        def forward(self, prior_cache=None):
            from bayeslim_amd.dataset import MapData
            m = MapData()
            m.setup_meta(name='s')
            m.setup_data(freqs=None, data=torch.ones(1, 1, 4, 3), angs=torch.zeros(2, 3))
            return m
    rime.sky = Sky()
    with pytest.raises(RuntimeError, match='GPU only'):
        rime()


def test_bipoly_weights_match_golden():
    g = load_golden('interp_rect')
    tg, pg = torch.as_tensor(g['theta_grid']), torch.as_tensor(g['phi_grid'])
    zen, az = torch.as_tensor(g['zen']), torch.as_tensor(g['az'])
    Npb = len(tg) * len(pg)
    for mode, deg in [('nearest', (0, 0)), ('linear', (1, 1)), ('quadratic', (2, 2)), ('cubic', (3, 3)),
                      ('linear_quadratic', (1, 2))]:
        inds, w = utils.bipoly_interp_weights(pg, tg, az, zen, deg)
        for r in range(len(zen)):
            d1, d2 = np.zeros(Npb), np.zeros(Npb)
            np.add.at(d1, inds[r].numpy(), w[r].numpy())
            np.add.at(d2, g[mode + '__inds'][r], g[mode + '__wgts'][r])
            assert np.abs(d1 - d2).max() < 2e-9, (mode, r)
        same = (inds.numpy() == g[mode + '__inds']).all(1)
        assert same.sum() >= len(same) - 1


def test_pixinterp_needs_gpu():
    PI = utils.PixInterp('rect', interp_mode='linear', theta_grid=torch.arange(0, 91.0),
                         phi_grid=torch.arange(0, 360.0))
    with pytest.raises(RuntimeError, match='GPU'):
        PI.get_interp(torch.tensor([10.0]), torch.tensor([20.0]))


def test_healpix_self_consistency():
    """parity unpinned (healpy absent): structural checks"""
    for nside in (1, 2, 4, 16):
        th, ph = healpix.pix2ang(nside)
        th_o, ph_o = orc.healpix_pix2ang(nside)
        assert np.abs(th - th_o).max() < 1e-14 and np.abs(ph - ph_o).max() < 1e-14
        npix = healpix.nside2npix(nside)
        assert abs(healpix.nside2pixarea(nside) * npix - 4 * np.pi) < 1e-12
        # interpolation: weights sum to one, are non-negative, and are exact at pixel centres
        pix, w = healpix.get_interp_weights(nside, th, ph)
        assert pix.shape == (4, npix) and np.abs(w.sum(0) - 1).max() < 1e-12 and w.min() > -1e-12
        got = (w * (pix == np.arange(npix)[None])).sum(0)
        assert np.abs(got - 1).max() < 1e-9
        rng = np.random.default_rng(nside)
        t, p = np.arccos(rng.uniform(-1, 1, 500)), rng.uniform(0, 2 * np.pi, 500)
        pix, w = healpix.get_interp_weights(nside, t, p)
        assert pix.min() >= 0 and pix.max() < npix and np.abs(w.sum(0) - 1).max() < 1e-12
    # a smooth function is reproduced to O(pixel size^2)
    nside = 32
    th, ph = healpix.pix2ang(nside)
    f = lambda a, b: np.cos(a) + 0.3 * np.sin(a) * np.cos(b)
    pix, w = healpix.get_interp_weights(nside, t, p)
    assert np.abs((w * f(th, ph)[pix]).sum(0) - f(t, p)).max() < 3e-3


def test_gen_sph2pix_matches_golden():
    g = load_golden('sph_harm')
    lm = sph_harm.gen_lm(int(g['lmax']))
    assert (lm[0] == g['l']).all() and (lm[1] == g['m']).all()
    for real, key in [(False, 'comp'), (True, 'real')]:
        Y, norm, mult = sph_harm.gen_sph2pix(g['theta'] * utils.D2R, g['phi'] * utils.D2R, lm[0], lm[1],
                                             real=real)
        assert np.abs(Y.numpy() - g['Ylm_' + key]).max() < 1e-13
        assert (mult.numpy() == g['alm_mult_' + key]).all()
    (T, P), _, mult = sph_harm.gen_sph2pix(g['theta_grid'] * utils.D2R, g['phi_grid'] * utils.D2R,
                                           lm[0], lm[1], separable=True)
    assert np.abs(T.numpy() - g['Theta']).max() < 1e-13 and np.abs(P.numpy() - g['Phi']).max() < 1e-13
    full = sph_harm.inflate_Ylm((T, P))
    assert full.shape == (len(lm[0]), len(g['theta_grid']) * len(g['phi_grid']))


def test_module_attr_protocol_and_pickle():
    """utils.py:1159-1167, 1453-1545 (what optim.LogProb.set_main_params relies on)"""
    freqs = torch.linspace(120e6, 130e6, 4)
    sky = sky_model.PointSky(torch.ones(1, 1, 2, 5), torch.zeros(2, 5),
                             R=sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs[0]))
    top = utils.Sequential({'sky': sky})
    assert isinstance(top['sky.params'], torch.nn.Parameter)
    top['sky.params'] = torch.full((1, 1, 2, 5), 2.0)
    assert isinstance(sky.params, torch.nn.Parameter) and float(sky.params.detach().sum()) == 20.0
    # replace by a non-leaf graph tensor, as LogProb.set_main_params does
    main = torch.nn.Parameter(torch.ones(10))
    utils.set_model_attr(top, 'sky.params', (main * 3).reshape(1, 1, 2, 5), clobber_param=True, no_grad=False)
    assert not sky.params.is_leaf and type(sky.params) is torch.Tensor
    out = sky()
    out.data.sum().backward()
    assert main.grad is not None
    del top['sky.params']
    assert not hasattr(sky, 'params')
    sky.params = torch.nn.Parameter(torch.ones(1, 1, 2, 5))
    sky2 = pickle.loads(pickle.dumps(sky))
    assert torch.equal(sky2.params, sky.params)
    copy.deepcopy(sky)


def test_rime_and_almmodel_leave_their_derived_caches_behind_when_copied():
    """host side of tests/test_objects_gpu.py: RIME.__getstate__ / AlmModel.__getstate__ -- the derived caches hold ctypes
    tables, device buffers and entries keyed on addresses of the ORIGINAL's tensors; pickle and deepcopy must succeed with
    such (unpicklable) content attached and hand over a model whose caches are empty (io.py:50-66, optim.py:1517-1523)"""
    import ctypes
    from bayeslim_amd import rime_model, telescope_model, beam_model, sph_harm
    freqs = torch.linspace(120e6, 130e6, 4)
    ants, vecs = utils._make_hex(2, D=14.6)
    arr = telescope_model.ArrayModel(utils.AntposDict(ants, vecs), freqs=freqs)
    tel = telescope_model.TelescopeModel((21.42827, -30.72148))
    sky = sky_model.PointSky(torch.ones(1, 1, 2, 5), torch.zeros(2, 5),
                             R=sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs[0]), name='pts')
    beam = beam_model.PixelBeam(torch.ones(1, 1, 1, 1, 1) * 14.0, freqs, R=beam_model.AiryResponse(powerbeam=True), pol='e',
                                powerbeam=True, fov=180, parameter=False)
    bls = arr.get_bls(uniq_bls=False, keep_autos=False)
    rime = rime_model.RIME(sky, tel, beam, arr, bls, np.array([2459861.0, 2459861.01]), freqs)
    unpicklable = ctypes.pointer(ctypes.c_int(3))
    with pytest.raises(Exception):
        pickle.dumps(unpicklable)
    rime._geom_cache[('k', 1)] = dict(geom=unpicklable)
    rime._ant_like[(0, 1)] = unpicklable
    rime._inflate_cache = {0: (None, unpicklable)}
    rime._blnum_cache = {0: (rime.data_bls, np.arange(3))}
    rime._zenaz_cache[('pts', 5, 1.0)] = (None, (torch.zeros(5), torch.zeros(5)))
    for cl in (pickle.loads(pickle.dumps(rime, protocol=4)), copy.deepcopy(rime)):
        for k in rime_model.RIME._DERIVED:
            assert getattr(cl, k) == {}, k
        assert cl.sim_bls == rime.sim_bls and cl.Nbatch == rime.Nbatch and torch.equal(cl.sim_blvecs, rime.sim_blvecs)
        assert cl.sky is not rime.sky and torch.equal(cl.sky.params, rime.sky.params)
    assert len(rime._geom_cache) == 1 and rime._ant_like[(0, 1)] is unpicklable          # the original keeps its caches
    lm = sph_harm.gen_lm(4)
    A = sph_harm.AlmModel(lm[0], lm[1], real_output=True)
    A.setup_Ylm(np.linspace(10, 80, 7), np.linspace(0, 300, 7), generate=True)
    A._Ylm_cast_cache = {id(A.Ylm): (A.Ylm, 0, unpicklable)}
    A._Ylm_pack_cache = {id(A.Ylm): (A.Ylm, 0, unpicklable)}
    A._inflated, A._inflated_key = unpicklable, 123
    for B in (pickle.loads(pickle.dumps(A, protocol=4)), copy.deepcopy(A)):
        assert not hasattr(B, '_Ylm_cast_cache') and not hasattr(B, '_Ylm_pack_cache') and B._inflated_key is None
        assert torch.equal(B.Ylm, A.Ylm) and len(B.Ylm_cache) == 1
    assert A._inflated is unpicklable


def test_sky_and_beam_responses_cpu_pieces():
    """elementwise responses are plain torch ops and may be evaluated anywhere"""
    g = load_golden('responses')
    freqs, zen, az = torch.as_tensor(g['freqs']), torch.as_tensor(g['zen']), torch.as_tensor(g['az'])
    R = sky_model.PointSkyResponse(freqs, freq_mode='powerlaw', f0=freqs[0])
    assert np.abs(R(torch.as_tensor(g['point_params'])).numpy() - g['point_powerlaw']).max() < 1e-12
    RA = beam_model.AiryResponse(powerbeam=True)
    b = RA(torch.ones(1, 1, 1, 1, 1) * 14.0, zen, az, freqs)
    assert np.abs(b.numpy() - g['airy_D14']).max() < 1e-13
    RG = beam_model.GaussResponse()
    assert np.abs(RG(torch.as_tensor(g['gauss_params']), zen, az, freqs).numpy() - g['gauss']).max() < 1e-14
    tg, pg = torch.as_tensor(g['pr_theta_grid']), torch.as_tensor(g['pr_phi_grid'])
    p = torch.as_tensor(g['pr_params'])
    for tag, kw in [('abs', {}), ('log', dict(log=True)), ('beam0', dict(beam0=torch.as_tensor(g['pr_beam0']))),
                    ('normpix', dict(norm_pix=3)), ('nonpower', dict(powerbeam=False))]:
        Rp = beam_model.PixelResponse(freqs, 'rect', interp_mode='linear', theta_grid=tg, phi_grid=pg, **kw)
        assert np.abs(Rp.forward(p.clone()).numpy() - g['pr_fwd_' + tag]).max() < 1e-14, tag


def test_apply_beam_layout_matches_reference():
    """apply_beam is pure torch: check all polarisation modes against the golden psky"""
    g = load_golden('apply_beam')
    bls = [tuple(b) for b in g['bls']]
    freqs = torch.linspace(120e6, 130e6, 4)
    for k in sorted({n.split('__')[0] for n in g if '__' in n}):
        beam = torch.as_tensor(g[k + '__beam'])
        a2b = g[k + '__ant2beam']
        pb = beam_model.PixelBeam(beam.clone(), freqs, parameter=False, powerbeam=bool(g[k + '__powerbeam']),
                                  ant2beam=None if a2b[0] < 0 else {i: int(a2b[i]) for i in range(3)}, pol='e')
        psky = pb.apply_beam(beam, bls, torch.as_tensor(g[k + '__sky']))
        assert psky.shape == g[k + '__psky'].shape, k
        assert np.abs(psky.numpy() - g[k + '__psky']).max() < 1e-12 * np.abs(g[k + '__psky']).max(), k


def test_eq2top_basic_geometry():
    """own LST rotation (parity unpinned vs astropy): a source at dec = lat transits at zenith"""
    tel = telescope_model.TelescopeModel((21.42827, -30.72148))
    jd = 2459861.3
    lst = telescope_model.JD2LST(jd, 21.42827)
    zen, az = telescope_model.eq2top(tel.location, jd, np.array([lst, lst, lst + 90]), np.array([-30.72148, 0.0, 0.0]))
    assert zen[0] < 1e-6 and abs(zen[1] - 30.72148) < 1e-6 and abs(az[1] - 0.0) < 1e-6
    assert abs(zen[2] - 90.0) < 1e-6 and abs(az[2] - 90.0) < 1e-6       # RA = LST + 6h: rising due east
    angs = tel.eq2top(jd, torch.tensor([0.0]), torch.tensor([0.0]), store=True)
    assert tel.hash(jd, torch.tensor([0.0])) in tel.conv_cache and angs.shape == (2, 1)   # tests/test_telescope.py:27-38


def _check_blocks(pairs, blocks, bl_mp=None, ant_model=None):
    """every baseline lands in exactly one slot of exactly one block, with the orientation rules
    include/rime_hip.h documents: V[row, col] = sum conj(E_row) E_col, direct = (row -> col)"""
    seen = np.zeros(len(pairs), dtype=int)
    for blk in blocks:
        ai = blk['ants_i']
        aj = blk['ants_j'] if blk['ants_j'] is not None else ai
        nd = nc = 0
        for tab, is_conj in ((blk['direct'], False), (blk['conj'], True)):
            ii, jj = np.nonzero(tab >= 0)
            for i, j in zip(ii, jj):
                b = tab[i, j]
                seen[b] += 1
                a1, a2 = pairs[b]
                row, col = ai[i], aj[j]
                assert (a1, a2) == ((col, row) if is_conj else (row, col))
                if bl_mp is not None:
                    assert bl_mp[b] == blk['mp']
                if blk['ants_j'] is None:
                    assert i // 32 <= j // 32                     # upper-triangular tiles only
                    if is_conj:
                        assert i // 32 < j // 32
                nd, nc = nd + (not is_conj), nc + is_conj
        assert blk['cpass'] == (1 if nc == 0 else (-1 if nd == 0 else 0))
        if blk['ants_j'] is not None:
            assert (blk['rows_i'], blk['rows_j']) in ((32, 32), (32, 64), (64, 64), (128, 128))
            assert len(ai) <= blk['rows_i'] and len(aj) <= blk['rows_j']
        if ant_model is not None:
            assert len({ant_model[a] for a in ai}) == 1 and len({ant_model[a] for a in aj}) == 1
    assert (seen == 1).all()


def test_astrometry_against_sofa_known_answers():
    """the eq2top chain (bayeslim_amd/astrometry.py, replacing astropy's ICRS->AltAz of
    telescope_model.py:469-502) against SOFA's published known answers, component by component"""
    import json
    import os
    from bayeslim_amd import astrometry as A
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'sofa_vectors.json')))
    jd = lambda mjd: 2400000.5 + mjd
    cen = lambda mjd: (jd(mjd) - 2451545.0) / 36525.0        # SOFA's test calls pass the same date as UT1 and TT
    assert abs(A.era(jd(g['era00']['mjd'])) - g['era00']['value']) < g['era00']['tol']
    assert abs(A.gmst(jd(g['gmst06']['mjd']), cen(g['gmst06']['mjd'])) - g['gmst06']['value']) < g['gmst06']['tol']
    assert abs(A.mean_obliquity(cen(g['obl06']['mjd'])) - g['obl06']['value']) < g['obl06']['tol']
    for k in ('nut80', 'nut00b'):                              # 31-term truncation: a few mas from either full series
        dpsi, deps = A.nutation(cen(g[k]['mjd']))
        assert abs(dpsi - g[k]['dpsi']) < g[k]['tol'] and abs(deps - g[k]['deps']) < g[k]['tol'], k
    T = cen(g['gst06a']['mjd'])
    dpsi, deps = A.nutation(T)
    assert abs(A.gast(jd(g['gst06a']['mjd']), T, dpsi, A.mean_obliquity(T)) - g['gst06a']['value']) < g['gst06a']['tol']
    assert np.abs(A.frame_bias() - np.array(g['bp00_rb']['value'])).max() < g['bp00_rb']['tol']
    rbp = A.precession_matrix(cen(g['pmat06']['mjd'])) @ A.frame_bias()
    assert np.abs(rbp - np.array(g['pmat06']['value'])).max() < g['pmat06']['tol']
    T = cen(g['epv00']['mjd'])
    v = (A.precession_matrix(T) @ A.frame_bias()).T @ A.earth_velocity(T) * A.C_AUDAY
    vb = np.array(g['epv00']['vel_bary_au_per_day'])
    assert np.linalg.norm(v - vb) / np.linalg.norm(vb) < g['epv00']['rel_tol']
    # leap seconds / TT
    assert A.tai_minus_utc(2459861.0) == 37.0 and A.tai_minus_utc(2451545.0) == 32.0 and A.tai_minus_utc(2441317.5) == 10.0
    assert abs(A.tt_centuries(2451545.0 - (32.0 + 32.184) / 86400.0)) < 1e-15


def test_eq2top_cache_miss_runs_the_astrometry_chain_and_says_so():
    """a conv_cache miss is no longer a bare LST rotation: precession since J2000 (~0.3 deg in 2022), nutation and
    aberration are applied, the first miss warns, and the chain is self-consistent (a source at the apparent
    zenith place comes out at the zenith; rigid rotation + sub-arcminute aberration of separations)"""
    import warnings
    from bayeslim_amd import telescope_model, astrometry as A
    telescope_model._WARNED = False
    loc = (21.42827, -30.72148)
    tel = telescope_model.TelescopeModel(loc)
    jdv = 2459861.3
    rng = np.random.default_rng(0)
    ra, dec = rng.uniform(0, 360, 400), np.rad2deg(np.arcsin(rng.uniform(-1, 1, 400)))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        za = tel.eq2top(jdv, torch.as_tensor(ra), torch.as_tensor(dec), store=True)
        tel.eq2top(jdv + 0.01, torch.as_tensor(ra), torch.as_tensor(dec))
    assert len([x for x in w if 'astrometry' in str(x.message)]) == 1
    assert tel.hash(jdv, torch.as_tensor(ra)) in tel.conv_cache and za.shape == (2, 400)
    zen, az = za[0].numpy(), za[1].numpy()
    z0, a0 = telescope_model.eq2top(loc, jdv, ra, dec)                   # plain rotation
    s = lambda z, a: np.stack([np.sin(np.deg2rad(z)) * np.sin(np.deg2rad(a)), np.sin(np.deg2rad(z)) * np.cos(np.deg2rad(a)),
                               np.cos(np.deg2rad(z))])
    sep = np.rad2deg(np.arccos(np.clip((s(zen, az) * s(z0, a0)).sum(0), -1, 1)))
    assert 0.15 < np.median(sep) < 0.45 and sep.max() < 0.5              # precession over 22.8 years + nutation + aberration
    # pairwise separations are preserved up to differential aberration (< 41 arcsec)
    p = np.stack([np.cos(np.deg2rad(dec)) * np.cos(np.deg2rad(ra)), np.cos(np.deg2rad(dec)) * np.sin(np.deg2rad(ra)),
                  np.sin(np.deg2rad(dec))])
    d_in = np.arccos(np.clip(p[:, :200].T @ p[:, 200:], -1, 1))
    d_out = np.arccos(np.clip(s(zen, az)[:, :200].T @ s(zen, az)[:, 200:], -1, 1))
    assert np.abs(d_in - d_out).max() < 2.1e-4
    # the direction whose TRUE place is (GAST + lon, lat) is the zenith: build its ICRS place by inverting N P B
    M, vb, vd = A.observation_frame(loc, jdv)
    T = A.tt_centuries(jdv)
    dpsi, deps = A.nutation(T)
    eps0 = A.mean_obliquity(T)
    NPB = A.nutation_matrix(eps0, dpsi, deps) @ A.precession_matrix(T) @ A.frame_bias()
    lst = A.gast(jdv, T, dpsi, eps0) + np.deg2rad(loc[0])
    ptrue = np.array([np.cos(np.deg2rad(loc[1])) * np.cos(lst), np.cos(np.deg2rad(loc[1])) * np.sin(lst), np.sin(np.deg2rad(loc[1]))])
    picrs = NPB.T @ ptrue
    assert np.allclose(M @ picrs, [0, 0, 1], atol=1e-14)
    zz, _ = A.icrs_to_topo(loc, jdv, np.rad2deg(np.arctan2(picrs[1], picrs[0])), np.rad2deg(np.arcsin(picrs[2])))
    assert float(np.max(zz)) < 21.0 / 3600.0                                             # only aberration moves it: <= 20.5 + 0.3 arcsec


def _healpix_vectors():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'healpix_vectors.json')))


def test_healpix_pinned_to_docstring_and_hand_derived_vectors():
    """RING pix2ang and get_interp_weights (utils.py:765-769 calls healpy's) against healpy's documented
    examples and hand-derived values: pixel centres, phi wrap, ring boundaries, shifted rings, polar caps"""
    from bayeslim_amd import healpix
    g = _healpix_vectors()
    for c in g['pix2ang']:
        ip = c.get('pix', list(range(12 * c['nside'] ** 2)))
        th, ph = healpix.pix2ang(c['nside'], ip)
        assert np.abs(np.cos(th) - np.array(c['z'])).max() < 1e-15 and np.abs(ph - np.array(c['phi'])).max() < 1e-15
    for c in g['interp']:
        pix, w = healpix.get_interp_weights(c['nside'], c['theta'], c['phi'])
        assert pix.shape == (4, 1) and abs(w.sum() - 1) < 1e-15
        dense = np.zeros(12 * c['nside'] ** 2)
        np.add.at(dense, pix[:, 0], w[:, 0])
        want = np.zeros_like(dense)
        for k, v in c['weights'].items():
            want[int(k)] = v
        assert np.abs(dense - want).max() < 1e-14, c['tag']


def test_mirror_pairs_and_row_order():
    """round 5, host side of the conjugate-phasor pairing (ops._mirror_pairs / _mirror_order): hexagons (with an outrigger),
    an explicitly mirrored random set with a 3-D centre, sets without symmetry, a set whose symmetry holds in x, y but not in
    z; the row order puts the two antennas of a pair 8 rows apart inside mirror groups and keeps the block's row capacity"""
    from bayeslim_amd import ops
    rng = np.random.default_rng(1)
    hex7 = utils._make_hex(7, D=14.6)[1]
    c, pairs, singles = ops._mirror_pairs(hex7)
    assert len(pairs) == 63 and len(singles) == 1 and np.abs(c).max() < 1e-9 and np.abs(hex7[singles[0]]).max() < 1e-9
    ant = np.vstack([hex7, [[250.0, 0.0, 0.0]]]) + [5.0, -3.0, 1.0]
    c, pairs, singles = ops._mirror_pairs(ant)
    assert len(pairs) == 63 and len(singles) == 2 and np.abs(c - [5.0, -3.0, 1.0]).max() < 1e-9
    for a, b in pairs:
        assert np.abs(ant[a] + ant[b] - 2 * c).max() < 1e-9
    rows, mask, c2 = ops._mirror_order(ant)
    assert len(rows) == 128 and mask == 0x7f and sorted(rows) == list(range(128))      # 7 mirror groups + one plain group of 16
    for g in range(7):
        for i in range(8):
            assert np.abs(ant[rows[16 * g + i]] + ant[rows[16 * g + 8 + i]] - 2 * c2).max() < 1e-9
    # HERA-19 / HERA-37 (the packed shape: mirror groups in the first row tile only, <= 48 rows)
    rows, mask, _ = ops._mirror_order(utils._make_hex(3, D=14.6)[1])
    assert mask == 3 and len(rows) <= 32 and sorted(r for r in rows if r >= 0) == list(range(19))
    rows, mask, _ = ops._mirror_order(utils._make_hex(4, D=14.6)[1])
    assert mask == 3 and len(rows) == 37 and sorted(rows) == list(range(37))
    # mirrored random set, tilted, centre off the origin, shuffled; singles fill the last slots / a plain group
    h = rng.normal(0, 60.0, (20, 3))
    ant = np.vstack([h, -h, rng.normal(0, 60.0, (5, 3))])[rng.permutation(45)] + [100.0, 20.0, -7.0]
    c, pairs, singles = ops._mirror_pairs(ant)
    assert len(pairs) == 20 and len(singles) == 5 and np.abs(c - [100.0, 20.0, -7.0]).max() < 1e-9
    rows, mask, _ = ops._mirror_order(ant)
    assert mask == 3 and len(rows) <= 48
    # no symmetry: nothing; symmetry in the plane only: nothing
    assert ops._mirror_pairs(rng.normal(0, 60.0, (40, 3))) is None
    flat = np.vstack([h, -h])
    flat[:, 2] = rng.normal(0, 1.0, 40)
    assert ops._mirror_pairs(flat) is None
    # a mismatch above the tolerance breaks a pair, one below it does not
    ant = np.vstack([h, -h])
    ant[3] += [2e-9, 0, 0]
    assert len(ops._mirror_pairs(ant)[1]) == 19
    ant = np.vstack([h, -h])
    ant[3] += [2e-10, 0, 0]
    assert len(ops._mirror_pairs(ant)[1]) == 20


def _pair_form_emulated(P, seed):
    """numpy emulation of the conjugate-pair kernels' table logic (csrc/fringe_mfma.hip, pair_fwd_body's epilogue and
    fringe_pair_bwd_kernel's staging) on the block ops._pair_block builds for positions P, all pairs with random orientation:
    returns (rows, hub, forward error, backward error) against the baseline formulation in float64"""
    from bayeslim_amd import ops
    n = len(P)
    rng = np.random.default_rng(seed)
    bls = [(i, j) if rng.random() < 0.5 else (j, i) for i in range(n) for j in range(i + 1, n)]
    raw = ops._antenna_blocks(bls, n)
    assert len(raw) == 1
    blk = dict(nrows=n, cross=0, mp=0, direct=torch.as_tensor(raw[0]['direct'].reshape(-1)), conj=torch.as_tensor(raw[0]['conj'].reshape(-1)))
    Pb = P[np.asarray(raw[0]['ants_i'])]
    q = ops._pair_block(blk, Pb, 'cpu')
    if q is None:
        return None
    F, pos, npx = q['nrows'], q['pos'].numpy(), 24
    s, w = rng.normal(size=(npx, 3)), rng.normal(size=npx)
    E = np.zeros((64, npx), complex)
    E[:F] = np.exp(2j * np.pi * (pos @ s.T))
    A, B = np.conj(E) @ np.diag(w) @ E.T, E @ np.diag(w) @ E.T
    direct, conj = q['direct'].numpy().reshape(128, 128), q['conj'].numpy().reshape(128, 128)
    cen = None if q['centre'] is None else q['centre'].numpy().reshape(2, 128)
    vis, cnt = np.zeros(len(bls), complex), np.zeros(len(bls), int)

    def put(r, c, v):
        for tab, val in ((direct[r, c], v), (conj[r, c], np.conj(v))):
            if tab >= 0:
                vis[tab] = val
                cnt[tab] += 1
    tiles = [(0, 1), (0, 0), (1, 1)]
    for ti, tj in tiles:
        for i in range(32 * ti, 32 * ti + 32):
            for j in range(32 * tj, 32 * tj + 32):
                put(i, j, A[i, j]); put(64 + i, 64 + j, np.conj(A[i, j])); put(i, 64 + j, np.conj(B[i, j]))     # V[i, j'] = conj B
                if ti != tj:
                    put(j, 64 + i, np.conj(B[i, j]))                                                       # V[j, i'] (B = B^T)
    if cen is not None:
        col = E @ w
        for r in range(64):
            for tab, val in ((cen[0][r], col[r]), (cen[1][r], np.conj(col[r])), (cen[0][64 + r], np.conj(col[r])), (cen[1][64 + r], col[r])):
                if tab >= 0:
                    vis[tab] = val
                    cnt[tab] += 1
    assert (cnt == 1).all()                                     # every baseline written exactly once
    Fb = np.array([np.exp(2j * np.pi * ((Pb[b] - Pb[a]) @ s.T)) for a, b in bls])
    ref = Fb @ w
    g = rng.normal(size=len(bls)) + 1j * rng.normal(size=len(bls))
    gref = np.real(np.conj(g) @ Fb)

    def grad_of(r, c):
        return (g[direct[r, c]] if direct[r, c] >= 0 else 0) + (np.conj(g[conj[r, c]]) if conj[r, c] >= 0 else 0)

    def entry(i, j, off):
        return np.conj(grad_of(i, j)) + grad_of(64 + i, 64 + j), grad_of(j, 64 + i) + (grad_of(i, 64 + j) if off else 0)
    N1, N2, N3, N4 = (np.zeros((64, 64)) for _ in range(4))
    for ti, tj in tiles:
        for i in range(32 * ti, 32 * ti + 32):
            for j in range(32 * tj, 32 * tj + 32):
                a, b = entry(i, j, ti != tj)
                if ti != tj:
                    N1[i, j], N2[i, j], N3[i, j], N4[i, j] = a.real + b.real, a.real - b.real, -(a.imag + b.imag), a.imag - b.imag
                else:
                    a2, b2 = entry(j, i, False)
                    N1[i, j] = 0.5 * ((a.real + b.real) + (a2.real + b2.real))
                    N2[i, j] = 0.5 * ((a.real - b.real) + (a2.real - b2.real))
                    N3[i, j] = (a2.imag - b2.imag) - (a.imag + b.imag)
    accR, accI = N1 @ E.real + N3 @ E.imag, N2 @ E.imag + N4 @ E.real
    if cen is not None:
        for r in range(64):
            for tab, sv in ((cen[0][r], 1), (cen[1][r], -1), (cen[0][64 + r], -1), (cen[1][64 + r], 1)):
                if tab >= 0:
                    accR[r] += g[tab].real
                    accI[r] += sv * g[tab].imag
    gp = (E.real * accR + E.imag * accI).sum(0)
    return F, q['hub'], np.abs(vis - ref).max() / np.abs(ref).max(), np.abs(gp - gref).max() / np.abs(gref).max()


def test_conjugate_pair_form_tables_and_algebra():
    """round 5, host side of the conjugate-pair form (ops._pair_layout / _pair_block) and the algebra its kernels implement,
    emulated in numpy: the headline array (63 pairs + outrigger in 64 rows, the hub outside them), a bare 127-antenna hexagon
    (the hub takes the 64th row), a 91-antenna hexagon; every baseline is written exactly once and both directions agree with
    the baseline formulation to float64 rounding.  37 antennas take one row tile (19 rows); blocks of up to 32 antennas, sets
    without symmetry and sets whose firsts and singles exceed the rows (32 / 64 + hub) keep their kernels"""
    from bayeslim_amd import ops
    hex7 = np.asarray(utils._make_hex(7, D=14.6)[1])
    hera128 = np.vstack([hex7, [[250.0, 0.0, 0.0]]]) + [3.0, -2.0, 0.5]
    F, hub, ef, eb = _pair_form_emulated(hera128, 0)
    assert F == 64 and hub is not None and np.abs(hera128[hub] - [3.0, -2.0, 0.5]).max() < 1e-9 and ef < 1e-11 and eb < 1e-11
    F, hub, ef, eb = _pair_form_emulated(hex7, 1)
    assert F == 64 and hub is None and ef < 1e-11 and eb < 1e-11
    F, hub, ef, eb = _pair_form_emulated(np.asarray(utils._make_hex(6, D=14.6)[1]), 2)
    assert F == 46 and hub is None and ef < 1e-11 and eb < 1e-11
    rng = np.random.default_rng(5)
    F, hub, ef, eb = _pair_form_emulated(np.asarray(utils._make_hex(4, D=14.6)[1]), 3)            # 37 antennas: one row tile
    assert F == 19 and hub is None and ef < 1e-11 and eb < 1e-11
    assert _pair_form_emulated(np.asarray(utils._make_hex(3, D=14.6)[1]), 6) is None             # 19 antennas: one-tile kernels
    h2 = rng.normal(0, 60.0, (20, 3))
    assert _pair_form_emulated(np.vstack([h2, -h2, rng.normal(0, 60.0, (13, 3))]), 7) is None    # 53 antennas, 33 rows
    assert _pair_form_emulated(rng.normal(0, 60.0, (100, 3)), 4) is None                         # no symmetry
    h = rng.normal(0, 60.0, (60, 3))
    assert _pair_form_emulated(np.vstack([h, -h, rng.normal(0, 60.0, (8, 3))]), 5) is None       # 68 rows, no hub
    lay = ops._pair_layout(np.vstack([h, -h, np.zeros((1, 3)), rng.normal(0, 60.0, (4, 3))]))    # 60 pairs + 4 singles + hub
    assert lay is not None and len(lay[0]) == 64 and lay[2] == 120
    assert ops._pair_layout(np.vstack([h, -h, np.zeros((1, 3)), rng.normal(0, 60.0, (5, 3))])) is None   # 65 rows + hub


def test_antenna_block_tables():
    """pair tables of the matrix-core path: groups of <= 128 antennas, diagonal + cross blocks"""
    from bayeslim_amd import ops
    rng = np.random.default_rng(0)
    Nant = 300                                   # groups of 128, 128, 44
    pairs = [(i, j) for i in range(Nant) for j in range(i, Nant) if rng.random() < 0.05]
    pairs = [p if rng.random() < 0.5 else p[::-1] for p in pairs]
    blocks = ops._antenna_blocks(pairs, Nant)
    assert len(blocks) == 6
    _check_blocks(pairs, blocks)
    # a pair and its reverse are distinct slots; the same pair twice cannot be represented
    assert ops._antenna_blocks([(1, 2), (2, 1)], 4) is not None
    assert ops._antenna_blocks([(130, 2), (2, 130)], 140) is not None
    assert ops._antenna_blocks([(1, 2), (1, 2)], 4) is None
    assert ops._antenna_blocks([(2, 130), (2, 130)], 140) is None


def test_antenna_blocks_per_beam_model_and_small_groups():
    """several beam models: groups never mix models and every (group pair, model pair) is its own block
    with its psky plane; groups of 32 (rank-local tile shards) give one-tile blocks"""
    from bayeslim_amd import ops
    rng = np.random.default_rng(1)
    Nant = 90
    ant_model = [int(a % 3 == 0) + 2 * int(a % 7 == 0 and a % 3 != 0) for a in range(Nant)]       # three models, unequal sizes
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    pairs = [p if rng.random() < 0.8 else p[::-1] for p in pairs]
    uniq = sorted({(ant_model[a], ant_model[b]) for a, b in pairs})
    bl_mp = [uniq.index((ant_model[a], ant_model[b])) for a, b in pairs]
    blocks = ops._antenna_blocks(pairs, Nant, bl_mp, ant_model)
    _check_blocks(pairs, blocks, bl_mp, ant_model)
    assert {blk['mp'] for blk in blocks} == set(range(len(uniq)))
    # one model, groups of 32: 128 antennas -> 4 diagonal + 6 cross one-tile blocks
    pairs = [(i, j) for i in range(128) for j in range(i + 1, 128)]
    blocks = ops._antenna_blocks(pairs, 128, group=32)
    _check_blocks(pairs, blocks)
    assert sum(b['ants_j'] is None for b in blocks) == 4 and sum(b['ants_j'] is not None for b in blocks) == 6
    assert all((b['rows_i'], b['rows_j']) == (32, 32) for b in blocks if b['ants_j'] is not None)


def test_eq2top_device_matches_host():
    """the torch restatement of the LST rotation (used when the sky angles live on the GPU) agrees with
    the numpy one to rounding"""
    from bayeslim_amd import telescope_model
    rng = np.random.default_rng(0)
    ra, dec = rng.uniform(0, 360, 5000), np.rad2deg(np.arcsin(rng.uniform(-1, 1, 5000)))
    for jd in (2459861.0, 2459861.37, 2460000.123456):
        zen, az = telescope_model.eq2top((21.42827, -30.72148), jd, ra, dec)
        za = telescope_model.eq2top_device((21.42827, -30.72148), jd, torch.as_tensor(ra), torch.as_tensor(dec))
        assert za.dtype == torch.float64 and za.shape == (2, 5000)
        assert np.abs(za[0].numpy() - zen).max() < 1e-9
        daz = np.abs(za[1].numpy() - az)
        assert np.minimum(daz, 360 - daz).max() < 1e-8          # (wraps at 0 / 360)


def test_visdata_selection_and_copy():
    """VisData.get_inds / get_data / get_icov / copy (dataset.py:556-1042): what imaging.VisMapper reads through"""
    from bayeslim_amd import dataset, utils
    bls = [(0, 1), (0, 2), (1, 2), (1, 3)]
    times, freqs = [2459861.0, 2459861.1, 2459861.2], [1.0e8, 1.1e8, 1.2e8, 1.3e8, 1.4e8]
    data = torch.arange(60, dtype=torch.float64).reshape(1, 1, 4, 3, 5) * (1 + 1j)
    vd = dataset.VisData()
    vd.setup_data(bls, times, freqs, pol='ee', data=data, icov=data.real + 1)
    assert utils._list2slice([1, 3, 5]) == slice(1, 7, 2) and utils._list2slice([3, 1]) == [3, 1]
    assert utils._list2slice(2) == slice(2, 3) and utils._slice2tensor(slice(1, 4)).tolist() == [1, 2, 3]
    assert vd.get_inds(bl=[(0, 2), (1, 3)])[2] == slice(1, 5, 2)
    assert vd.get_inds(times=2459861.1)[3] == slice(1, 2, 1) and vd.get_inds(freqs=[1.1e8, 1.4e8])[4] == slice(1, 7, 3)
    assert vd.get_inds(pol='ee')[:2] == (slice(0, 1), slice(0, 1))
    with pytest.raises(ValueError):
        vd.get_inds(bl=(2, 3))
    with pytest.raises(AssertionError):
        vd.get_inds(bl_inds=[0, 2, 3], time_inds=[0, 2, 1])                        # two fancy-indexed axes
    assert torch.equal(vd.get_data(bl=(1, 2)), data[0, 0, 2])
    assert torch.equal(vd.get_data(bl_inds=[3, 0], squeeze=False), data[:, :, [3, 0]])
    assert torch.equal(vd.get_icov(time_inds=2, freq_inds=slice(1, 3), squeeze=False), (data.real + 1)[:, :, :, 2:3, 1:3])
    view = vd.get_data(time_inds=slice(0, 1), squeeze=False, try_view=True)
    assert view.data_ptr() == data.data_ptr() and vd.get_data(time_inds=slice(0, 1), squeeze=False).data_ptr() != data.data_ptr()
    c = vd.copy(copydata=True, copymeta=True)
    assert c.data.data_ptr() != data.data_ptr() and torch.equal(c.data, data) and c.bls == vd.bls and c.pol == 'ee'
    assert vd.copy().data.data_ptr() == data.data_ptr()


def test_logprob_container_on_a_toy_model():
    """optim.LogProb on CPU with a two-minibatch toy model: closure() accumulation and averaging, priors counted once,
    main-parameter tensor (non-leaf tensors on the module, values follow main_params, Parameters restored), stochastic
    mode, gradient modifiers, negate / compute switches (optim.py:385-1389)"""
    from bayeslim_amd import optim, dataset, utils

    class Toy(utils.Module):
        def __init__(self):
            super().__init__(name='toy')
            self.params = torch.nn.Parameter(torch.tensor([[1.0, 2.0, 3.0], [0.5, -1.0, 2.0]], dtype=torch.float64))
            self.Nbatch, self.batch_idx = 2, 0

        def forward(self, inp=None, prior_cache=None, **kw):
            self.eval_prior(prior_cache)
            td = dataset.TensorData()
            td.data = self.params[self.batch_idx] * torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
            return td

    toy = Toy()
    toy.set_priors(priors_inp_params=[optim.LogGaussPrior(torch.zeros(2, 3, dtype=torch.float64), torch.full((2, 3), 4.0, dtype=torch.float64),
                                                          density=False)])
    tds = []
    for i in range(2):
        td = dataset.TensorData()
        td.data = torch.tensor([0.5, 3.0, 10.0], dtype=torch.float64) * (i + 1)
        td.set_cov(None, None, icov=torch.tensor([1.0, 0.5, 2.0], dtype=torch.float64))
        tds.append(td)
    prob = optim.LogProb(toy, dataset.Dataset(tds), complex_circular=False)
    w, ic = np.array([1.0, 2.0, 3.0]), np.array([1.0, 0.5, 2.0])
    p = toy.params.detach().numpy().copy()

    def expect(p):
        like = []
        for i in range(2):
            res = p[i] * w - np.array([0.5, 3.0, 10.0]) * (i + 1)
            like.append(0.5 * (res ** 2 * ic).sum() + 0.5 * (3 * np.log(2 * np.pi) + (-np.log(ic)).sum()))
        prior = 0.5 * (p ** 2 / 4.0).sum()
        grad = np.stack([(p[i] * w - np.array([0.5, 3.0, 10.0]) * (i + 1)) * ic * w for i in range(2)]) + p / 4.0
        return (like[0] + like[1] + prior) / 2, grad, like, prior

    loss, grad, like, prior = expect(p)
    assert abs(float(prob.closure()) - loss) < 1e-12 and np.allclose(toy.params.grad.numpy(), grad, atol=1e-12)
    assert prob.batch_idx == 0 and prob.closure_eval == 1 and prob.prior_cache == {}
    assert abs(float(prob.forward_like(1).detach()) - like[1]) < 1e-12
    prob.grad_type = 'stochastic'
    prob.batch_idx = 1
    assert abs(float(prob.closure()) - like[1]) < 1e-12                  # the prior belongs to batch 0 only
    assert np.allclose(toy.params.grad.numpy()[0], 0) and np.allclose(toy.params.grad.numpy()[1], grad[1] - p[1] / 4.0)
    prob.grad_type = 'accumulate'
    # main parameters: row 1 only
    prob.set_main_params([('params', (1, [0, 2]), 'row1')])
    assert prob.main_params.tolist() == [0.5, 2.0] and not toy.params.is_leaf
    assert abs(float(prob.closure()) - loss) < 1e-12 and np.allclose(prob.main_params.grad.numpy(), grad[1][[0, 2]], atol=1e-12)
    with torch.no_grad():
        prob.main_params += torch.tensor([1.0, -1.0], dtype=torch.float64)
    p2 = p.copy()
    p2[1, [0, 2]] += [1.0, -1.0]
    assert abs(float(prob.closure()) - expect(p2)[0]) < 1e-12 and np.allclose(toy.params.detach().numpy(), p2)
    prob.set_main_params(None)
    assert isinstance(toy.params, torch.nn.Parameter) and np.allclose(toy.params.detach().numpy(), p2)
    # gradient modifiers, sign and distribution switches
    prob.set_grad_mod([('model.params', {'mod_type': 'mult', 'value': 2.0, 'index': (0,)}),
                       ('model.params', {'mod_type': 'replace', 'value': 7.0, 'index': (1, 1)})], alpha=0.5)
    prob.closure()
    g2 = expect(p2)[1]
    assert np.allclose(toy.params.grad.numpy()[0], g2[0]) and toy.params.grad[1, 1] == 3.5        # 2.0 * alpha = 1, 7 * alpha
    prob.set_grad_mod()
    prob.negate = False
    assert abs(float(prob.closure()) + expect(p2)[0]) < 1e-12
    prob.negate, prob.compute = True, 'prior'
    assert abs(float(prob.closure()) - expect(p2)[3] / 2) < 1e-12
    u = optim.LogUniformPrior(torch.tensor(0.0), torch.tensor(2.0))
    # the normalisation is summed over the BOUNDS' shape (scalar bounds: once), as in the reference
    assert float(u(torch.tensor([0.5, 1.5]))) == float(np.log(0.5)) and float(u(torch.tensor([0.5, 2.5]))) == -np.inf


def test_trainer_loop_and_chain():
    """optim.Trainer (optim.py:1631-1833): epochs of optimiser steps on LogProb.closure, loss / time records, parameter
    chain and revert"""
    from bayeslim_amd import optim, dataset, utils

    class Line(utils.Module):
        def __init__(self):
            super().__init__(name='line')
            self.params = torch.nn.Parameter(torch.tensor([0.0, 0.0], dtype=torch.float64))

        def forward(self, inp=None, prior_cache=None, **kw):
            td = dataset.TensorData()
            td.data = self.params[0] + self.params[1] * torch.arange(5, dtype=torch.float64)
            return td

    td = dataset.TensorData()
    td.data = 1.0 + 2.0 * torch.arange(5, dtype=torch.float64)
    prob = optim.LogProb(Line(), dataset.Dataset([td]), complex_circular=False)
    tr = optim.Trainer(prob, torch.optim.SGD, track=True)
    assert list(tr.chain) == ['model.params'] and tr.Nbatch == 1
    tr.set_opt(torch.optim.SGD, lr=0.02)
    info = tr.train(Nepochs=200)
    assert info['duration'] > 0 and len(tr.loss) == 200 and tr.times[-1] >= tr.times[0]
    assert float(tr.loss[-1]) < 1e-6 * float(tr.loss[0]) + 1e-9
    assert torch.allclose(prob.model.params.detach(), torch.tensor([1.0, 2.0], dtype=torch.float64), atol=1e-3)
    chain = tr.get_chain('model.params')
    assert chain.shape == (200, 2) and torch.equal(chain[0], torch.zeros(2, dtype=torch.float64))
    tr.revert_chain(150)
    assert len(tr.loss) == 50 and torch.equal(prob.model.params.detach(), chain[50])


def test_astrometry_host_chain_against_the_independent_oracle():
    """the product's numpy float64 chain (astrometry.icrs_to_topo: what host tensors take, and what the device kernel
    is compared with to 1e-10 deg) against oracle/eq2top_oracle.py -- independent factorisation, pinned end to end
    to SOFA's atci13 / atio13 / atco13 known answers (tests/test_oracle_golden.py).  20 mas on the sky."""
    import numpy as np
    from bayeslim_amd import astrometry as A
    from oracle import eq2top_oracle as E
    rng = np.random.default_rng(0)
    ra = rng.uniform(0, 360, 5000)
    dec = np.rad2deg(np.arcsin(rng.uniform(-1, 1, 5000)))
    for loc in [(21.42827, -30.72148, 1050.0), (-107.6, 34.08, 2124.0)]:
        for jd, dut1 in [(2459861.37, -0.02), (2451545.0, 0.3), (2462000.25, 0.0)]:
            zen, az = A.icrs_to_topo(loc, jd, ra, dec, dut1=dut1)
            z2, a2 = E.eq2top(loc, jd, ra, dec, dut1)
            da = np.abs(az - a2)
            da = np.minimum(da, 360 - da) * np.sin(np.deg2rad(zen))
            assert np.abs(zen - z2).max() * 3600 < 0.020 and da.max() * 3600 < 0.020
    # a bare LST rotation (round 1's stand-in) is 0.3 deg away from the same oracle: the tolerance has teeth
    from bayeslim_amd import telescope_model
    zen, az = telescope_model.eq2top((21.42827, -30.72148), 2459861.37, ra, dec)
    z2, _ = E.eq2top((21.42827, -30.72148, 0.0), 2459861.37, ra, dec)
    assert np.abs(np.asarray(zen) - z2).max() > 0.1


def test_astrometry_with_earth_orientation_parameters_reproduces_sofa_atco13(tmp_path):
    """UT1-UTC and polar motion as inputs.  (i) the host chain with the dut1, xp, yp of SOFA's atco13
    case reproduces its published observed direction to the oracle's level (15 mas; 0.21 arcsec without the polar motion);
    (ii) against the independent oracle with polar motion on random directions; (iii) the same values read from IERS
    tables in the three accepted formats through TelescopeModel(iers_file=...), interpolated, across a leap second."""
    import json, math, os
    import numpy as np
    from bayeslim_amd import astrometry as A, telescope_model
    from oracle import eq2top_oracle as E
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'sofa_vectors.json')))
    ap, c = g['apco13'], g['atco13']
    jd = c['utc1'] + c['utc2']
    p = E.sofa_case_star_direction(c['rc'], c['dc'], c['pr'], c['pd'], c['px'], c['rv'], jd + (E.dat(jd) + 32.184) / 86400.0)
    ra, dec = math.degrees(math.atan2(p[1, 0], p[0, 0])), math.degrees(math.asin(p[2, 0]))
    loc = (math.degrees(c['elong']), math.degrees(c['phi']), c['hm'])
    ztrue = math.degrees(E.sofa_case_remove_refraction(c['zob'], ap['refa'], ap['refb']))
    mas = 1.0 / 3.6e6

    def err(zen, az):
        return (abs(float(zen[0]) - ztrue) / mas, abs(float(az[0]) - math.degrees(c['aob'])) * math.sin(c['zob']) / mas)

    e_with = err(*A.icrs_to_topo(loc, jd, [ra], [dec], c['dut1'], c['xp'], c['yp']))
    e_without = err(*A.icrs_to_topo(loc, jd, [ra], [dec], c['dut1']))
    assert max(e_with) < 15.0, e_with
    assert max(e_without) > 150.0, e_without                 # the polar motion of this case: 0.21 arcsec in zenith angle
    # (ii) random directions, two sites, polar motion and UT1-UTC of realistic size
    rng = np.random.default_rng(3)
    rr, dd = rng.uniform(0, 360, 3000), np.rad2deg(np.arcsin(rng.uniform(-1, 1, 3000)))
    for site in [(21.42827, -30.72148, 1050.0), (-107.6, 34.08, 2124.0)]:
        for jdv, dut1, xp, yp in [(2459861.37, -0.0123, 0.21 * A.AS2R, 0.35 * A.AS2R), (2457755.1, 0.591, -0.1 * A.AS2R, 0.5 * A.AS2R)]:
            zen, az = A.icrs_to_topo(site, jdv, rr, dd, dut1, xp, yp)
            z2, a2 = E.eq2top(site, jdv, rr, dd, dut1, xp, yp)
            da = np.abs(az - a2)
            da = np.minimum(da, 360 - da) * np.sin(np.deg2rad(zen))
            assert np.abs(zen - z2).max() * 3600 < 0.020 and da.max() * 3600 < 0.020
    # (iii) IERS tables.  Rows around the date of the SOFA case (+ a leap second on 2017-01-01, MJD 57754 = UT1-UTC jumps by +1 s)
    mjd0 = math.floor(jd - 2400000.5)
    as_ = 1.0 / A.AS2R
    rows = [(mjd0 - 1, c['xp'] * as_ - 0.002, c['yp'] * as_ + 0.001, c['dut1'] + 0.0011),
            (mjd0, c['xp'] * as_ - 0.0005, c['yp'] * as_ + 0.0003, c['dut1'] + 0.0005),
            (mjd0 + 1, c['xp'] * as_ + 0.0010, c['yp'] * as_ - 0.0004, c['dut1'] - 0.0006),
            (57753, 0.1, 0.3, -0.5910), (57754, 0.1, 0.3, 0.4080), (57755, 0.1, 0.3, 0.4070)]
    plain = tmp_path / 'eop.txt'
    plain.write_text('# MJD xp yp UT1-UTC\n' + ''.join('%d %.6f %.6f %.7f\n' % r for r in rows))
    c04 = tmp_path / 'eopc04.txt'
    c04.write_text('  # header line\n' + ''.join('2013   4   %d  %d  %.6f  %.6f  %.7f  0.001 0.0 0.0 0.0 0.0\n' % ((r[0] % 30,) + r) for r in rows))
    finals = tmp_path / 'finals2000A.all'
    finals.write_text(''.join('13 4 2 %8.2f I %9.6f 0.000040 %9.6f 0.000030  I%10.7f 0.0000050' % r + ' ' * 100 + '\n' for r in rows)
                      + ' ' * 7 + '%8.2f' % 99999.0 + ' ' * 170 + '\n'               # a trailing row without values
                      # the date-only rows a real finals2000A.all ends with (no flags, no values; single-digit month and day
                      # give FOUR tokens, which the plain-text branch used to read as a row at MJD 17: ADVICE r04)
                      + '17 5 1 57874.00' + ' ' * 170 + '\n' + '17 5 2 57875.00\n' + '171231 58118.00' + ' ' * 60 + '\n')
    frac = jd - 2400000.5 - mjd0
    want_dut1 = rows[1][3] + frac * (rows[2][3] - rows[1][3])
    want_xp = (rows[1][1] + frac * (rows[2][1] - rows[1][1])) * A.AS2R
    for path in (plain, c04, finals):
        eop = A.EarthOrientation.from_file(str(path))
        assert len(eop.mjd) == 6
        d, x, y = eop.at(jd)
        assert abs(d - want_dut1) < 1e-9 and abs(x - want_xp) < 1e-15, path
        # half a day before the leap second: interpolated on UT1 - TAI, not across the 1-s jump
        d2, _, _ = eop.at(2400000.5 + 57753.5)
        assert abs(d2 - (-0.5910 - 0.0005)) < 1e-9, (path, d2)
        d3, _, _ = eop.at(2400000.5 + 57754.5)
        assert abs(d3 - 0.4075) < 1e-9
        assert not eop.extrapolated
        eop.at(2400000.5 + 60000.0)
        assert eop.extrapolated
    telescope_model._WARNED = True
    tel = telescope_model.TelescopeModel(loc, iers_file=str(finals))
    tel.use_astropy = False
    za = tel.eq2top(jd, np.array([ra]), np.array([dec])).numpy()
    assert max(err(za[0], za[1])) < 15.0
    tel0 = telescope_model.TelescopeModel(loc)
    tel0.use_astropy = False
    tel0.dut1 = c['dut1']
    assert max(err(*tel0.eq2top(jd, np.array([ra]), np.array([dec])).numpy())) > 150.0
    assert tel0.earth_orientation(jd) == (c['dut1'], 0.0, 0.0)


def test_rime_zenaz_cache_follows_the_conv_cache_entry_not_its_address():
    """RIME's per-key (zen, az) cache was validated by id() of the telescope's conv_cache entry; after
    clear_cache() + repopulate CPython can hand the same address to the new entry and the stale angles were served.
    The cache now holds the entry itself and compares with `is`: 200 clear / repopulate rounds never return a stale
    value (the old check failed this within a few rounds)."""
    import types
    import numpy as np
    import torch
    from bayeslim_amd import rime_model, telescope_model
    tel = telescope_model.TelescopeModel((21.42827, -30.72148))
    fake = types.SimpleNamespace(telescope=tel, _zenaz_cache={}, cache_eq2top=True)
    key = ('sky', 5, 2459861.0)
    ra = torch.linspace(0, 300, 5, dtype=torch.float64)
    dec = torch.linspace(-60, 20, 5, dtype=torch.float64)
    dev = torch.device('cpu')
    for k in range(200):
        tel.clear_cache()
        entry = torch.stack([torch.full((5,), float(k)), torch.full((5,), 2.0 * k)]).double()
        tel.conv_cache[key] = entry
        del entry
        zen, az = rime_model.RIME._zenaz(fake, key, 2459861.0, ra, dec, dev)
        assert float(zen[0]) == float(k) and float(az[0]) == 2.0 * k, k
        zen2, _ = rime_model.RIME._zenaz(fake, key, 2459861.0, ra, dec, dev)      # second call: the cached pair
        assert zen2 is zen
