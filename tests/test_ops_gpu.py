"""
GPU parity tests: the HIP kernels, called through the C ABI (bayeslim_amd.ops -> ctypes ->
librime_hip.so), against the float64 CPU oracle and the committed golden vectors.

Tolerances (BASELINE.json north_star): complex visibilities within rtol 1e-5 in fp32,
gradients within 1e-4, both measured against the fp64 oracle and scaled by the tensor's
max magnitude (SURVEY.md section 7 "hard parts": a per-element rtol is not meaningful for a sum
of ~1e3-turn phasors).  The float64 kernels must agree to roundoff (1e-11).
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import rime_oracle as orc

pytestmark = pytest.mark.gpu

T64 = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from bayeslim_amd import ops as _ops
    return _ops


def relmax(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def oracle_fringe_sum(psky, blvecs, zenaz, freqs, bl_mp, conj=False):
    """psky (Nt, Nmp, Npp, Nf, P) float64/complex128 CPU -> (Npp, Nbl, Nt, Nf)"""
    out = []
    for t in range(psky.shape[0]):
        fr = orc.gen_fringe(blvecs, zenaz[t, 0], zenaz[t, 1], freqs, conj=conj)    # (Nbl, Nf, P)
        sel = psky[t][torch.as_tensor(bl_mp)]                                       # (Nbl, Npp, Nf, P)
        out.append(torch.einsum('bfp,bqfp->qbf', fr, sel.to(fr.dtype)))
    return torch.stack(out, dim=2)


def make_case(seed, Nbl, Nt, Nf, P, Nmp, Npp, cplx, uniform=True, blen=60.0):
    rng = np.random.default_rng(seed)
    blvecs = rng.normal(0, blen, (Nbl, 3))
    blvecs[:, 2] *= 0.05
    if uniform == 'f32grid':
        # a float32-rounded linspace (what torch.linspace gives under the float32 default dtype):
        # uniform only to ~8 Hz -> the kernels' "near-uniform" recurrence + correction path
        freqs = np.linspace(120e6, 180e6, Nf).astype(np.float32).astype(np.float64)
    elif uniform:
        freqs = np.linspace(120e6, 180e6, Nf)
    else:
        freqs = np.sort(rng.uniform(100e6, 200e6, Nf))
    zen = np.rad2deg(np.arccos(rng.uniform(0.0, 1.0, (Nt, P))))
    az = rng.uniform(0, 360, (Nt, P))
    zenaz = np.stack([zen, az], axis=1)
    psky = rng.normal(size=(Nt, Nmp, Npp, Nf, P))
    if cplx:
        psky = psky + 1j * rng.normal(size=psky.shape)
    bl_mp = rng.integers(0, Nmp, Nbl)
    return T64(blvecs), T64(freqs), T64(zenaz), torch.as_tensor(psky), bl_mp


def to_gpu_geometry(ops, blvecs, freqs, zenaz, bl_mp, Nmp, conj=False):
    Nt, _, P = zenaz.shape
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    return ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, bl_mp=bl_mp, Nmp=Nmp, conj=conj), Ps


def pad_psky(psky, Ps):
    out = torch.zeros(psky.shape[:-1] + (Ps,), dtype=psky.dtype)
    out[..., :psky.shape[-1]] = psky
    return out


CONFIGS = [(1, False), (2, False), (1, True), (4, False), (4, True)]


@pytest.mark.parametrize('Npp,cplx', CONFIGS)
@pytest.mark.parametrize('dtype', ['f64', 'f32'])
@pytest.mark.parametrize('uniform', [True, 'f32grid', False])
def test_fringe_sum_fwd_bwd(ops, Npp, cplx, dtype, uniform):
    Nmp = 3 if (Npp, cplx) in [(1, True), (4, True)] else 1
    blvecs, freqs, zenaz, psky, bl_mp = make_case(11 + Npp, Nbl=77, Nt=2, Nf=19, P=333,
                                                  Nmp=Nmp, Npp=Npp, cplx=cplx, uniform=uniform)
    geom, Ps = to_gpu_geometry(ops, blvecs, freqs, zenaz, bl_mp, Nmp)
    assert geom.uniform == {True: 1, 'f32grid': 2, False: 0}[uniform]
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, bl_mp)
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()

    rdt = torch.float64 if dtype == 'f64' else torch.float32
    cdt = torch.complex128 if dtype == 'f64' else torch.complex64
    x = pad_psky(psky, Ps).to(cdt if cplx else rdt).cuda().requires_grad_(True)
    vis = ops.fringe_sum(x, geom)
    assert vis.shape == ref.shape and vis.dtype == cdt
    tol_f, tol_g = (1e-11, 1e-11) if dtype == 'f64' else (1e-5, 1e-4)
    if uniform == 'f32grid' and dtype == 'f64':
        tol_f = tol_g = 1e-9          # neglected second-order term of the channel correction
    assert relmax(vis, ref) < tol_f
    (vis * gv.to(cdt).cuda().conj()).real.sum().backward()
    g = x.grad[..., :psky.shape[-1]]
    assert relmax(g, ref_in.grad) < tol_g
    assert x.grad[..., psky.shape[-1]:].abs().max() < 1e30        # padded columns: finite


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_fringe_sum_long_baselines_and_conj(ops, dtype):
    """km baselines: step/channel > 0.3 turn -> standard rotation path; conj=True sign"""
    blvecs, freqs, zenaz, psky, bl_mp = make_case(3, Nbl=130, Nt=1, Nf=64, P=200, Nmp=1, Npp=1,
                                                  cplx=False, blen=2500.0)
    geom, Ps = to_gpu_geometry(ops, blvecs, freqs, zenaz, bl_mp, 1, conj=True)
    assert geom.max_blen * geom.df / 2.99792458e8 > 0.3
    ref = oracle_fringe_sum(psky, blvecs, zenaz, freqs, bl_mp, conj=True)
    rdt = torch.float64 if dtype == 'f64' else torch.float32
    vis = ops.fringe_sum(pad_psky(psky, Ps).to(rdt).cuda(), geom)
    assert relmax(vis, ref) < (1e-10 if dtype == 'f64' else 1e-5)


def test_fringe_sum_strided_psky(ops):
    """time-inner (Npp, Nmp, Nf, Nt, P) storage passed as a permuted view: read in place"""
    blvecs, freqs, zenaz, psky, bl_mp = make_case(21, Nbl=40, Nt=3, Nf=20, P=150, Nmp=2, Npp=1,
                                                  cplx=True)
    geom, Ps = to_gpu_geometry(ops, blvecs, freqs, zenaz, bl_mp, 2)
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, bl_mp)
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape)) + 0j)
    (ref * gv.conj()).real.sum().backward()
    base = pad_psky(psky, Ps).permute(2, 1, 3, 0, 4).contiguous().cuda().requires_grad_(True)   # (Npp,Nmp,Nf,Nt,Ps)
    view = base.permute(3, 1, 0, 2, 4)                                                         # (Nt,Nmp,Npp,Nf,Ps)
    assert not view.is_contiguous()
    vis = ops.fringe_sum(view, geom)
    assert relmax(vis, ref) < 1e-11
    (vis * gv.cuda().conj()).real.sum().backward()
    g = base.grad.permute(3, 1, 0, 2, 4)[..., :psky.shape[-1]]
    assert relmax(g, ref_in.grad) < 1e-11


def make_antenna_case(seed, Nant, Nt, Nf, P, frac=1.0, autos=0, conj=False):
    """baselines as antenna pairs (random orientation, optional subset / autos) + beam-like psky"""
    rng = np.random.default_rng(seed)
    ant = rng.normal(0, 80.0, (Nant, 3))
    ant[:, 2] *= 0.02
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    pairs = [p if rng.random() < 0.5 else p[::-1] for p in pairs]
    if frac < 1.0:
        keep = rng.random(len(pairs)) < frac
        pairs = [p for p, k in zip(pairs, keep) if k]
    pairs += [(a, a) for a in range(autos)]
    order = rng.permutation(len(pairs))
    pairs = [pairs[k] for k in order]
    blvecs = np.stack([ant[b] - ant[a] for a, b in pairs])
    freqs = np.linspace(120e6, 180e6, Nf)
    zen = np.rad2deg(np.arccos(rng.uniform(0.0, 1.0, (Nt, P))))
    az = rng.uniform(0, 360, (Nt, P))
    env = np.exp(-13.8 * rng.uniform(size=(Nt, 1, 1, Nf, P)))            # six decades, like beam x sky
    psky = rng.normal(size=(Nt, 1, 1, Nf, P)) * env * 3e-5
    return T64(ant), pairs, T64(blvecs), T64(freqs), T64(np.stack([zen, az], axis=1)), torch.as_tensor(psky)


@pytest.mark.parametrize('Nant,frac,autos,force', [(70, 1.0, 3, 'auto'), (128, 0.6, 0, 'auto'),
                                                    (10, 1.0, 2, True), (33, 0.9, 0, True),
                                                    # 33..48 antennas: the packed-second-row-tile forward kernel (round 4)
                                                    (34, 1.0, 0, 'auto'), (37, 1.0, 2, 'auto'), (41, 0.8, 0, 'auto'),
                                                    (45, 0.7, 45, 'auto'), (48, 1.0, 1, 'auto'), (49, 1.0, 0, 'auto')])
@pytest.mark.parametrize('conj', [False, True])
def test_fringe_sum_matrix_core_path(ops, Nant, frac, autos, force, conj):
    """antenna-factored MFMA kernels (float32 1-pol): visibilities and psky gradient against the
    fp64 oracle of the baseline formulation; partial pair sets, both pair orientations,
    autocorrelations, antenna counts that do not fill the 32-wide tiles"""
    ant, pairs, blvecs, freqs, zenaz, psky = make_antenna_case(Nant, Nant, Nt=2, Nf=11, P=700,
                                                               frac=frac, autos=autos)
    Nt, _, P = zenaz.shape
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, conj=conj, antpos=ant.cuda(),
                              bl_ants=pairs, mfma=force)
    assert geom.ant is not None
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, [0] * len(pairs), conj=conj)
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    x = pad_psky(psky, Ps).float().cuda().requires_grad_(True)
    prof = []
    ops.PROFILE = prof
    try:
        vis = ops.fringe_sum(x, geom)
        assert relmax(vis, ref) < 1e-5
        (vis * gv.to(torch.complex64).cuda().conj()).real.sum().backward()
    finally:
        ops.PROFILE = None
    assert [k[0] for k in prof] == ['fringe_ant_fwd_kernel', 'fringe_ant_bwd_kernel']   # the MFMA kernels ran
    assert relmax(x.grad[..., :P], ref_in.grad) < 1e-4
    # same answer as the baseline-formulation kernels
    geom2 = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, conj=conj)
    assert geom2.ant is None
    vis2 = ops.fringe_sum(x.detach(), geom2)
    assert relmax(vis, vis2.cpu().numpy()) < 1e-5
    # float64 / complex / multi-pol inputs keep using the baseline-formulation kernels
    v64 = ops.fringe_sum(pad_psky(psky, Ps).cuda(), geom)
    assert relmax(v64, ref) < 1e-11


@pytest.mark.parametrize('Nant', [33, 37, 40, 44, 48])
def test_fringe_sum_packed_forward_kernel_non_negative_rows_and_repeat(ops, Nant):
    """33..48 antennas (fringe_ant_fwd_packed_kernel): a non-negative psky (rows of the mask-free instantiation) mixed with
    signed rows, several pixel splits with an odd number of panels, against the float64 oracle and the float64
    vector-ALU kernels; bitwise identical when repeated"""
    ant, pairs, blvecs, freqs, zenaz, psky = make_antenna_case(1000 + Nant, Nant, Nt=2, Nf=7, P=17000 + 32 * Nant, frac=1.0)
    psky = psky.abs()
    psky[1, :, :, 2] *= -1.0                                  # one all-negative row
    psky[0, :, :, 4, ::3] *= -1.0                             # one mixed row
    Nt, _, P = zenaz.shape
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, antpos=ant.cuda(), bl_ants=pairs, mfma='auto')
    assert geom.ant is not None and len(geom.ant['blocks']) == 1
    x = pad_psky(psky, Ps).float().cuda()
    vis = ops.fringe_sum(x, geom)
    v64 = ops.fringe_sum(pad_psky(psky, Ps).cuda(), ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs))
    assert relmax(vis, v64.cpu().numpy()) < 1e-5
    ref = oracle_fringe_sum(psky[:, :, :, :2], blvecs[:60], zenaz, freqs[:2], [0] * 60)
    assert relmax(vis[:, :60, :, :2], ref) < 1e-5
    assert torch.equal(vis, ops.fringe_sum(x, geom))


@pytest.mark.parametrize('Npp,cplx', [(2, False), (1, True), (4, True)])
@pytest.mark.parametrize('conj', [False, True])
def test_fringe_sum_matrix_core_polarised(ops, Npp, cplx, conj):
    """multi-pol / complex psky on the antenna-factored kernels (one launch per real plane),
    forward and backward against the fp64 oracle, with a time-inner strided psky view"""
    ant, pairs, blvecs, freqs, zenaz, _ = make_antenna_case(50, 50, Nt=2, Nf=7, P=1500, frac=0.8, autos=2)
    rng = np.random.default_rng(11)
    Nt, _, P = zenaz.shape
    psky = rng.normal(size=(Nt, 1, Npp, 7, P)) * np.exp(-9.0 * rng.uniform(size=(Nt, 1, Npp, 7, P)))
    if cplx:
        psky = psky + 1j * rng.normal(size=psky.shape) * np.exp(-9.0 * rng.uniform(size=psky.shape))
    psky = torch.as_tensor(psky)
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, conj=conj, antpos=ant.cuda(),
                              bl_ants=pairs, mfma=True)
    assert geom.ant is not None
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, [0] * len(pairs), conj=conj)
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    cdt = torch.complex64 if cplx else torch.float32
    # layout as RIME builds it: (Npp, Nmp, Nf, Nt, Ps) buffer viewed as (Nt, Nmp, Npp, Nf, Ps)
    buf = pad_psky(psky, Ps).to(cdt).permute(2, 1, 3, 0, 4).contiguous().cuda().requires_grad_(True)
    x = buf.permute(3, 1, 0, 2, 4)
    prof = []
    ops.PROFILE = prof
    try:
        vis = ops.fringe_sum(x, geom)
        assert relmax(vis, ref) < 1e-5
        (vis * gv.to(torch.complex64).cuda().conj()).real.sum().backward()
    finally:
        ops.PROFILE = None
    assert [k[0] for k in prof] == ['fringe_ant_fwd_kernel', 'fringe_ant_bwd_kernel']
    gx = buf.grad.permute(3, 1, 0, 2, 4)
    assert relmax(gx[..., :P], ref_in.grad) < 1e-4


@pytest.mark.parametrize('Nant,frac,Npp', [(200, 0.3, 1), (260, 0.2, 1), (150, 0.5, 2)])
def test_fringe_sum_matrix_core_antenna_groups(ops, Nant, frac, Npp):
    """more than 128 antennas: groups of 128, diagonal + cross blocks (both pair orientations in
    the cross blocks, a last group that does not fill its tiles), forward and backward"""
    ant, pairs, blvecs, freqs, zenaz, _ = make_antenna_case(Nant, Nant, Nt=2, Nf=3, P=400, frac=frac, autos=3)
    rng = np.random.default_rng(3)
    Nt, _, P = zenaz.shape
    psky = torch.as_tensor(rng.normal(size=(Nt, 1, Npp, 3, P)) * np.exp(-9.0 * rng.uniform(size=(Nt, 1, Npp, 3, P))))
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, antpos=ant.cuda(), bl_ants=pairs, mfma=True)
    assert geom.ant is not None
    ngroups = (Nant + 127) // 128
    assert ngroups <= len(geom.ant['blocks']) <= ngroups * (ngroups + 1) // 2   # empty blocks are skipped
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, [0] * len(pairs))
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    x = pad_psky(psky, Ps).float().cuda().requires_grad_(True)
    vis = ops.fringe_sum(x, geom)
    assert relmax(vis, ref) < 1e-5
    (vis * gv.to(torch.complex64).cuda().conj()).real.sum().backward()
    assert relmax(x.grad[..., :P], ref_in.grad) < 1e-4


def _ant_setup(ops, Nant, Nt, Nf, P, frac, autos, seed=None, orient=None):
    ant, pairs, blvecs, freqs, zenaz, _ = make_antenna_case(seed or Nant, Nant, Nt=Nt, Nf=Nf, P=P, frac=frac, autos=autos)
    if orient is not None:                        # force one orientation: all i < j ('up') or all i > j ('down')
        pairs = [(min(p), max(p)) if orient == 'up' else (max(p), min(p)) for p in pairs]
        blvecs = torch.stack([ant[b] - ant[a] for a, b in pairs])
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    return ant, pairs, blvecs, freqs, zenaz, sdir, Ps


def _check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, bl_mp, P, Ps, cplx, conj=False, tv=1e-5, tg=1e-4):
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, bl_mp, conj=conj)
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    x = pad_psky(psky, Ps).to(torch.complex64 if cplx else torch.float32).cuda().requires_grad_(True)
    prof = []
    ops.PROFILE = prof
    try:
        vis = ops.fringe_sum(x, geom)
        (vis * gv.to(torch.complex64).cuda().conj()).real.sum().backward()
    finally:
        ops.PROFILE = None
    assert [k[0] for k in prof] == ['fringe_ant_fwd_kernel', 'fringe_ant_bwd_kernel']
    assert relmax(vis, ref) < tv
    assert relmax(x.grad[..., :P], ref_in.grad) < tg
    return vis


def _symmetric_array(kind, rng):
    """antenna positions with point symmetry (and some antennas without a partner), centre away from the origin"""
    from bayeslim_amd import utils
    c = np.array([31.7, -12.3, 4.1])
    if kind.startswith('hex'):
        side, extra = {'hex19': (3, 0), 'hex37': (4, 0), 'hex61': (5, 0), 'hex91': (6, 0), 'hex127': (7, 0), 'hex127+1': (7, 1)}[kind.rstrip('t')]
        ant = utils._make_hex(side, D=14.6)[1]
        if extra:
            ant = np.vstack([ant, [[250.0, 3.0, 0.0]]])
        if kind.endswith('t'):                                   # tilted: the plane of the array is not z = const (no `flat` licence)
            t = np.deg2rad(3.0)
            ant = ant @ np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]]).T
    else:
        # `half` random antennas, their mirror images, `single` antennas without a partner; tilted (z matters)
        half, single = {'rand45': (20, 5), 'rand70': (33, 4), 'rand100': (45, 10), 'rand128': (60, 8)}[kind]
        h = rng.normal(0, 70.0, (half, 3)) * [1, 1, 0.05]
        ant = np.vstack([h, -h, rng.normal(0, 70.0, (single, 3)) * [1, 1, 0.05]])
    ant = ant[rng.permutation(len(ant))] + c
    return ant


@pytest.mark.parametrize('kind,groups', [('hex19', (2, 2)), ('hex37', (2, 3)), ('hex61', (4, 4)), ('hex91', (6, 6)), ('hex127+1', (7, 8)),
                                         ('rand45', (2, 3)), ('rand70', (5, 5)), ('rand128', (7, 8))])
@pytest.mark.parametrize('conj', [False, True])
def test_fringe_sum_mirror_pairs(ops, kind, groups, conj, monkeypatch):
    """arrays with point symmetry (round 5): the antennas with r' - c = -(r - c) are found on the host, a diagonal block's rows
    are ordered so that the second octet of a 16-row group holds the mirror antennas of the first, and the kernels use the
    conjugate of the first octet's phasors instead of evaluating them -- forward (1 / 2 / 3 / 4 row tiles, the packed 33..48
    shape) and backward, against the float64 oracle of the baseline formulation; both pair
    orientations, a partial pair set, autocorrelations, antennas without a partner, a centre away from the origin; equal
    to 2e-6 -- not bitwise -- to the run without the pairing (RIME_MIRROR=0), whose geometry has no mirrored blocks"""
    rng = np.random.default_rng(abs(hash(kind)) % 1000)
    ant = _symmetric_array(kind, rng)
    Nant, Nt, Nf, P = len(ant), 2, 7, 700
    pairs = [(i, j) if rng.random() < 0.5 else (j, i) for i in range(Nant) for j in range(i + 1, Nant) if rng.random() < 0.9]
    pairs += [(a, a) for a in range(3)]
    pairs = [pairs[k] for k in rng.permutation(len(pairs))]
    blvecs = T64(np.stack([ant[b] - ant[a] for a, b in pairs]))
    freqs = T64(np.linspace(120e6, 180e6, Nf))
    zenaz = T64(np.stack([np.rad2deg(np.arccos(rng.uniform(0, 1, (Nt, P)))), rng.uniform(0, 360, (Nt, P))], axis=1))
    psky = torch.as_tensor(rng.normal(size=(Nt, 1, 1, Nf, P)) * np.exp(-9.0 * rng.uniform(size=(Nt, 1, 1, Nf, P))))
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    res = {}
    monkeypatch.setattr(ops, 'PAIR', False)                  # (the conjugate-pair form has a test of its own below)
    for on in (True, False):
        monkeypatch.setattr(ops, 'MIRROR', on)
        geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, conj=conj, antpos=T64(ant).cuda(), bl_ants=pairs, mfma=True)
        assert geom.ant is not None and ('blocks_mirror' in geom.ant) == on
        if on:
            assert geom.ant['mirror_groups'] == [groups], geom.ant['mirror_groups']
            blk = geom.ant['blocks_mirror'][0]
            pos, n = blk['pos'].cpu().numpy(), blk['nrows']
            rows = blk['rows'] + [-1] * 16
            for g in range((n + 15) // 16):                      # the layout the kernels rely on
                if (blk['mirror'] >> g) & 1:
                    for i in range(8):
                        a, b = rows[16 * g + i], rows[16 * g + 8 + i]
                        assert b < 0 or (a >= 0 and np.abs(pos[16 * g + i] + pos[16 * g + 8 + i]).max() < 1e-9)
        res[on] = _check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, [0] * len(pairs), P, Ps, False, conj=conj).detach()
    assert not torch.equal(res[True], res[False])
    assert relmax(res[True], res[False].cpu().numpy()) < 2e-6


@pytest.mark.parametrize('kind,pairs_rows_hub', [('hex91', (45, 46, 0)), ('hex127', (63, 64, 0)), ('hex127+1', (63, 64, 1)),
                                                 ('rand70', (33, 37, 0)), ('rand100', (45, 55, 0)), ('rand128', None),
                                                 ('hex37', (18, 19, 0)), ('hex61', (30, 31, 0)), ('rand45', (20, 25, 0)), ('hex19', None),
                                                 ('hex127+1t', (63, 64, 1)), ('hex37t', (18, 19, 0))])
@pytest.mark.parametrize('conj', [False, True])
@pytest.mark.parametrize('full', [False, True])
def test_fringe_sum_conjugate_pairs(ops, kind, pairs_rows_hub, conj, full, monkeypatch):
    """conjugate-pair form (round 5): blocks of more than 64 antennas of a point-symmetric array run on the images of ONE antenna
    of every mirror pair -- A = X^H s X and B = X^T s X from the same three real products (forward), the four real planes
    N1..N4 (backward), the hub of a full block on the vector ALU -- against the float64 oracle of the baseline formulation:
    46 / 64 / 37 / 55 rows, with and without the hub path, and 19 / 31 / 25 rows on the one-tile kernel (33..64 antennas), both
    pair orientations and fringe signs, coplanar arrays (the `flat` licence: no z term in the phase) and tilted ones, a partial
    pair set with autocorrelations and the full set; an array whose firsts and
    singles do not fit into 64 rows (60 pairs + 8 singles) and one of up to 32 antennas keep the mirror-pair kernels.  Equal to 2e-6 to the run on those kernels (RIME_PAIR=0)."""
    rng = np.random.default_rng(abs(hash(kind)) % 1000 + 7)
    ant = _symmetric_array(kind, rng)
    Nant, Nt, Nf, P = len(ant), 2, 5, 700
    if full:
        pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    else:
        pairs = [(i, j) if rng.random() < 0.5 else (j, i) for i in range(Nant) for j in range(i + 1, Nant) if rng.random() < 0.9]
        hub = int(np.argmin(np.abs(ant - ant.mean(0)).sum(1))) if kind.startswith('hex127+1') else -1
        pairs += [(a, a) for a in range(Nant) if a % 40 == 3 and a != hub]       # (the hub's autocorrelation declines the form)
        pairs = [pairs[k] for k in rng.permutation(len(pairs))]
    blvecs = T64(np.stack([ant[b] - ant[a] for a, b in pairs]))
    freqs = T64(np.linspace(120e6, 180e6, Nf))
    zenaz = T64(np.stack([np.rad2deg(np.arccos(rng.uniform(0, 1, (Nt, P)))), rng.uniform(0, 360, (Nt, P))], axis=1))
    psky = rng.normal(size=(Nt, 1, 1, Nf, P)) * np.exp(-9.0 * rng.uniform(size=(Nt, 1, 1, Nf, P)))
    psky[1, :, :, 2] = np.abs(psky[1, :, :, 2])              # a row without a negative value: the mask-free instantiation
    psky = torch.as_tensor(psky)
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, 'PAIR', on)
        geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, conj=conj, antpos=T64(ant).cuda(), bl_ants=pairs, mfma=True)
        assert geom.ant is not None
        if on and pairs_rows_hub is not None:
            assert geom.ant['pair_blocks'] == [pairs_rows_hub], geom.ant['pair_blocks']
            blk = geom.ant['blocks_real'][0]
            assert blk['pair'] == 1 and (blk['centre'] is not None) == bool(pairs_rows_hub[2]) and geom.ant['mirror_groups'] == []
            # coplanar arrays (the hexagons as generated, z = const) carry the `flat` licence, tilted and random ones do not
            assert blk['flat'] == int(kind.startswith('hex') and not kind.endswith('t')), (kind, blk['flat'])
            pos = blk['pos'].cpu().numpy()                       # rows: firsts and singles, measured from the centre
            for k, (a, b) in enumerate(zip(blk['firsts'], blk['partner'])):
                if b >= 0:
                    assert np.abs((ant[a] - pos[k]) - (ant[b] + pos[k])).max() < 1e-9
        else:
            assert not geom.ant.get('pair_blocks')
        res[on] = _check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, [0] * len(pairs), P, Ps, False, conj=conj).detach()
    if pairs_rows_hub is not None:
        assert not torch.equal(res[True], res[False])
    assert relmax(res[True], res[False].cpu().numpy()) < 2e-6


@pytest.mark.parametrize('kind,pairs_rows_hub', [('hex37', (18, 19, 0)), ('hex91', (45, 46, 0)), ('hex127+1', (63, 64, 1)), ('rand100', (45, 55, 0))])
@pytest.mark.parametrize('conj', [False, True])
def test_fringe_sum_conjugate_pairs_complex_psky(ops, kind, pairs_rows_hub, conj, monkeypatch):
    """a COMPLEX psky (Jones layouts) on a point-symmetric array: V is linear in psky, so a block with the conjugate-pair form
    takes one pair pass per real plane (forward: the second plane's visibilities enter as i V; backward: the imaginary plane's
    gradient from -i g) in place of the one-pass self block -- against the float64 oracle, both fringe signs, mixed pair
    orientations (the plain blocks would need two passes as well), and equal to 2e-6 to the plain blocks (RIME_PAIR_CPLX=0)"""
    rng = np.random.default_rng(abs(hash(kind)) % 1000 + 11)
    ant = _symmetric_array(kind, rng)
    Nant, Nt, Nf, P = len(ant), 2, 4, 500
    hub = int(np.argmin(np.abs(ant - ant.mean(0)).sum(1))) if kind.startswith('hex127+1') else -1
    pairs = [(i, j) if rng.random() < 0.5 else (j, i) for i in range(Nant) for j in range(i + 1, Nant) if rng.random() < 0.9]
    pairs += [(a, a) for a in range(Nant) if a % 30 == 2 and a != hub]
    pairs = [pairs[k] for k in rng.permutation(len(pairs))]
    blvecs = T64(np.stack([ant[b] - ant[a] for a, b in pairs]))
    freqs = T64(np.linspace(120e6, 180e6, Nf))
    zenaz = T64(np.stack([np.rad2deg(np.arccos(rng.uniform(0, 1, (Nt, P)))), rng.uniform(0, 360, (Nt, P))], axis=1))
    shape = (Nt, 1, 2, Nf, P)                                   # two polarisation products
    psky = torch.as_tensor(rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape))
                           + 1j * rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape)))
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, 'PAIR_CPLX', on)
        geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, conj=conj, antpos=T64(ant).cuda(), bl_ants=pairs, mfma=True)
        assert geom.ant['pair_blocks'] == [pairs_rows_hub] and ('blocks_cplx' in geom.ant) == on
        if on:
            assert geom.ant['blocks_cplx'][0]['pair'] == 1 and float(geom.ant['two_pass_mask_cplx'].sum()) == len(pairs)
        res[on] = _check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, [0] * len(pairs), P, Ps, True, conj=conj).detach()
    assert not torch.equal(res[True], res[False])
    assert relmax(res[True], res[False].cpu().numpy()) < 2e-6


@pytest.mark.parametrize('hub', [True, False])
def test_conjugate_pair_kernels_at_the_headline_size_are_repeatable(ops, hub, monkeypatch):
    """the conjugate-pair kernels are the first here whose blocks share a CU (three per CU): at the headline size -- 98 304
    directions, enough blocks that many are dispatched while others stream MFMAs on the same SIMDs -- both directions must
    be bit-identical from run to run and agree with the mirror-pair kernels.  (Round 5: packed-f32 adds in the backward's
    staging code gave wrong G planes in a fraction of the late blocks, different every run; csrc/fringe_mfma.hip, keep_scalar.)"""
    from bayeslim_amd import utils
    ant = utils._make_hex(7, D=14.6)[1]
    if hub:
        ant = np.vstack([ant, [[250.0, 0.0, 0.0]]])
    n, Nt, Nf, P = len(ant), 2, 24, 98304
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    rng = np.random.default_rng(0)
    blvecs = T64(np.stack([ant[b] - ant[a] for a, b in pairs])).cuda()
    freqs = T64(np.linspace(120e6, 180e6, Nf))
    s = rng.normal(size=(Nt, 3, P))
    s /= np.linalg.norm(s, axis=1, keepdims=True)
    s[:, 2] = np.abs(s[:, 2])
    sdir = T64(s).cuda()
    g = torch.as_tensor(rng.normal(size=(1, len(pairs), Nt, Nf)) + 1j * rng.normal(size=(1, len(pairs), Nt, Nf))).to(torch.complex64).cuda()
    psky = torch.as_tensor(rng.normal(size=(Nt, 1, 1, Nf, P))).float().cuda()
    out = {}
    for pair in (True, False):
        monkeypatch.setattr(ops, 'PAIR', pair)
        geom = ops.FringeGeometry(blvecs, sdir, freqs, antpos=T64(ant).cuda(), bl_ants=pairs, mfma=True)
        assert bool(geom.ant.get('pair_blocks')) == pair
        bwd = [ops.fringe_adjoint(g, geom).clone() for _ in range(3)]
        fwd = [ops.fringe_sum(psky, geom).clone() for _ in range(3)]
        assert all(torch.equal(bwd[0], b) for b in bwd[1:]) and all(torch.equal(fwd[0], v) for v in fwd[1:])
        out[pair] = (bwd[0], fwd[0])
    for a, b in zip(out[True], out[False]):
        assert float((a - b).abs().max()) < 3e-6 * float(b.abs().max())


@pytest.mark.parametrize('Nant,group,frac', [(128, 32, 1.0), (100, 32, 0.7), (128, 64, 1.0), (90, 64, 0.8), (40, 32, 1.0)])
def test_fringe_sum_matrix_core_small_groups(ops, Nant, group, frac):
    """groups of 32 / 64 antennas (rank-local tile shards): one-tile diagonal blocks with the K-split
    wave deal and the (32,32) / (32,64) / (64,64) cross shapes, both pair orientations, ragged last group"""
    ant, pairs, blvecs, freqs, zenaz, sdir, Ps = _ant_setup(ops, Nant, 2, 5, 900, frac, 2)
    rng = np.random.default_rng(4)
    P = zenaz.shape[-1]
    psky = torch.as_tensor(rng.normal(size=(2, 1, 1, 5, P)) * np.exp(-9.0 * rng.uniform(size=(2, 1, 1, 5, P))))
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, antpos=ant.cuda(), bl_ants=pairs, mfma=True, group=group)
    assert geom.ant is not None
    ng = (Nant + group - 1) // group
    assert len(geom.ant['blocks']) == ng * (ng + 1) // 2
    shapes = {(b['cross'], b['nrows'] - b['cross']) for b in geom.ant['blocks'] if b['cross']}
    assert shapes <= {(32, 32), (32, 64), (64, 64)} and shapes
    vis = _check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, [0] * len(pairs), P, Ps, False)
    # identical to the 128-antenna grouping up to float32 summation order
    geom0 = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, antpos=ant.cuda(), bl_ants=pairs, mfma=True)
    v0 = ops.fringe_sum(pad_psky(psky, Ps).float().cuda(), geom0)
    assert relmax(vis, v0) < 2e-6


@pytest.mark.parametrize('Nant,nmodel,Npp,cplx', [(60, 2, 1, False), (90, 3, 1, False), (70, 2, 2, False), (64, 2, 1, True),
                                                  (200, 2, 1, False)])
def test_fringe_sum_matrix_core_beam_models(ops, Nant, nmodel, Npp, cplx):
    """several beam models (beam_model.py:303-327): one block per (group pair, model pair), each on its own
    psky plane -- forward and backward against the fp64 oracle, no vector-ALU fallback"""
    ant, pairs, blvecs, freqs, zenaz, sdir, Ps = _ant_setup(ops, Nant, 2, 4, 600, 0.9, 2)
    rng = np.random.default_rng(7)
    P = zenaz.shape[-1]
    ant_model = [int(rng.integers(0, nmodel)) if a % 5 else a % nmodel for a in range(Nant)]
    uniq = sorted({(ant_model[a], ant_model[b]) for a, b in pairs})
    bl_mp = [uniq.index((ant_model[a], ant_model[b])) for a, b in pairs]
    shape = (2, len(uniq), Npp, 4, P)
    psky = rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape))
    if cplx:
        psky = psky + 1j * rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape))
    psky = torch.as_tensor(psky)
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, bl_mp=bl_mp, Nmp=len(uniq), antpos=ant.cuda(),
                              bl_ants=pairs, mfma=True, mp_pairs=uniq)
    assert geom.ant is not None and geom.ant['multi_model']
    assert {b['mp'] for b in geom.ant['blocks']} == set(range(len(uniq)))
    _check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, bl_mp, P, Ps, cplx)


def test_fringe_sum_single_nonzero_plane_without_model_table(ops):
    """Nmp > 1 planes, every baseline on plane 1, no mp_pairs table -- the matrix-core blocks would read
    plane 0; the geometry must decline that path (or read the right plane): result against the fp64 oracle"""
    ant, pairs, blvecs, freqs, zenaz, sdir, Ps = _ant_setup(ops, 48, 1, 4, 300, 1.0, 0)
    rng = np.random.default_rng(11)
    P = zenaz.shape[-1]
    bl_mp = [1] * len(pairs)
    psky = torch.as_tensor(rng.normal(size=(1, 2, 1, 4, P)))
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, bl_mp=bl_mp, Nmp=2, antpos=ant.cuda(), bl_ants=pairs, mfma=True)
    ref = oracle_fringe_sum(psky, blvecs, zenaz, freqs, bl_mp)
    x = pad_psky(psky, Ps).float().cuda().requires_grad_(True)
    vis = ops.fringe_sum(x, geom)
    assert relmax(vis, ref) < 1e-5
    g = torch.as_tensor(rng.normal(size=tuple(vis.shape)) + 1j * rng.normal(size=tuple(vis.shape)))
    (vis * g.to(vis.dtype).cuda().conj()).real.sum().backward()
    assert float(x.grad[:, 0].abs().max()) == 0.0 and float(x.grad[:, 1].abs().max()) > 0.0


@pytest.mark.parametrize('Nant,orient,Npp', [(200, 'up', 1), (200, 'down', 2), (200, None, 1), (300, 'up', 4), (100, 'up', 1)])
@pytest.mark.parametrize('conj', [False, True])
def test_fringe_sum_matrix_core_complex_single_pass(ops, Nant, orient, Npp, conj):
    """complex (Jones) psky: cross blocks with one pair orientation contract it in ONE pass (forward), every
    one-orientation block writes both gradient planes from one backward pass; mixed blocks take two passes"""
    ant, pairs, blvecs, freqs, zenaz, sdir, Ps = _ant_setup(ops, Nant, 2, 3, 500, 0.5, 0, orient=orient)
    rng = np.random.default_rng(8)
    P = zenaz.shape[-1]
    shape = (2, 1, Npp, 3, P)
    psky = torch.as_tensor(rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape))
                           + 1j * rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape)))
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, conj=conj, antpos=ant.cuda(), bl_ants=pairs, mfma=True)
    assert geom.ant is not None
    cp = [b['cpass'] for b in geom.ant['blocks']]
    if orient == 'up':
        assert all(c == 1 for c in cp)
    elif orient == 'down':
        assert all((c == -1) == bool(b['cross']) for c, b in zip(cp, geom.ant['blocks']))     # diagonal blocks: mixed tables
    _check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, [0] * len(pairs), P, Ps, True, conj=conj)
    if orient == 'up':
        # diagonal blocks of 1, 2 or 4 row tiles run as triangular self-cross blocks in the complex forward
        for b in geom.ant['blocks']:
            if not b['cross']:
                assert b['self_pos'] is not None and b['fwd_cpass'] == 1


@pytest.mark.parametrize('Nant,group', [(60, 32), (64, 64), (128, 128), (37, 128), (90, 128)])
def test_fringe_sum_self_blocks_equal_two_real_passes(ops, Nant, group):
    """complex psky, forward: a diagonal block as ONE self-cross pass (1 to 4 row tiles) against the two real-plane
    passes of the diagonal kernel and against the fp64 oracle"""
    ant, pairs, blvecs, freqs, zenaz, sdir, Ps = _ant_setup(ops, Nant, 2, 3, 700, 1.0, 0, orient='up')
    rng = np.random.default_rng(9)
    P = zenaz.shape[-1]
    shape = (2, 1, 2, 3, P)
    psky = torch.as_tensor(rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape))
                           + 1j * rng.normal(size=shape) * np.exp(-9.0 * rng.uniform(size=shape)))
    out = []
    for flag in (True, False):
        ops.SELF_BLOCKS = flag
        try:
            geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, antpos=ant.cuda(), bl_ants=pairs, mfma=True, group=group)
        finally:
            ops.SELF_BLOCKS = True
        diag = [b for b in geom.ant['blocks'] if not b['cross']]
        assert all((b['self_pos'] is not None) == flag for b in diag) and diag
        out.append(_check_ant_path(ops, geom, psky, blvecs, zenaz, freqs, [0] * len(pairs), P, Ps, True))
    assert relmax(out[0], out[1]) < 3e-6


def test_fringe_sum_full_size_properties(ops):
    """BASELINE config 4 at full size (128 antennas / 8128 baselines, 256 channels, 98 304 visible
    pixels): size-independent properties -- linearity, the adjoint identity Re<V(x), G> == <x, V^T(G)>
    between the forward and backward kernels, agreement of the antenna-factored (matrix-core) and
    baseline-formulation (vector-ALU) kernels -- plus a sampled float64 oracle comparison of visibilities
    and gradient entries at this size"""
    rng = np.random.default_rng(0)
    Nant, Nf, P = 128, 256, 98304
    ant = rng.normal(0, 80.0, (Nant, 3)); ant[:, 2] *= 0.02
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    blvecs = T64(np.stack([ant[b] - ant[a] for a, b in pairs])).cuda()
    cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P)
    sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz])[None]).cuda()
    freqs = torch.linspace(120e6, 180e6, Nf, dtype=torch.float64)
    gm = ops.FringeGeometry(blvecs, sdir, freqs, antpos=T64(ant).cuda(), bl_ants=pairs)
    gv = ops.FringeGeometry(blvecs, sdir, freqs, mfma=False)
    assert gm.ant is not None and gv.ant is None
    gen = torch.Generator(device='cuda').manual_seed(1)
    env = torch.exp(-9.0 * torch.rand(1, 1, 1, Nf, P, device='cuda', generator=gen))
    x1 = torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen) * env
    x2 = torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen) * env
    v1, v2 = ops.fringe_sum(x1, gm), ops.fringe_sum(x2, gm)
    scale = float(v1.abs().max())
    # linearity
    v12 = ops.fringe_sum(0.75 * x1 - 1.5 * x2, gm)
    assert float((v12 - (0.75 * v1 - 1.5 * v2)).abs().max()) < 2e-5 * scale
    # matrix-core vs vector-ALU kernels
    assert float((v1 - ops.fringe_sum(x1, gv)).abs().max()) < 1e-5 * scale
    # adjoint identity, both kernel families
    G = torch.randn(v1.shape, device='cuda', generator=gen) + 1j * torch.randn(v1.shape, device='cuda', generator=gen)
    for geom in (gm, gv):
        x = x1.clone().requires_grad_(True)
        v = ops.fringe_sum(x, geom)
        lhs = (v.detach() * G.conj()).real.double().sum()
        (v * G.conj()).real.sum().backward()
        rhs = (x.grad.double() * x1.double()).sum()
        assert abs(float(lhs - rhs)) < 1e-4 * abs(float(lhs)) + 1e-6
        gx = x.grad.detach()
        # SAMPLED oracle comparison at full size: float64 sums evaluated for a few (baseline, channel) visibilities over
        # all 98 304 pixels and a few (channel, pixel) gradient entries over all 8 128 baselines (torch float64 on the GPU:
        # the defining formula telescope_model.py:350-356 + rime_model.py:429, not the kernels)
        srng = np.random.default_rng(3)
        bsel = torch.as_tensor(srng.choice(len(pairs), 24, replace=False), device='cuda')
        fsel = torch.as_tensor(srng.choice(Nf, 5, replace=False), device='cuda')
        tau = blvecs[bsel] @ sdir[0]                                                        # (24, P) metres
        ph = 2j * np.pi / 2.99792458e8 * freqs.cuda()[fsel][None, :, None] * tau[:, None, :]
        ref = (torch.exp(ph) * x1[0, 0, 0][fsel].double()[None]).sum(-1)                    # (24, 5)
        got = v.detach()[0][bsel][:, 0][:, fsel]
        assert float((got - ref).abs().max()) < 1e-5 * scale
        psel = torch.as_tensor(srng.choice(P, 48, replace=False), device='cuda')
        tau = blvecs @ sdir[0][:, psel]                                                     # (Nbl, 48)
        ph = 2j * np.pi / 2.99792458e8 * freqs.cuda()[fsel][None, :, None] * tau[:, None, :]
        gref = (torch.exp(ph).conj() * G[0][:, 0][:, fsel].to(torch.complex128)[:, :, None]).real.sum(0)    # (5, 48)
        ggot = gx[0, 0, 0][fsel][:, psel]
        assert float((ggot - gref).abs().max()) < 1e-4 * float(gx.abs().max())


def test_fringe_sum_c5_size_complex_blocks_sampled_oracle(ops):
    """BASELINE config 5's array and pixel count (512 stations / 130 816 baselines, 393 216 pixels, complex psky; 4 of
    the 512 channels): every block kind of the complex single pass -- self blocks on the diagonal, (128, 128) cross
    blocks -- against float64 sums of the defining formula for sampled visibilities and gradient entries, plus the
    adjoint identity between the forward and backward kernels"""
    rng = np.random.default_rng(12)
    Nant, Nf, P = 512, 4, 393216
    ant = rng.normal(0, 300.0, (Nant, 3)); ant[:, 2] = 0.0
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    antp = T64(ant).cuda()
    i1 = torch.as_tensor([a for a, _ in pairs], device='cuda')
    i2 = torch.as_tensor([b for _, b in pairs], device='cuda')
    blvecs = antp[i2] - antp[i1]
    cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P)
    sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz])[None]).cuda()
    freqs = torch.linspace(120e6, 121e6, Nf, dtype=torch.float64)
    geom = ops.FringeGeometry(blvecs, sdir, freqs, antpos=antp, bl_ants=pairs)
    blocks = geom.ant['blocks']
    assert len(blocks) == 10 and sum(1 for b in blocks if b['self_pos'] is not None) == 4
    assert all(b['fwd_cpass'] == 1 and b['cpass'] == 1 for b in blocks)
    gen = torch.Generator(device='cuda').manual_seed(2)
    env = torch.exp(-9.0 * torch.rand(1, 1, 1, Nf, P, device='cuda', generator=gen))
    x1 = torch.complex(torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen),
                       torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen)) * env
    x = x1.clone().requires_grad_(True)
    v = ops.fringe_sum(x, geom)
    assert v.shape == (1, len(pairs), 1, Nf)
    scale = float(v.detach().abs().max())
    srng = np.random.default_rng(4)
    # one sampled baseline from every block: self blocks (same group) and cross blocks (different groups)
    grp = lambda a: a // 128
    want = {(gi, gj) for gi in range(4) for gj in range(gi, 4)}
    bsel = []
    for k in srng.permutation(len(pairs)):
        key = (grp(pairs[k][0]), grp(pairs[k][1]))
        if key in want:
            want.discard(key)
            bsel.append(int(k))
        if not want:
            break
    bsel += srng.choice(len(pairs), 14, replace=False).tolist()
    bsel = torch.as_tensor(bsel, device='cuda')
    tau = blvecs[bsel] @ sdir[0]
    ph = 2j * np.pi / 2.99792458e8 * freqs.cuda()[None, :, None] * tau[:, None, :]
    ref = (torch.exp(ph) * x1[0, 0, 0].to(torch.complex128)[None]).sum(-1)              # (24, Nf)
    assert float((v.detach()[0][bsel][:, 0] - ref).abs().max()) < 1e-5 * scale
    G = torch.complex(torch.randn(v.shape, device='cuda', generator=gen), torch.randn(v.shape, device='cuda', generator=gen))
    lhs = (v.detach() * G.conj()).real.double().sum()
    (v * G.conj()).real.sum().backward()
    rhs = (x.grad.conj() * x1).real.double().sum()
    assert abs(float(lhs - rhs)) < 1e-4 * abs(float(lhs)) + 1e-6
    psel = torch.as_tensor(srng.choice(P, 16, replace=False), device='cuda')
    tau = blvecs @ sdir[0][:, psel]                                                     # (Nbl, 16)
    gref = torch.zeros(Nf, 16, dtype=torch.complex128, device='cuda')
    for f in range(Nf):                                                                 # grad = sum_b conj(F) g
        ph = 2j * np.pi / 2.99792458e8 * float(freqs[f]) * tau
        gref[f] = (torch.exp(ph).conj() * G[0][:, 0, f].to(torch.complex128)[:, None]).sum(0)
    ggot = x.grad.detach()[0, 0, 0][:, psel]
    assert float((ggot - gref).abs().max()) < 1e-4 * float(x.grad.abs().max())
    # EVERY visibility and gradient entry against the baseline-formulation (vector-ALU) kernels, complex and real psky,
    # and run-to-run identity of the matrix-core gradients (a sporadic defect of the complex backward showed only here)
    gv = ops.FringeGeometry(blvecs, sdir, freqs, mfma=False)
    for xin in (x1, x1.real.contiguous()):
        grads = []
        for g_ in (geom, gv, geom):
            xx = xin.clone().requires_grad_(True)
            vv = ops.fringe_sum(xx, g_)
            (vv * G.conj()).real.sum().backward()
            grads.append((vv.detach(), xx.grad.detach()))
        assert relmax(grads[0][0], grads[1][0].cpu().numpy()) < 1e-5
        assert float((grads[0][1] - grads[1][1]).abs().max()) < 1e-5 * float(grads[1][1].abs().max())
        assert torch.equal(grads[0][1], grads[2][1]) and torch.equal(grads[0][0], grads[2][0])


@pytest.mark.parametrize('group,nmodel', [(32, 1), (64, 1), (128, 2)])
def test_fringe_sum_large_pixel_count_block_kinds(ops, group, nmodel):
    """393 216 pixels (the C5 sky) through the other block kinds -- one-tile diagonal blocks with the K-split wave deal
    and (32, 32) / (32, 64) / (64, 64) cross blocks of rank-local tile shards, blocks per beam-model pair --, real and
    complex psky: every visibility and gradient entry against the vector-ALU kernels, run-to-run identical"""
    rng = np.random.default_rng(13)
    Nant, Nf, P = 128, 2, 393216
    ant = rng.normal(0, 200.0, (Nant, 3)); ant[:, 2] = 0.0
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    antp = T64(ant).cuda()
    blvecs = antp[torch.as_tensor([b for _, b in pairs], device='cuda')] - antp[torch.as_tensor([a for a, _ in pairs], device='cuda')]
    cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P)
    sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz])[None]).cuda()
    freqs = torch.linspace(150e6, 151e6, Nf, dtype=torch.float64)
    kw = {}
    Nmp = 1
    if nmodel > 1:
        ant_model = [a % nmodel for a in range(Nant)]
        uniq = sorted({(ant_model[a], ant_model[b]) for a, b in pairs})
        kw = dict(bl_mp=[uniq.index((ant_model[a], ant_model[b])) for a, b in pairs], Nmp=len(uniq), mp_pairs=uniq)
        Nmp = len(uniq)
    gm = ops.FringeGeometry(blvecs, sdir, freqs, antpos=antp, bl_ants=pairs, mfma=True, group=group, **kw)
    gv = ops.FringeGeometry(blvecs, sdir, freqs, mfma=False, **{k: v for k, v in kw.items() if k != 'mp_pairs'})
    assert gm.ant is not None and gv.ant is None
    gen = torch.Generator(device='cuda').manual_seed(4)
    xc = torch.complex(torch.randn(1, Nmp, 1, Nf, P, device='cuda', generator=gen), torch.randn(1, Nmp, 1, Nf, P, device='cuda', generator=gen))
    G = None
    for xin in (xc, xc.real.contiguous()):
        res = []
        for g_ in (gm, gv, gm):
            xx = xin.clone().requires_grad_(True)
            vv = ops.fringe_sum(xx, g_)
            if G is None:
                G = torch.complex(torch.randn(vv.shape, device='cuda', generator=gen), torch.randn(vv.shape, device='cuda', generator=gen))
            (vv * G.conj()).real.sum().backward()
            res.append((vv.detach(), xx.grad.detach()))
        assert float((res[0][0] - res[1][0]).abs().max()) < 1e-5 * float(res[1][0].abs().max())
        assert float((res[0][1] - res[1][1]).abs().max()) < 1e-5 * float(res[1][1].abs().max())
        assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])


@pytest.mark.parametrize('world', [4, 8])
def test_tile_shards_stitch_to_the_unsharded_result_at_c4_size(ops, world):
    """dist.plan_tile_shards at the headline array (128 antennas, 8128 baselines, 98 304 pixels): every rank's shard
    evaluated with the plan's block grouping on this GPU, stitched with the plan's inverse permutation, equals the
    unsharded matrix-core result (visibilities; and the summed per-rank psky gradients the all-reduce would form)"""
    from bayeslim_amd import dist as rdist
    rng = np.random.default_rng(21)
    Nant, Nf, P = 128, 8, 98304
    ant = rng.normal(0, 120.0, (Nant, 3)); ant[:, 2] = 0.0
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    antp = T64(ant).cuda()
    blv = antp[torch.as_tensor([b for _, b in pairs], device='cuda')] - antp[torch.as_tensor([a for a, _ in pairs], device='cuda')]
    cz, az = rng.uniform(0, 1, P), rng.uniform(0, 2 * np.pi, P)
    sz = np.sqrt(1 - cz ** 2)
    sdir = T64(np.stack([sz * np.sin(az), sz * np.cos(az), cz])[None]).cuda()
    freqs = torch.linspace(150e6, 152e6, Nf, dtype=torch.float64)
    gen = torch.Generator(device='cuda').manual_seed(8)
    x0 = torch.randn(1, 1, 1, Nf, P, device='cuda', generator=gen)
    full = ops.FringeGeometry(blv, sdir, freqs, antpos=antp, bl_ants=pairs, mfma=True)
    x = x0.clone().requires_grad_(True)
    v_full = ops.fringe_sum(x, full)
    G = torch.complex(torch.randn(v_full.shape, device='cuda', generator=gen), torch.randn(v_full.shape, device='cuda', generator=gen))
    (v_full * G.conj()).real.sum().backward()
    plan = rdist.plan_tile_shards(pairs, Nant, world)
    assert sorted(i for r in plan['rank_bls'] for i in r) == list(range(len(pairs)))
    inv = torch.as_tensor(plan['inverse'], device='cuda')
    parts, gsum = [], torch.zeros_like(x0)
    for r in range(world):
        sel = torch.as_tensor(plan['rank_bls'][r], device='cuda')
        geom = ops.FringeGeometry(blv[sel], sdir, freqs, antpos=antp, bl_ants=[pairs[i] for i in plan['rank_bls'][r]],
                                  mfma=True, group=plan['group'])
        assert geom.ant is not None and len(geom.ant['blocks']) == plan['nblocks'][r]
        xr = x0.clone().requires_grad_(True)
        vr = ops.fringe_sum(xr, geom)
        (vr * G[:, sel].conj()).real.sum().backward()
        parts.append(vr.detach())
        gsum += xr.grad
    stitched = torch.cat(parts, dim=1).index_select(1, inv)
    assert float((stitched - v_full.detach()).abs().max()) < 3e-6 * float(v_full.detach().abs().max())
    assert float((gsum - x.grad).abs().max()) < 1e-5 * float(x.grad.abs().max())


def test_alm2pix_c3_size_against_float64(ops):
    """alm2pix at the C3 shape (128 rows, lmax 128 -> 8385 coefficients, 49 152 pixels): the f16-split matrix-core
    kernels (forward, LDS-DMA backward) against the float64 kernels on every entry, and run-to-run identity"""
    gen = torch.Generator(device='cuda').manual_seed(3)
    R, Nc, Npix = 128, 8385, 49152
    a = torch.complex(torch.randn(R, Nc, device='cuda', generator=gen), torch.randn(R, Nc, device='cuda', generator=gen))
    Y = torch.complex(torch.randn(Nc, Npix, device='cuda', generator=gen), torch.randn(Nc, Npix, device='cuda', generator=gen))
    g = torch.randn(R, Npix, device='cuda', generator=gen)
    outs = []
    for dt in (torch.complex64, torch.complex64, torch.complex128):
        x = a.detach().to(dt).clone().requires_grad_(True)
        y = ops.alm2pix(x, Y.to(dt))
        (y * g.to(y.dtype)).sum().backward()
        outs.append((y.detach(), x.grad.detach()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][0].double() - outs[2][0]).abs().max()) < 1e-5 * float(outs[2][0].abs().max())
    assert float((outs[0][1].to(torch.complex128) - outs[2][1]).abs().max()) < 1e-5 * float(outs[2][1].abs().max())


def test_fringe_sum_matrix_core_splits_and_degenerate_rows(ops):
    """MFMA path with several pixel splits (partial slabs + transposing reduction), an all-zero
    psky row (power-of-two scale of an empty row), an all-negative row (sign masks on every
    pixel) and a time/channel count that does not fill the 32-wide reduction tiles"""
    ant, pairs, blvecs, freqs, zenaz, psky = make_antenna_case(40, 40, Nt=3, Nf=5, P=20000, frac=1.0, autos=0)
    psky[0, :, :, 1] = 0.0
    psky[2, :, :, 3] = -psky[2, :, :, 3].abs()
    Nt, _, P = zenaz.shape
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    geom = ops.FringeGeometry(blvecs.cuda(), sdir.cuda(), freqs, antpos=ant.cuda(), bl_ants=pairs, mfma=True)
    assert geom.ant is not None
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, [0] * len(pairs))
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    x = pad_psky(psky, Ps).float().cuda().requires_grad_(True)
    vis = ops.fringe_sum(x, geom)
    assert torch.isfinite(torch.view_as_real(vis)).all()
    assert relmax(vis, ref) < 1e-5
    assert float(vis.detach()[0, :, 0, 1].abs().max()) == 0.0               # the empty row gives exact zeros
    (vis * gv.to(torch.complex64).cuda().conj()).real.sum().backward()
    assert relmax(x.grad[..., :P], ref_in.grad) < 1e-4


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_fringe_sum_split_paths(ops, dtype):
    """few baselines x many pixels -> pixel-split partial slabs + reduce (fwd);
    few pixels x many baselines -> baseline-split (bwd); also exercises the 2048-term flush"""
    for (Nbl, P) in [(3, 9000), (700, 10)]:
        blvecs, freqs, zenaz, psky, bl_mp = make_case(7, Nbl=Nbl, Nt=2, Nf=33, P=P, Nmp=1, Npp=1,
                                                      cplx=False)
        geom, Ps = to_gpu_geometry(ops, blvecs, freqs, zenaz, bl_mp, 1)
        ref_in = psky.clone().requires_grad_(True)
        ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, bl_mp)
        gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape)) + 0j)
        (ref * gv.conj()).real.sum().backward()
        rdt = torch.float64 if dtype == 'f64' else torch.float32
        x = pad_psky(psky, Ps).to(rdt).cuda().requires_grad_(True)
        vis = ops.fringe_sum(x, geom)
        assert relmax(vis, ref) < (1e-11 if dtype == 'f64' else 1e-5)
        (vis * gv.cuda().to(vis.dtype).conj()).real.sum().backward()
        assert relmax(x.grad[..., :P], ref_in.grad) < (1e-11 if dtype == 'f64' else 1e-4)


def test_fringe_sum_many_time_steps(ops):
    """more (time, split) blocks than a grid y/z dimension holds (65535): the launch geometry keeps
    them on grid.x; every time step repeats the same inputs, so all must equal the oracle's single one"""
    blvecs, freqs, zenaz, psky, bl_mp = make_case(3, Nbl=3, Nt=1, Nf=2, P=40, Nmp=1, Npp=1, cplx=False)
    ref = oracle_fringe_sum(psky, blvecs, zenaz, freqs, bl_mp)                   # (1, Nbl, 1, Nf)
    Nt = 66000
    geom, Ps = to_gpu_geometry(ops, blvecs, freqs, zenaz.expand(Nt, -1, -1).contiguous(), bl_mp, 1)
    x = pad_psky(psky, Ps).expand(Nt, -1, -1, -1, -1).contiguous().cuda()
    vis = ops.fringe_sum(x, geom)
    assert vis.shape == (1, 3, Nt, 2)
    assert relmax(vis[:, :, :1], ref) < 1e-11
    assert float((vis - vis[:, :, :1]).abs().max()) == 0.0
    g = ops.gen_fringe(blvecs.cuda(), geom.sdir[0], freqs, dtype=torch.float64)
    assert relmax(g[..., :40], orc.gen_fringe(blvecs, zenaz[0, 0], zenaz[0, 1], freqs)) < 1e-11


def test_fringe_sum_golden_prod_and_sum(ops):
    g = load_golden('prod_and_sum')
    zenaz = T64(np.stack([g['zen'], g['az']]))[None]
    psky = (T64(g['beam']) * T64(g['sky'])[:, :, None])[0, 0][None, :, None][:, :, 0][None]  # (1,1,1,Nf,P)
    psky = (T64(g['beam'])[0, 0] * T64(g['sky'])[0, 0][None]).reshape(1, 1, 1, *g['sky'].shape[2:])
    geom, Ps = to_gpu_geometry(ops, T64(g['blvecs']), T64(g['freqs']), zenaz, None, 1)
    vis = ops.fringe_sum(pad_psky(psky, Ps).cuda(), geom)          # (1, Nbl, 1, Nf)
    ref = g['sum_sky'][0, 0]                                       # (Nbl, Nf)
    assert relmax(vis[0, :, 0], ref) < 1e-11
    vis32 = ops.fringe_sum(pad_psky(psky, Ps).float().cuda(), geom)
    assert relmax(vis32[0, :, 0], ref) < 1e-5


def test_fringe_matches_golden_gen_fringe(ops):
    """one-hot psky recovers individual fringe values exp(2 pi i nu b.s/c) of gen_fringe"""
    for tag in ('uniform', 'ragged'):
        g = load_golden('fringe_' + tag)
        P = len(g['zen'])
        zenaz = T64(np.stack([g['zen'], g['az']]))[None]
        Nf = len(g['freqs'])
        for conj, key in [(False, 'fringe'), (True, 'fringe_conj')]:
            geom, Ps = to_gpu_geometry(ops, T64(g['blvecs']), T64(g['freqs']), zenaz, None, 1,
                                       conj=conj)
            for p in (0, 17, P - 1):
                psky = torch.zeros(1, 1, 1, Nf, Ps, dtype=torch.float64)
                psky[..., p] = 1.0
                vis = ops.fringe_sum(psky.cuda(), geom)[0, :, 0]
                assert relmax(vis, g[key][:, :, p]) < 1e-11
                vis32 = ops.fringe_sum(psky.float().cuda(), geom)[0, :, 0]
                assert np.abs(vis32.cpu().numpy() - g[key][:, :, p]).max() < 2e-6


@pytest.mark.parametrize('mode', ['nearest', 'linear', 'quadratic', 'cubic', 'linear,quadratic'])
@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_interp_gather(ops, mode, dtype):
    g = load_golden('interp_rect')
    key = mode.replace(',', '_')
    Npb = len(g['theta_grid']) * len(g['phi_grid'])
    # the fixture does not store the beam map: it is re-drawn from the generator's stream, which drew the 120
    # (zen, az) samples first (tests/golden/make_golden.py, gen_interp_cases) -- the two draws below replay them
    rng = np.random.default_rng(3)
    rng.uniform(0.02, 89.9, 120)
    rng.uniform(0.0, 359.999, 120)
    m = T64(rng.normal(size=(2, 3, Npb)))
    inds, wgts = torch.as_tensor(g[key + '__inds']), T64(g[key + '__wgts'])
    rdt = torch.float64 if dtype == 'f64' else torch.float32
    st = ops.InterpStencil(inds.cuda(), wgts.to(rdt).cuda(), Npb)
    x = m.to(rdt).cuda().requires_grad_(True)
    P = inds.shape[0]
    y = ops.interp_gather(x, st, out_stride=ops.pad_to_tile(P))
    assert y.shape == (2, 3, ops.pad_to_tile(P))
    assert (y[..., P:] == 0).all()
    tol = 1e-12 if dtype == 'f64' else 2e-6
    assert relmax(y[..., :P], g[key + '__out']) < tol
    gout = torch.zeros_like(y)
    gout[..., :P] = T64(g[key + '__gout']).to(rdt).cuda()
    (y * gout).sum().backward()
    ref = np.zeros(2 * 3 * Npb)
    ref[g[key + '__gm_nnz_idx']] = g[key + '__gm_nnz_val']
    assert relmax(x.grad.reshape(-1), ref) < (1e-12 if dtype == 'f64' else 2e-6)
    # determinism of the adjoint
    x.grad = None
    y2 = ops.interp_gather(x, st, out_stride=ops.pad_to_tile(P))
    (y2 * gout).sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    y3 = ops.interp_gather(x, st, out_stride=ops.pad_to_tile(P))
    (y3 * gout).sum().backward()
    assert torch.equal(g1, x.grad)


def test_interp_gather_complex_and_ragged_stencil(ops):
    rng = np.random.default_rng(0)
    Npb, P, Nnn = 500, 130, 6
    inds = torch.as_tensor(rng.integers(0, Npb, (P, Nnn)))
    wgts = T64(rng.normal(size=(P, Nnn)))
    m = torch.as_tensor(rng.normal(size=(5, Npb)) + 1j * rng.normal(size=(5, Npb)))
    mr = m.clone().requires_grad_(True)
    ref = orc.interp(mr, inds, wgts)
    gv = torch.as_tensor(rng.normal(size=(5, P)) + 1j * rng.normal(size=(5, P)))
    (ref * gv.conj()).real.sum().backward()
    st = ops.InterpStencil(inds.cuda(), wgts.cuda(), Npb)
    x = m.cuda().requires_grad_(True)
    y = ops.interp_gather(x, st)
    assert relmax(y, ref) < 1e-12
    (y * gv.cuda().conj()).real.sum().backward()
    assert relmax(x.grad, mr.grad) < 1e-12


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
@pytest.mark.parametrize('Nnn', [4, 9, 6])
def test_beam_sky_product(ops, dtype, Nnn):
    """fused interpolate - cut - multiply (ops.beam_sky_product) against oracle interp x cut sky,
    forward and both gradients; ragged time steps with zero padding, sky pixels seen 0, 1 or 2 times"""
    rng = np.random.default_rng(Nnn)
    R, Npb, Npix, Nt, Ps = (37 if Nnn == 6 else 72), 300, 500, 3, 192     # 72: the 4-channel vector loads, partial tile
    npix_t = [150, 192, 101]
    cuts = [np.sort(rng.choice(Npix, n, replace=False)) for n in npix_t]
    cut = np.full((Nt, Ps), Npix, dtype=np.int64)
    pos = np.full((Nt, Npix), -1, dtype=np.int64)
    for t, c in enumerate(cuts):
        cut[t, :len(c)] = c
        pos[t, c] = np.arange(len(c))
    inds = torch.as_tensor(rng.integers(0, Npb, (Nt * Ps, Nnn)))
    wgts = T64(rng.normal(size=(Nt * Ps, Nnn)))
    bmap, sky = T64(rng.normal(size=(R, Npb))), T64(rng.normal(size=(R, Npix)))
    b_ref, s_ref = bmap.clone().requires_grad_(True), sky.clone().requires_grad_(True)
    sky_ext = torch.cat([s_ref, torch.zeros(R, 1, dtype=torch.float64)], dim=1)
    ref = orc.interp(b_ref, inds, wgts) * sky_ext[:, torch.as_tensor(cut.reshape(-1))]
    gv = T64(rng.normal(size=tuple(ref.shape)))
    (ref * gv).sum().backward()
    rdt = torch.float64 if dtype == 'f64' else torch.float32
    st = ops.InterpStencil(inds.cuda(), wgts.to(rdt).cuda(), Npb)
    b, s = bmap.to(rdt).cuda().requires_grad_(True), sky.to(rdt).cuda().requires_grad_(True)
    out = ops.beam_sky_product(b, s, st, torch.as_tensor(cut.reshape(-1), dtype=torch.int32).cuda(),
                               torch.as_tensor(pos, dtype=torch.int32).cuda(), Nt, Ps)
    tol = 1e-12 if dtype == 'f64' else 3e-6
    assert relmax(out, ref) < tol
    (out * gv.to(rdt).cuda()).sum().backward()
    assert relmax(b.grad, b_ref.grad) < tol and relmax(s.grad, s_ref.grad) < tol
    # padded points give exact zeros
    assert float(out.detach().reshape(R, Nt, Ps)[:, 0, 150:].abs().max()) == 0.0


@pytest.mark.parametrize('Nt,R', [(1, 64), (2, 128), (5, 192), (9, 68), (3, 32), (4, 8), (2, 100)])
def test_beam_sky_product_time_pipeline_edges(ops, Nt, R):
    """the sky gradient's software pipeline over the time steps (stencils of step t+1 staged through LDS while step t is
    contracted, positions two steps ahead) at its edges: a single step, two steps, an odd count, whole 64-channel tiles and
    partial ones (round 5: masked lanes on the pipelined path -- 68 = 64 + 4, 100, and fewer than one tile: 32, 8 channels), sky pixels never visible / visible at every step, a time step with NO visible pixel in whole tiles"""
    rng = np.random.default_rng(100 * Nt + R)
    Npb, Npix, Ps, Nnn = 257, 700, 256, 4
    cut = np.full((Nt, Ps), Npix, dtype=np.int64)
    pos = np.full((Nt, Npix), -1, dtype=np.int64)
    for t in range(Nt):
        if Nt > 2 and t == 1:
            c = np.arange(640, 700)                       # only the last tile of sky pixels sees this step
        else:
            c = np.sort(rng.choice(Npix - 64, int(rng.integers(100, Ps)), replace=False)) + (0 if t % 2 else 32)
        cut[t, :len(c)] = c
        pos[t, c] = np.arange(len(c))
    inds = torch.as_tensor(rng.integers(0, Npb, (Nt * Ps, Nnn)))
    wgts = T64(rng.normal(size=(Nt * Ps, Nnn)))
    bmap, sky = T64(rng.normal(size=(R, Npb))), T64(rng.normal(size=(R, Npix)))
    b_ref, s_ref = bmap.clone().requires_grad_(True), sky.clone().requires_grad_(True)
    sky_ext = torch.cat([s_ref, torch.zeros(R, 1, dtype=torch.float64)], dim=1)
    ref = orc.interp(b_ref, inds, wgts) * sky_ext[:, torch.as_tensor(cut.reshape(-1))]
    gv = T64(rng.normal(size=tuple(ref.shape)))
    (ref * gv).sum().backward()
    for rdt, tol in ((torch.float64, 1e-12), (torch.float32, 3e-6)):
        st = ops.InterpStencil(inds.cuda(), wgts.to(rdt).cuda(), Npb)
        b, s_ = bmap.to(rdt).cuda().requires_grad_(True), sky.to(rdt).cuda().requires_grad_(True)
        out = ops.beam_sky_product(b, s_, st, torch.as_tensor(cut.reshape(-1), dtype=torch.int32).cuda(),
                                   torch.as_tensor(pos, dtype=torch.int32).cuda(), Nt, Ps)
        assert relmax(out, ref) < tol
        (out * gv.to(rdt).cuda()).sum().backward()
        assert relmax(b.grad, b_ref.grad) < tol and relmax(s_.grad, s_ref.grad) < tol
        # twice the same bits
        b2, s2 = bmap.to(rdt).cuda().requires_grad_(True), sky.to(rdt).cuda().requires_grad_(True)
        out2 = ops.beam_sky_product(b2, s2, st, torch.as_tensor(cut.reshape(-1), dtype=torch.int32).cuda(),
                                    torch.as_tensor(pos, dtype=torch.int32).cuda(), Nt, Ps)
        (out2 * gv.to(rdt).cuda()).sum().backward()
        assert torch.equal(out, out2) and torch.equal(b.grad, b2.grad) and torch.equal(s_.grad, s2.grad)


def test_side_kernels_at_c4_size(ops):
    """the HBM-bound kernels either side of the fringe sum at BASELINE config 4's sizes, float32 against the same
    kernels in float64 (themselves pinned to the oracle at small sizes) and against torch compositions: the fused
    beam-interpolate x cut-sky product with both gradients (256 channels, 2 times x 98 304 points, 32 760-node beam
    grid, 196 608 sky pixels), the chi-square epilogue and the gain application on (1, 1, 8128, 8, 256) visibilities"""
    from bayeslim_amd import calibration
    gen = torch.Generator(device='cuda').manual_seed(7)
    R, Npb, Npix, Nt, Ps, Nnn = 256, 32760, 196608, 2, 98304, 4
    rng = np.random.default_rng(7)
    cut = np.full((Nt, Ps), Npix, dtype=np.int64)
    pos = np.full((Nt, Npix), -1, dtype=np.int64)
    for t in range(Nt):
        c = np.sort(rng.choice(Npix, Ps - 1000 * t, replace=False))
        cut[t, :len(c)] = c
        pos[t, c] = np.arange(len(c))
    inds = torch.randint(0, Npb, (Nt * Ps, Nnn), device='cuda', generator=gen)
    wgts = torch.rand(Nt * Ps, Nnn, device='cuda', generator=gen, dtype=torch.float64)
    bmap = torch.randn(R, Npb, device='cuda', generator=gen, dtype=torch.float64)
    sky = torch.randn(R, Npix, device='cuda', generator=gen, dtype=torch.float64)
    gv = torch.randn(R, Nt * Ps, device='cuda', generator=gen, dtype=torch.float64)
    cut_d = torch.as_tensor(cut.reshape(-1), dtype=torch.int32).cuda()
    pos_d = torch.as_tensor(pos, dtype=torch.int32).cuda()
    res = []
    for rdt in (torch.float32, torch.float64):
        st = ops.InterpStencil(inds, wgts.to(rdt), Npb)
        b, s_ = bmap.to(rdt).requires_grad_(True), sky.to(rdt).requires_grad_(True)
        out = ops.beam_sky_product(b, s_, st, cut_d, pos_d, Nt, Ps)
        (out * gv.to(rdt)).sum().backward()
        res.append((out.detach().double(), b.grad.double(), s_.grad.double()))
    for a, b in zip(*res):
        assert float((a - b).abs().max()) < 1e-5 * float(b.abs().max())
    del res, bmap, sky, gv, inds, wgts
    # chi-square and gain application on the headline visibility tensor
    shape = (1, 1, 8128, 8, 256)
    pred = torch.complex(torch.randn(shape, device='cuda', generator=gen), torch.randn(shape, device='cuda', generator=gen))
    data = torch.complex(torch.randn(shape, device='cuda', generator=gen), torch.randn(shape, device='cuda', generator=gen))
    icov = torch.rand(shape, device='cuda', generator=gen) + 0.5
    x = pred.clone().requires_grad_(True)
    c = ops.chisq(x, data, icov)
    c.backward()
    r64 = (pred - data).to(torch.complex128)
    ref = float((r64.abs() ** 2 * icov.double()).sum())
    assert abs(float(c.detach()) - ref) < 1e-5 * ref
    assert float((x.grad - 2 * (pred - data) * icov).abs().max()) < 1e-5 * float(x.grad.abs().max())
    Nant = 128
    pairs = [(i, j) for i in range(Nant) for j in range(i + 1, Nant)]
    g1 = torch.as_tensor([p[0] for p in pairs], device='cuda')
    g2 = torch.as_tensor([p[1] for p in pairs], device='cuda')
    gains = torch.complex(torch.randn(1, 1, Nant, 8, 256, device='cuda', generator=gen),
                          torch.randn(1, 1, Nant, 8, 256, device='cuda', generator=gen))
    cot = data
    v, gn = pred.clone().requires_grad_(True), gains.clone().requires_grad_(True)
    out, _ = calibration._apply_cal(v, gn, g1.int(), g2.int())
    (out * cot.conj()).real.sum().backward()
    v2, gn2 = pred.clone().requires_grad_(True), gains.clone().requires_grad_(True)
    ref = gn2.index_select(2, g1) * v2 * gn2.index_select(2, g2).conj()
    (ref * cot.conj()).real.sum().backward()
    assert float((out - ref).detach().abs().max()) < 1e-5 * float(ref.detach().abs().max())
    assert float((v.grad - v2.grad).abs().max()) < 1e-5 * float(v2.grad.abs().max())
    assert float((gn.grad - gn2.grad).abs().max()) < 1e-4 * float(gn2.grad.abs().max())


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_alm2pix(ops, dtype):
    g = load_golden('sph_harm')
    a = torch.as_tensor(g['alm'])
    Y = torch.as_tensor(g['Ylm_comp'])
    mult = T64(g['alm_mult_comp'])
    cdt = torch.complex128 if dtype == 'f64' else torch.complex64
    x = torch.view_as_real(a).clone().to(torch.float64 if dtype == 'f64' else torch.float32).cuda()
    x.requires_grad_(True)
    xa = torch.view_as_complex(x) * mult.to(x.dtype).cuda()
    y = ops.alm2pix(xa, Y.to(cdt).cuda())
    tol = 1e-12 if dtype == 'f64' else 3e-6
    assert relmax(y, g['fwd_full']) < tol
    (y * T64(g['gout_full']).to(y.dtype).cuda()).sum().backward()
    assert relmax(x.grad, g['galm_full']) < tol


def test_alm2pix_larger_random(ops):
    rng = np.random.default_rng(1)
    l, m = orc.gen_lm(20)
    th, ph = np.arccos(rng.uniform(-1, 1, 777)), rng.uniform(0, 2 * np.pi, 777)
    Y = torch.as_tensor(orc.sph_Ylm(th, ph, l, m))
    a = torch.as_tensor(rng.normal(size=(2, 1, 37, len(l))) + 1j * rng.normal(size=(2, 1, 37, len(l))))
    ar = a.clone().requires_grad_(True)
    ref = orc.forward_alm(ar, Y)
    gv = T64(rng.normal(size=tuple(ref.shape)))
    (ref * gv).sum().backward()
    for cdt, tol in [(torch.complex128, 1e-12), (torch.complex64, 1e-5)]:
        x = a.to(cdt).cuda().requires_grad_(True)
        y = ops.alm2pix(x, Y.to(cdt).cuda())
        assert relmax(y, ref) < tol
        (y * gv.to(y.dtype).cuda()).sum().backward()
        assert relmax(x.grad, ar.grad) < tol


@pytest.mark.parametrize('R,lmax,Npix', [(128, 24, 3000), (40, 12, 1111), (3, 40, 5000)])
def test_alm2pix_mfma_shapes(ops, R, lmax, Npix):
    """float32 path = f32 matrix cores: row-tile counts 4 / 2 / 1, ragged pixel and coefficient tails,
    pixel-split backward"""
    rng = np.random.default_rng(R)
    l, m = orc.gen_lm(lmax)
    th, ph = np.arccos(rng.uniform(-1, 1, Npix)), rng.uniform(0, 2 * np.pi, Npix)
    Y = torch.as_tensor(orc.sph_Ylm(th, ph, l, m))
    a = torch.as_tensor(rng.normal(size=(R, len(l))) + 1j * rng.normal(size=(R, len(l))))
    ar = a.clone().requires_grad_(True)
    ref = orc.forward_alm(ar, Y)
    gv = T64(rng.normal(size=tuple(ref.shape)))
    (ref * gv).sum().backward()
    x = a.to(torch.complex64).cuda().requires_grad_(True)
    y = ops.alm2pix(x, Y.to(torch.complex64).cuda())
    assert relmax(y, ref) < 1e-5
    (y * gv.float().cuda()).sum().backward()
    assert relmax(x.grad, ar.grad) < 1e-5
    g1 = x.grad.clone()
    x.grad = None
    y = ops.alm2pix(x, Y.to(torch.complex64).cuda())
    (y * gv.float().cuda()).sum().backward()
    assert torch.equal(g1, x.grad)                      # deterministic


@pytest.mark.parametrize('R,lmax,Npix', [(128, 24, 3000), (40, 12, 1111), (3, 40, 5000), (70, 31, 2048), (128, 15, 12289)])
def test_alm2pix_packed_ylm_equals_unpacked(ops, R, lmax, Npix, monkeypatch):
    """the cached pre-split copies of Ylm in fragment order (rime_alm2pix_pack / _fwd_packed / _bwd_packed): same
    split and the same products as the kernels that split Ylm on the fly -- forward bitwise equal (same summation order;
    these shapes run without a K split, the split forward is compared in test_alm2pix_packed_forward_with_a_k_split),
    backward to 2e-6 (another deal of the pixel chunks) -- and both against the float64 oracle; row tiles 4 / 2 / 1, ragged coefficient and pixel tails, one-block and many-block grids"""
    rng = np.random.default_rng(R + Npix)
    l, m = orc.gen_lm(lmax)
    th, ph = np.arccos(rng.uniform(-1, 1, Npix)), rng.uniform(0, 2 * np.pi, Npix)
    Y = torch.as_tensor(orc.sph_Ylm(th, ph, l, m))
    a = torch.as_tensor(rng.normal(size=(R, len(l))) + 1j * rng.normal(size=(R, len(l))))
    ar = a.clone().requires_grad_(True)
    ref = orc.forward_alm(ar, Y)
    gv = T64(rng.normal(size=tuple(ref.shape)))
    (ref * gv).sum().backward()
    Yd = Y.to(torch.complex64).cuda()
    res = {}
    for packed in (False, True):
        monkeypatch.setattr(ops, 'ALM_PACKED', packed)
        monkeypatch.setattr(ops, 'ALM_PACKED_MIN_BYTES', 0)
        x = a.to(torch.complex64).cuda().requires_grad_(True)
        y = ops.alm2pix(x, Yd)
        (y * gv.float().cuda()).sum().backward()
        assert (ops.ylm_packed_state(Yd) is not None) == packed
        if packed:
            assert set(ops.ylm_packed_state(Yd)[3]) == {0, 1} and all(b is not False for b in ops.ylm_packed_state(Yd)[3].values())
        res[packed] = (y.detach(), x.grad.detach())
        assert relmax(y, ref) < 1e-5 and relmax(x.grad, ar.grad) < 1e-5
    assert torch.equal(res[False][0], res[True][0])
    # backward: the packed kernels deal the pixel chunks to their blocks differently (256-coefficient blocks, one K step
    # per chunk; odd pixel counts: the unpacked path is another kernel altogether): same products, another order of the
    # float32 partial sums
    assert relmax(res[True][1], res[False][1]) < 2e-6


def test_alm2pix_packed_forward_with_a_k_split(ops, monkeypatch):
    """the packed forward splits K over blockIdx.z when the map offers too few blocks (2 splits at the C3 shape) and sums
    the partial planes in a second kernel: another order of the float32 sums than the unsplit kernel -- equal to 5e-6, not
    bitwise; forced here with a pixel count small enough for the planner to split"""
    rng = np.random.default_rng(77)
    l, m = orc.gen_lm(60)
    Npix, R = 4096, 64
    th, ph = np.arccos(rng.uniform(-1, 1, Npix)), rng.uniform(0, 2 * np.pi, Npix)
    Y = torch.as_tensor(orc.sph_Ylm(th, ph, l, m))
    a = torch.as_tensor(rng.normal(size=(R, len(l))) + 1j * rng.normal(size=(R, len(l))))
    ref = orc.forward_alm(a, Y)
    Yd = Y.to(torch.complex64).cuda()
    res = {}
    for packed in (False, True):
        monkeypatch.setattr(ops, 'ALM_PACKED', packed)
        monkeypatch.setattr(ops, 'ALM_PACKED_MIN_BYTES', 0)
        res[packed] = ops.alm2pix(a.to(torch.complex64).cuda(), Yd).detach()
        assert relmax(res[packed], ref) < 1e-5
    assert relmax(res[True], res[False].cpu().numpy()) < 5e-6        # measured 1.6e-6: not bitwise


def test_alm2pix_packed_ylm_is_repacked_when_ylm_changes(ops, monkeypatch):
    """the packed copies live on the Ylm tensor object with its version counter: an in-place update or another tensor
    (what AlmModel.setup_Ylm installs) must not be served from a stale copy"""
    monkeypatch.setattr(ops, 'ALM_PACKED', True)
    monkeypatch.setattr(ops, 'ALM_PACKED_MIN_BYTES', 0)
    rng = np.random.default_rng(5)
    l, m = orc.gen_lm(20)
    Npix = 4096
    th, ph = np.arccos(rng.uniform(-1, 1, Npix)), rng.uniform(0, 2 * np.pi, Npix)
    Y = torch.as_tensor(orc.sph_Ylm(th, ph, l, m)).to(torch.complex64).cuda()
    a = torch.as_tensor(rng.normal(size=(8, len(l))) + 1j * rng.normal(size=(8, len(l)))).to(torch.complex64).cuda()
    y1 = ops.alm2pix(a, Y)
    first = ops.ylm_packed_state(Y)[3][0]
    assert ops.alm2pix(a, Y) is not None and ops.ylm_packed_state(Y)[3][0] is first          # second call: the cached copy
    Y.mul_(2.0)                                                                   # in place: version counter moves
    y2 = ops.alm2pix(a, Y)
    assert ops.ylm_packed_state(Y)[3][0] is not first
    assert relmax(y2, 2.0 * y1) < 1e-6
    Y2 = (Y * 0.25).contiguous()                                                  # another tensor
    y3 = ops.alm2pix(a, Y2)
    assert relmax(y3, 0.5 * y1) < 1e-6
    # through the model: setup_Ylm replaces the matrix
    from bayeslim_amd import sph_harm
    A = sph_harm.AlmModel(l, m, real_output=True)
    A.device = torch.device('cuda', 0)
    A.setup_Ylm(np.rad2deg(th), np.rad2deg(ph), Ylm=Y2, alm_mult=None)
    o1 = A(a)
    A.setup_Ylm(np.rad2deg(th), np.rad2deg(ph), Ylm=(Y2 * 3.0).contiguous(), alm_mult=None)
    assert relmax(A(a), 3.0 * o1) < 1e-6
    # explicit release (two buffers of Ylm's size per matrix) and use from another stream: the pack kernel's event orders it
    ops.release_ylm_packed(Y2)
    assert ops.ylm_packed_state(Y2) is None
    side = torch.cuda.Stream()
    y4 = ops.alm2pix(a, Y2)                                                       # packs on the current stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y5 = ops.alm2pix(a, Y2)                                                   # cached copy, other stream: waits for the event
    side.synchronize()
    assert torch.equal(y4, y5) and relmax(y4, 0.5 * y1) < 1e-6
    # a backward pass after the caller dropped its Ylm object falls back to the kernels that split on the fly
    x = a.clone().requires_grad_(True)
    Y3 = (Y2 * 1.0).contiguous()
    out = ops.alm2pix(x, Y3)
    del Y3
    out.sum().backward()
    assert torch.isfinite(torch.view_as_real(x.grad)).all()


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
@pytest.mark.parametrize('beam_complex,Nmp,same', [(False, 1, True), (True, 1, True), (False, 3, False), (True, 2, False)])
def test_jones_apply_against_the_einsum(ops, dtype, beam_complex, Nmp, same):
    """fused J1 S J2^dagger (csrc/jones.hip) against the einsum of the reference's 4-pol apply_beam branch
    (beam_model.py:345-363, "ab...,bc...,dc...->ad...") in float64 on the CPU: values and the gradients with respect
    to both beams and the sky, real and complex (Jones) beams, one sky shared by several beam-model pairs"""
    rng = np.random.default_rng(17 + Nmp)
    Nf, P = 5, 333
    rdt, cdt = (torch.float64, torch.complex128) if dtype == 'f64' else (torch.float32, torch.complex64)
    tol = 1e-12 if dtype == 'f64' else 2e-6

    def beam():
        b = rng.normal(size=(2, 2, Nmp, Nf, P))
        return torch.as_tensor(b + 1j * rng.normal(size=b.shape)) if beam_complex else torch.as_tensor(b)

    b1 = beam()
    b2 = b1 if same else beam()
    sk = torch.as_tensor(rng.normal(size=(2, 2, 1, Nf, P)) + 1j * rng.normal(size=(2, 2, 1, Nf, P)))
    g = torch.as_tensor(rng.normal(size=(2, 2, Nmp, Nf, P)) + 1j * rng.normal(size=(2, 2, Nmp, Nf, P)))
    # reference (float64, CPU)
    r1 = b1.clone().requires_grad_(True)
    r2 = r1 if same else b2.clone().requires_grad_(True)
    rs = sk.clone().requires_grad_(True)
    ref = torch.einsum('ab...,bc...,dc...->ad...', r1.to(torch.complex128), rs.expand(2, 2, Nmp, Nf, P), r2.conj().to(torch.complex128))
    (ref * g.conj()).real.sum().backward()
    # device
    bdt = cdt if beam_complex else rdt
    d1 = b1.to(bdt).cuda().requires_grad_(True)
    d2 = d1 if same else b2.to(bdt).cuda().requires_grad_(True)
    ds = sk.to(cdt).cuda().requires_grad_(True)
    out = ops.jones_apply(d1, d2, ds)
    assert out.shape == (2, 2, Nmp, Nf, P) and out.dtype == cdt
    (out * g.to(cdt).cuda().conj()).real.sum().backward()
    assert relmax(out, ref) < tol
    assert relmax(d1.grad, r1.grad) < tol and relmax(ds.grad, rs.grad) < tol
    if not same:
        assert relmax(d2.grad, r2.grad) < tol
    assert d1.grad.dtype == bdt and ds.grad.shape == ds.shape


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('fshape', [(3, 1, 1, 1), (3, 1, 7, 1), (3, 1, 1, 501), (3, 1, 7, 501)])
def test_stokes2coherency_fused_against_the_torch_composition(ops, dtype, fshape):
    """sky_model.Stokes2Coherency on a Stokes-I sky with fractional (Q, U, V): the fused kernels (rime_stokes2coh_fwd / _bwd)
    against the module's own torch composition (the reference's arithmetic, sky_model.py:1160-1300) -- values and the gradient
    w.r.t. the Stokes-I map, fractions broadcast over channels and / or pixels; fractions that require a gradient keep torch"""
    from bayeslim_amd import sky_model
    gen = torch.Generator(device='cuda').manual_seed(3)
    I = torch.randn(1, 1, 7, 501, device='cuda', dtype=dtype, generator=gen)
    fr = 0.3 * torch.randn(*fshape, device='cuda', dtype=dtype, generator=gen)
    cdt = torch.complex64 if dtype == torch.float32 else torch.complex128
    G = torch.randn(2, 2, 7, 501, device='cuda', dtype=cdt, generator=gen)
    s2c = sky_model.Stokes2Coherency(params=fr)
    a = I.clone().requires_grad_(True)
    out = s2c(a)
    assert out.shape == (2, 2, 7, 501) and out.dtype == cdt
    (out * G.conj()).real.sum().backward()
    # the composition, spelled out
    b = I.clone().requires_grad_(True)
    i0 = b[0, 0]
    Q, U, V = i0 * fr[0, 0], i0 * fr[1, 0], i0 * fr[2, 0]
    ref = torch.stack([torch.stack([i0 + Q, U - 1j * V]), torch.stack([U + 1j * V, i0 - Q])])
    (ref * G.conj()).real.sum().backward()
    tol = 1e-6 if dtype == torch.float32 else 1e-14
    assert float((out - ref).abs().max()) <= tol * float(ref.abs().max())
    assert float((a.grad - b.grad).abs().max()) <= tol * float(b.grad.abs().max())
    assert ops.stokes2coherency(I[0, 0], fr) is not None
    assert ops.stokes2coherency(I[0, 0], fr.clone().requires_grad_(True)) is None          # torch path keeps the fraction gradient
    frg = fr.clone().requires_grad_(True)
    out2 = sky_model.Stokes2Coherency(params=frg)(I)
    (out2 * G.conj()).real.sum().backward()
    assert frg.grad is not None and float((out2 - ref).abs().max()) <= 10 * tol * float(ref.abs().max())


def test_complex_row_scale_kernel_against_torch(ops):
    """rime_fringe_row_scale_cplx (scale from max(|re|, |im|), per-plane minima of interleaved complex rows through strides)
    against the torch passes it replaces, on a permuted (time-inner) psky view with an all-zero and a tiny row"""
    from bayeslim_amd import _lib
    gen = torch.Generator(device='cuda').manual_seed(5)
    Nt, Nmp, Npp, Nf, Ps = 3, 2, 4, 5, 1000
    f32 = torch.float32                                                 # whatever default dtype an earlier test left behind
    base = torch.randn(Npp, Nmp, Nf, Nt, Ps, 2, device='cuda', generator=gen, dtype=f32) * torch.exp(
        6 * torch.randn(Npp, Nmp, Nf, Nt, 1, 1, device='cuda', generator=gen, dtype=f32))
    base[1, 0, 2, 1] = 0.0
    base[2, 1, 0, 0] *= 1e-30
    inp = base.permute(3, 1, 0, 2, 4, 5)                               # (Nt, Nmp, Npp, Nf, Ps, 2), not contiguous
    scale = torch.empty(Nmp, Npp, Nt, Nf, device='cuda', dtype=f32)
    lo = [torch.empty_like(scale), torch.empty_like(scale)]
    _lib.check(_lib.lib.rime_fringe_row_scale_cplx(inp.data_ptr(), Nmp, Npp, Nt, Nf, inp.stride(1) // 2, inp.stride(2) // 2,
                                                   inp.stride(0) // 2, inp.stride(3) // 2, Ps, scale.data_ptr(),
                                                   lo[0].data_ptr(), lo[1].data_ptr(), torch.cuda.current_stream().cuda_stream), 'row_scale_cplx')
    amax = inp.abs().amax(dim=(-1, -2)).permute(1, 2, 0, 3)
    assert torch.equal(scale, ops._pow2_scale(amax).contiguous())
    ref_lo = inp.amin(dim=-2)
    for c in range(2):
        assert torch.equal(lo[c], ref_lo[..., c].permute(1, 2, 0, 3).contiguous())


def test_ops_refuse_cpu_tensors(ops):
    with pytest.raises(RuntimeError):
        ops.alm2pix(torch.zeros(2, 3, dtype=torch.complex64), torch.zeros(3, 4, dtype=torch.complex64))


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_chisq_epilogue(ops, dtype):
    """fused chi-square (residual, diagonal inverse covariance, sum) and its backward against the golden
    vectors of the imported reference (optim.apply_icov, cov_axis=None) and, at a larger size with
    broadcast icov / no data, against the oracle; bitwise reproducible"""
    g = load_golden('chisq')
    cdt = torch.complex128 if dtype == 'f64' else torch.complex64
    rdt = torch.float64 if dtype == 'f64' else torch.float32
    tol = 1e-12 if dtype == 'f64' else 3e-6
    data, icov = torch.as_tensor(g['data']).to(cdt).cuda(), torch.as_tensor(g['icov']).to(rdt).cuda()
    for tag, ic in (('icov', icov), ('noicov', None)):
        pred = torch.as_tensor(g['pred']).to(cdt).cuda().requires_grad_(True)
        c = ops.chisq(pred, data, ic)
        assert abs(float(c.detach()) - float(g['sum_' + tag])) < tol * abs(float(g['sum_' + tag]))
        (3.0 * c).backward()
        assert relmax(pred.grad, 3.0 * g['gpred_' + tag]) < tol
    rng = np.random.default_rng(0)
    shape = (1, 1, 301, 7, 129)
    p64 = torch.as_tensor(rng.normal(size=shape) + 1j * rng.normal(size=shape))
    ic64 = T64(rng.uniform(0.5, 2.0, size=(1, 1, 301, 1, 129)))                 # broadcast over time
    pr = p64.clone().requires_grad_(True)
    ref = orc.chisq(pr, torch.zeros_like(pr), ic64)
    ref.backward()
    x = p64.to(cdt).cuda().requires_grad_(True)
    c = ops.chisq(x, None, ic64.to(rdt).cuda())
    assert abs(float(c.detach()) - float(ref.detach())) < tol * float(ref.detach())
    c.backward()
    assert relmax(x.grad, pr.grad) < tol
    assert float(ops.chisq(x.detach(), None, ic64.to(rdt).cuda())) == float(c.detach())    # deterministic reduction
    # the reference-named entry point: (chisq, res) of LogProb.forward_chisq
    from bayeslim_amd import optim
    c2, res = optim.forward_chisq(x.detach(), None, ic64.to(rdt).cuda())
    assert res is None and float(c2) == float(c.detach())
    c3, res3 = optim.forward_chisq(x.detach(), torch.zeros_like(x.detach()), ic64.to(rdt).cuda(), sum_chisq=False)
    assert res3.shape == x.shape and abs(float(c3.sum()) - float(c.detach())) < 1e-5 * float(c.detach())


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
def test_apply_cal(ops, dtype):
    """fused gain application G1 V G2^dagger (1-pol, broadcast gains, 2-pol diagonal, 4-pol) and both
    gradients against the golden vectors of the imported reference (calibration._apply_cal), then at a
    larger size with gains broadcast over time against the oracle, through the reference-named entry point"""
    from bayeslim_amd import calibration
    g = load_golden('apply_cal')
    cdt = torch.complex128 if dtype == 'f64' else torch.complex64
    tol = 1e-12 if dtype == 'f64' else 1e-5
    for tag, two in (('1pol', False), ('1pol_bcast', False), ('2pol', True), ('4pol', False)):
        vis = torch.as_tensor(g['vis_' + tag]).to(cdt).cuda().requires_grad_(True)
        gains = torch.as_tensor(g['gains_' + tag]).to(cdt).cuda().requires_grad_(True)
        out, cov = calibration._apply_cal(vis, gains, g['g1_idx'], g['g2_idx'], cal_2pol=two)
        assert cov is None and out.shape == vis.shape
        assert relmax(out.detach(), g['vout_' + tag]) < tol, tag
        (out * torch.as_tensor(g['cot_' + tag]).to(cdt).cuda().conj()).real.sum().backward()
        assert relmax(vis.grad, g['gvis_' + tag]) < tol, tag
        assert gains.grad.shape == gains.shape
        assert relmax(gains.grad, g['ggains_' + tag]) < tol, tag
    rng = np.random.default_rng(5)
    Nant, Nt, Nf = 23, 5, 67
    bls = [(i, j) for i in range(Nant) for j in range(i, Nant)]
    ants = list(range(100, 100 + Nant))
    blnames = [(ants[i], ants[j]) for i, j in bls]
    cplx = lambda *s: torch.as_tensor(rng.normal(size=s) + 1j * rng.normal(size=s))
    for Np, two, gshape in ((1, False, (Nant, 1, Nf)), (2, True, (Nant, Nt, 1)), (2, False, (Nant, 1, 1)), (2, False, (Nant, Nt, Nf))):
        v64, g64, cot = cplx(Np, Np, len(bls), Nt, Nf), cplx(Np, Np, *gshape), cplx(Np, Np, len(bls), Nt, Nf)
        vr, gr = v64.clone().requires_grad_(True), g64.clone().requires_grad_(True)
        ref = orc.apply_cal(vr, gr, [b[0] for b in bls], [b[1] for b in bls], cal_2pol=two)
        (ref * cot.conj()).real.sum().backward()
        v, gn = v64.to(cdt).cuda().requires_grad_(True), g64.to(cdt).cuda().requires_grad_(True)
        out, _ = calibration.apply_cal(v, blnames, gn, ants, cal_2pol=two)
        assert relmax(out.detach(), ref.detach()) < tol
        (out * cot.to(cdt).cuda().conj()).real.sum().backward()
        assert relmax(v.grad, vr.grad) < tol
        assert relmax(gn.grad, gr.grad) < 10 * tol
    # the remaining branches of the reference function: undo + covariance (1-pol, 2-pol), delay-type visibilities
    rdt = torch.float64 if dtype == 'f64' else torch.float32
    for tag, two in (('1pol', False), ('2pol', True)):
        vis = torch.as_tensor(g['u_vis_' + tag]).to(cdt).cuda().requires_grad_(True)
        gains = torch.as_tensor(g['u_gains_' + tag]).to(cdt).cuda().requires_grad_(True)
        cov = torch.as_tensor(g['u_cov_' + tag]).to(rdt).cuda()
        out, cout = calibration._apply_cal(vis, gains, g['g1_idx'], g['g2_idx'], cal_2pol=two, cov=cov, undo=True)
        assert relmax(out.detach(), g['u_vout_' + tag]) < 10 * tol, tag
        assert relmax(cout, g['u_cout_' + tag]) < 10 * tol, tag
        (out * torch.as_tensor(g['u_cot_' + tag]).to(cdt).cuda().conj()).real.sum().backward()
        assert relmax(vis.grad, g['u_gvis_' + tag]) < 10 * tol, tag
        assert relmax(gains.grad, g['u_ggains_' + tag]) < 100 * tol, tag
    dv, dgn = torch.as_tensor(g['dly_vis']).to(rdt).cuda(), torch.as_tensor(g['dly_gains']).to(rdt).cuda()
    for undo in (False, True):
        out, _ = calibration._apply_cal(dv, dgn, g['g1_idx'], g['g2_idx'], vis_type='dly', undo=undo)
        assert relmax(out, g['dly_vout_undo%d' % undo]) < tol
    # 4-pol undo: the proper 2 x 2 inverse (the reference's torch.pinv branch cannot run): undo(apply(V)) == V
    v4 = cplx(2, 2, len(bls), Nt, Nf).to(cdt).cuda()
    g4 = (cplx(2, 2, Nant, Nt, Nf) + 2 * torch.eye(2)[:, :, None, None, None]).to(cdt).cuda()
    fwd, _ = calibration.apply_cal(v4, blnames, g4, ants)
    back, _ = calibration.apply_cal(fwd, blnames, g4, ants, undo=True)
    assert relmax(back, v4) < (1e-10 if dtype == 'f64' else 1e-3)


def test_eq2top_kernel_matches_float64_host_chain(ops):
    """rime_eq2top (per-direction part of the ICRS -> (zen, az) chain, telescope_model.py:469-502) against the
    numpy float64 restatement on the same host-built frame; includes the pole, the zenith and az wrap"""
    from bayeslim_amd import astrometry as A, telescope_model
    loc = (21.42827, -30.72148, 1050.0)
    rng = np.random.default_rng(0)
    ra = np.concatenate([rng.uniform(0, 360, 20000), [0.0, 0.0, 123.4, 359.999999]])
    dec = np.concatenate([np.rad2deg(np.arcsin(rng.uniform(-1, 1, 20000))), [90.0, -90.0, -30.72148, 0.0]])
    for jd in (2459861.0, 2459861.37, 2451545.0, 2460676.5):
        M, vb, vd = A.observation_frame(loc, jd)
        za = ops.eq2top(torch.as_tensor(ra).cuda(), torch.as_tensor(dec).cuda(), M, vb, vd).cpu().numpy()
        zen, az = A.icrs_to_topo(loc, jd, ra, dec)
        assert np.abs(za[0] - zen).max() < 1e-10
        daz = np.abs(za[1] - az)
        daz = np.minimum(daz, 360.0 - daz) * np.sin(np.deg2rad(zen))        # az is ill-defined at the zenith
        assert daz.max() < 1e-10
        assert (za[1] >= 0).all() and (za[1] < 360).all()
    # through the model: sky angles on the GPU take the kernel, host tensors the numpy chain
    telescope_model._WARNED = True
    tel = telescope_model.TelescopeModel(loc)
    a = tel.eq2top(2459861.2, torch.as_tensor(ra).cuda(), torch.as_tensor(dec).cuda())
    b = tel.eq2top(2459861.2, torch.as_tensor(ra), torch.as_tensor(dec))
    assert a.is_cuda and np.abs(a.cpu().numpy()[0] - b.numpy()[0]).max() < 1e-10


def test_eq2top_kernel_against_the_independent_oracle(ops):
    """rime_eq2top + the host-built frame (the product's ICRS -> (zen, az) chain, replacing telescope_model.py:469-502)
    against oracle/eq2top_oracle.py: an independent float64 restatement with a different factorisation (CIO based,
    Fukushima-Williams precession, differentiated VSOP87 Earth, spherical triangle) that is itself pinned END TO END
    to SOFA's published atci13 / atio13 / atco13 answers at the 4-10 mas level (tests/test_oracle_golden.py).
    Tolerance 20 mas = 5.6e-6 deg on the sky (measured: <= 11 mas; the difference is the product's low-precision
    Earth velocity and equinox-based sidereal time against the oracle's) -- against a 17 arcsec nutation, 20 arcsec
    aberration and 0.3 deg of precession that the chain has to get right.  astropy itself: parity unpinned."""
    from bayeslim_amd import astrometry as A, telescope_model
    from oracle import eq2top_oracle as E
    tol_deg = 0.020 / 3600.0
    rng = np.random.default_rng(1)
    ra = np.concatenate([rng.uniform(0, 360, 20000), [0.0, 123.4, 359.999999]])
    dec = np.concatenate([np.rad2deg(np.arcsin(rng.uniform(-1, 1, 20000))), [89.9, -30.72148, 0.0]])
    worst = 0.0
    for loc in [(21.42827, -30.72148, 1050.0), (116.67, -26.70, 377.0), (-107.6, 34.08, 2124.0)]:
        for jd, dut1 in [(2459861.0, 0.0), (2459861.37, -0.02), (2451545.0, 0.3), (2456384.969254051, 0.1550675), (2462000.25, 0.0)]:
            M, vb, vd = A.observation_frame(loc, jd, dut1)
            za = ops.eq2top(torch.as_tensor(ra).cuda(), torch.as_tensor(dec).cuda(), M, vb, vd).cpu().numpy()
            zen, az = E.eq2top(loc, jd, ra, dec, dut1)
            dz = np.abs(za[0] - zen)
            daz = np.abs(za[1] - az)
            daz = np.minimum(daz, 360.0 - daz) * np.sin(np.deg2rad(zen))
            worst = max(worst, dz.max(), daz.max())
            assert dz.max() < tol_deg and daz.max() < tol_deg, (loc, jd, dz.max() * 3.6e6, daz.max() * 3.6e6)
    # SOFA's own end-to-end case through the DEVICE path: the star of `atco13` at its date (space motion applied by the
    # oracle's adaptor), refraction removed from the published observed values; without the case's polar motion (0.21 arcsec)
    # 0.3 arcsec, with it (observation_frame(..., xp, yp), what TelescopeModel(iers_file=...) supplies) 20 mas
    import json, math
    c = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'sofa_vectors.json')))
    ap, c = c['apco13'], c['atco13']
    jd = c['utc1'] + c['utc2']
    p = E.sofa_case_star_direction(c['rc'], c['dc'], c['pr'], c['pd'], c['px'], c['rv'], jd + (E.dat(jd) + 32.184) / 86400.0)
    ra1, dec1 = math.degrees(math.atan2(p[1, 0], p[0, 0])), math.degrees(math.asin(p[2, 0]))
    loc = (math.degrees(c['elong']), math.degrees(c['phi']), c['hm'])
    ztrue = math.degrees(E.sofa_case_remove_refraction(c['zob'], ap['refa'], ap['refb']))
    for eop, tol in (((), 0.3), ((c['xp'], c['yp']), 0.02)):       # round 4: with the case's polar motion 20 mas, not 0.3 arcsec
        M, vb, vd = A.observation_frame(loc, jd, c['dut1'], *eop)
        za = ops.eq2top(torch.as_tensor([ra1]).cuda(), torch.as_tensor([dec1]).cuda(), M, vb, vd).cpu().numpy()
        assert abs(za[0, 0] - ztrue) * 3600 < tol and abs(za[1, 0] - math.degrees(c['aob'])) * math.sin(c['zob']) * 3600 < tol, (eop, za)
    # through the model (conv_cache miss -> astrometry chain on the device)
    telescope_model._WARNED = True
    tel = telescope_model.TelescopeModel((21.42827, -30.72148, 1050.0))
    a = tel.eq2top(2459861.2, torch.as_tensor(ra).cuda(), torch.as_tensor(dec).cuda()).cpu().numpy()
    zen, az = E.eq2top((21.42827, -30.72148, 1050.0), 2459861.2, ra, dec, getattr(tel, 'dut1', 0.0) or 0.0)
    assert np.abs(a[0] - zen).max() < tol_deg


@pytest.mark.parametrize('Npp,cplx,dtype', [(3, False, torch.float64), (5, False, torch.float32), (2, True, torch.float64),
                                            (3, True, torch.float32), (7, False, torch.float64)])
def test_fringe_sum_any_number_of_planes(ops, Npp, cplx, dtype):
    """plane counts the baseline-formulation kernels do not take natively (imaging puts Nmaps on this axis,
    imaging.py:717-815) run as several launches -- same values and gradients as the oracle"""
    blvecs, freqs, zenaz, psky, bl_mp = make_case(21, Nbl=9, Nt=2, Nf=5, P=130, Nmp=1, Npp=Npp, cplx=cplx)
    geom, Ps = to_gpu_geometry(ops, blvecs, freqs, zenaz, bl_mp, 1)
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, bl_mp)
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    cdt = {torch.float64: torch.complex128, torch.float32: torch.complex64}[dtype]
    x = pad_psky(psky, Ps).to(cdt if cplx else dtype).cuda().requires_grad_(True)
    vis = ops.fringe_sum(x, geom)
    tv, tg = (1e-11, 1e-11) if dtype == torch.float64 else (1e-5, 1e-4)
    assert vis.shape == (Npp, 9, 2, 5) and relmax(vis, ref) < tv
    (vis * gv.to(cdt).cuda().conj()).real.sum().backward()
    assert relmax(x.grad[..., :130], ref_in.grad) < tg
    if not cplx:
        adj = ops.fringe_adjoint(gv.to(cdt).cuda(), geom)
        assert relmax(adj[..., :130], ref_in.grad) < tg


@pytest.mark.parametrize('Nbl,Nmp,P', [(339, 4, 1), (730, 4, 64), (500, 3, 200), (97, 5, 30)])
def test_fringe_sum_baseline_formulation_model_pair_groups_workspace(ops, Nbl, Nmp, P):
    """every beam-model-pair group of the vector-ALU backward plans its own baseline splits; the workspace bound
    must cover the group with the most splits (a random cross-check found RIME_EWORKSPACE here)"""
    blvecs, freqs, zenaz, psky, bl_mp = make_case(Nbl, Nbl=Nbl, Nt=1, Nf=28, P=P, Nmp=Nmp, Npp=4, cplx=False)
    geom, Ps = to_gpu_geometry(ops, blvecs, freqs, zenaz, bl_mp, Nmp)
    ref_in = psky.clone().requires_grad_(True)
    ref = oracle_fringe_sum(ref_in, blvecs, zenaz, freqs, bl_mp)
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    x = pad_psky(psky, Ps).float().cuda().requires_grad_(True)
    vis = ops.fringe_sum(x, geom)
    assert relmax(vis, ref) < 1e-5
    (vis * gv.to(torch.complex64).cuda().conj()).real.sum().backward()
    assert relmax(x.grad[..., :P], ref_in.grad) < 1e-4


@pytest.mark.parametrize('mfma,dtype,cplx', [(False, torch.float64, False), (False, torch.float64, True), (True, torch.float32, False)])
def test_fringe_sum_gradient_wrt_baseline_vectors(ops, mfma, dtype, cplx):
    """d loss / d blvecs (the reference's gen_fringe is differentiable w.r.t. blvecs through autograd,
    telescope_model.py:350-356): three direction-cosine-weighted forward passes, against the fp64 oracle --
    through antenna positions on the matrix-core path"""
    ant, pairs, blvecs, freqs, zenaz, _ = make_antenna_case(9, 40, Nt=2, Nf=5, P=300, frac=1.0, autos=0)
    rng = np.random.default_rng(12)
    Nt, _, P = zenaz.shape
    psky = rng.normal(size=(Nt, 1, 2, 5, P))
    if cplx:
        psky = psky + 1j * rng.normal(size=psky.shape)
    psky = torch.as_tensor(psky)
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(Nt, 3, Ps, dtype=torch.float64)
    for t in range(Nt):
        sdir[t, :, :P] = orc.pointing_vectors(zenaz[t, 0], zenaz[t, 1])
    # oracle: gradient w.r.t. antenna positions through blvecs = ant[j] - ant[i]
    a64 = ant.clone().requires_grad_(True)
    i1, i2 = torch.as_tensor([a for a, _ in pairs]), torch.as_tensor([b for _, b in pairs])
    ref = oracle_fringe_sum(psky, a64[i2] - a64[i1], zenaz, freqs, [0] * len(pairs))
    gv = torch.as_tensor(np.random.default_rng(5).normal(size=tuple(ref.shape))
                         + 1j * np.random.default_rng(6).normal(size=tuple(ref.shape)))
    (ref * gv.conj()).real.sum().backward()
    cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
    ag = ant.cuda().requires_grad_(True)
    blv = ag[i2.cuda()] - ag[i1.cuda()]
    geom = ops.FringeGeometry(blv.detach(), sdir.cuda(), freqs, antpos=ant.cuda() if mfma else None,
                              bl_ants=pairs if mfma else None, mfma=True if mfma else 'auto')
    assert (geom.ant is not None) == mfma
    x = pad_psky(psky, Ps).to(cdt if cplx else dtype).cuda()
    vis = ops.fringe_sum(x, geom, blv)
    (vis * gv.to(cdt).cuda().conj()).real.sum().backward()
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    assert relmax(ag.grad, a64.grad) < tol


def test_rccl_wrappers_one_rank(ops):
    """rime_comm_* (thin RCCL wrappers of the C ABI): a one-rank communicator on the test GPU -- the all-gather
    returns the block, the all-reduce leaves the gradients, both stream-ordered with the surrounding torch work"""
    from bayeslim_amd import dist as rdist
    comm = rdist.RcclComm(1, 0, rdist.RcclComm.unique_id())
    try:
        v = torch.randn(3, 1000, 7, dtype=torch.complex64, device='cuda')
        out = comm.allgather_vis(v * 2)
        assert out.shape == (1, 3, 1000, 7) and torch.equal(out[0], v * 2)
        g32, g64 = torch.randn(100000, device='cuda'), torch.randn(33, 5, dtype=torch.complex128, device='cuda')
        a, b = g32.clone(), g64.clone()
        comm.reduce_grads([g32, g64])
        torch.cuda.synchronize()
        assert torch.equal(g32, a) and torch.equal(g64, b)
    finally:
        comm.close()
