"""
The OBJECT contract of the drop-in modules on the device (VERDICT r04 "missing" 2): the reference's users
pickle whole models (io.py:50-66 write_pkl / read_pkl), deep-copy them (one copy per device in optim.py:1517-1523, notebook
cell 54) and push them between dtypes and devices (rime_model.py:117-126 and the push() of every model).  Here a RIME that
has ALREADY run -- geometry cache with ctypes tables and device buffers, interpolation stencils, packed Ylm copies
attached -- goes through each of those and must compute the same bits afterwards; a forward under no_grad must leave
nothing behind but its output.
"""
import copy
import pickle

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_rime_gpu import T, DEV, _c2_setup, make_array, fill_cache, pixbeam_from_golden, relmax, ba  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture
def f32():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float32)
    yield
    torch.set_default_dtype(old)


from bayeslim_amd import utils as _utils  # noqa: E402


class CohSky(_utils.Module):
    """Stokes-I pixel sky -> coherency: the sky model of the 4-pol fixtures, at module level so that pickle finds it"""
    def __init__(self, stokes, s2c):
        super().__init__(name='cohsky')
        self.sky, self.s2c, self.device = stokes, s2c, stokes.device

    def forward(self, prior_cache=None, **kw):
        return self.s2c(self.sky(prior_cache=prior_cache))


def _pol40(ba):
    from bayeslim_amd import sky_model, rime_model
    g = load_golden('rime_pol40_mini')
    freqs = T(g['freqs'])
    arr, tel = make_array(ba, g, freqs)
    Nf, Npix = g['stokes_I'].shape[2:]
    Rs = sky_model.PixelSkyResponse(freqs, device=DEV)
    stokes = sky_model.PixelSky(T(g['stokes_I']), T(np.stack([g['ra'], g['dec']]), torch.float64),
                                float(g['px_area']), R=Rs, parameter=True, name='polsky')
    frac = T(g['frac_pol']).reshape(-1, 1, 1, 1) * torch.ones(len(g['frac_pol']), 1, Nf, Npix, device=DEV)
    sky = CohSky(stokes, sky_model.Stokes2Coherency(params=frac))
    ants = g['ants'].tolist()
    a2b = {a: int(g['ant2beam'][i]) for i, a in enumerate(ants)}
    beam = pixbeam_from_golden(ba, g, freqs, powerbeam=False, ant2beam=a2b)
    rime = rime_model.RIME(sky, tel, beam, arr, [tuple(b) for b in g['sim_bls'].tolist()], g['times'], freqs)
    fill_cache(tel, 'polsky', g)
    return g, rime, (lambda r: [r.sky.sky.params, r.beam.params])


def _c2(ba):
    g = load_golden('rime_c2_mini')
    rime, sky, beam = _c2_setup(ba, g)
    return g, rime, (lambda r: [r.sky.params, r.beam.params])


def _step(rime, params_of, g):
    """one forward + backward with the fixture's cotangent; returns (vis, gradients) detached"""
    ps = params_of(rime)
    for p in ps:
        p.grad = None
    vis = rime().data
    (vis * T(g['gvis']).conj()).real.sum().backward()
    return vis.detach().clone(), [p.grad.detach().clone() for p in ps]


@pytest.mark.parametrize('which', ['c2', 'pol40'])
def test_rime_pickle_and_deepcopy_after_a_forward_reproduce_the_same_bits(ba, f32, which):
    """pickle.loads(pickle.dumps(rime)) and copy.deepcopy(rime) of a model whose caches are populated (geometry with pair
    tables and ctypes offsets, interpolation / FoV-cut stencils, beam cache) each give a model with EMPTY derived caches
    that reproduces visibilities and gradients bit for bit on a fresh geometry; the original is untouched"""
    g, rime, params_of = (_c2 if which == 'c2' else _pol40)(ba)
    vis0, gr0 = _step(rime, params_of, g)
    assert len(rime._geom_cache) == 1 and len(rime._ant_like) == 1          # caches are attached when the copies are taken
    tv = 1e-5
    assert relmax(vis0, g['vis']) < tv
    clones = dict(pickle=pickle.loads(pickle.dumps(rime, protocol=4)), deepcopy=copy.deepcopy(rime))
    for tag, cl in clones.items():
        assert cl is not rime and cl.sky is not rime.sky and cl.beam is not rime.beam
        for k in rime._DERIVED:
            assert getattr(cl, k, {}) == {}, (tag, k)
        ps, qs = params_of(rime), params_of(cl)
        assert all(p.data_ptr() != q.data_ptr() and torch.equal(p, q) and q.requires_grad for p, q in zip(ps, qs))
        vis, gr = _step(cl, params_of, g)
        assert torch.equal(vis, vis0), tag
        for a, b in zip(gr, gr0):
            assert torch.equal(a, b), tag
        assert len(cl._geom_cache) == 1
        # the copy is independent: changing ITS parameters does not move the original's result
        with torch.no_grad():
            qs[0].mul_(2.0)
        assert not torch.equal(cl().data, vis0)
    vis1, gr1 = _step(rime, params_of, g)
    assert torch.equal(vis1, vis0) and all(torch.equal(a, b) for a, b in zip(gr1, gr0))
    assert len(rime._geom_cache) == 1


def test_rime_push_dtype_and_device_round_trips(ba, f32):
    """push(torch.float64) on every model of a float32 RIME that has run, then forward: equal (1e-12) to a float64 model BUILT
    from the same float32-rounded inputs -- nothing is left behind in float32, no stale cache is served -- and within the
    rounding of those inputs of the reference's float64 output; push back to float32: the first result to float32 accuracy;
    push('cpu') -> push('cuda'): bit for bit (rime_model.py:117-126 and the models' own push methods; derived caches are
    rebuilt, never converted)"""
    g, rime, params_of = _c2(ba)
    vis32, gr32 = _step(rime, params_of, g)
    assert vis32.dtype == torch.complex64

    def push_all(what):
        # (the antenna positions of the float32 model are float64, as everywhere in the tests and the bench: a dtype push
        #  leaves the array alone -- ArrayModel.push(float32) would ROUND them, as the reference's does)
        for m in (rime.sky, rime.beam, rime) + (() if isinstance(what, torch.dtype) else (rime.array,)):
            m.push(what)

    push_all(torch.float64)
    assert rime._geom_cache == {} and rime.sky.params.dtype == torch.float64 and rime.beam.params.dtype == torch.float64
    # the yardstick: a float64 model constructed from the float32-rounded channels and parameters.  Both run under a float64
    # default dtype (as in the reference, tensors made from Python numbers inside a forward take the default dtype)
    torch.set_default_dtype(torch.float64)
    try:
        vis64, gr64 = _step(rime, params_of, g)
        assert vis64.dtype == torch.complex128 and relmax(vis64, g['vis']) < 2e-6      # float32-ROUNDED inputs, float64 arithmetic
        gw = dict(g)
        for k in ('freqs', 'sky_params', 'beam_params', 'px_area'):      # what the float32 model holds in float32
            gw[k] = np.asarray(g[k]).astype(np.float32).astype(np.float64)
        ref, sky_w, beam_w = _c2_setup(ba, gw)
        vw, gw_grads = _step(ref, params_of, g)
    finally:
        torch.set_default_dtype(torch.float32)
    assert relmax(vis64, vw.cpu().numpy()) < 1e-12
    for a, b in zip(gr64, gw_grads):
        assert relmax(a, b.cpu().numpy()) < 1e-11
    # back to float32: the parameters and channels return to their float32 values exactly, but a dtype push also ROUNDS what
    # the float32 model kept in float64 (interpolation grids, sky angles -- as the reference's push does): float32 agreement
    push_all(torch.float32)
    vis32b, gr32b = _step(rime, params_of, g)
    assert vis32b.dtype == torch.complex64 and relmax(vis32b, vis32.cpu().numpy()) < 1e-5
    for a, b in zip(gr32b, gr32):
        assert relmax(a, b.cpu().numpy()) < 1e-4
    # a DEVICE round trip changes no value: bit for bit
    push_all('cpu')
    assert rime.sky.params.device.type == 'cpu'
    with pytest.raises(RuntimeError):
        rime()                                                   # no CPU path: loud, not a silent fallback
    push_all(DEV)
    vis, gr = _step(rime, params_of, g)
    assert torch.equal(vis, vis32b) and all(torch.equal(a, b) for a, b in zip(gr, gr32b))


@pytest.mark.parametrize('which', ['c2', 'pol40'])
def test_rime_no_grad_forward_keeps_nothing_but_its_output(ba, f32, which):
    """a forward under torch.no_grad() on warm caches leaves torch.cuda.memory_allocated() at its value before the call plus
    the output tensor: no fringe workspace, partial slab, psky or beam map stays referenced (the beam cache is cleared at
    the START of every forward -- the reference's protocol -- so it is warm on both sides of the measurement)"""
    g, rime, params_of = (_c2 if which == 'c2' else _pol40)(ba)
    _step(rime, params_of, g)
    with torch.no_grad():
        rime()
        rime()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    with torch.no_grad():
        vd = rime()
    torch.cuda.synchronize()
    out_bytes = vd.data.numel() * vd.data.element_size()
    grown = torch.cuda.memory_allocated() - before
    assert 0 <= grown - out_bytes <= 2048, (grown, out_bytes)      # allocator granularity: 512-byte blocks
    del vd
    assert torch.cuda.memory_allocated() == before


def test_almmodel_packed_ylm_is_rebuilt_for_a_copy(ba, f32, monkeypatch):
    """AlmModel with the cached pre-split Ylm copies (forward and backward) attached: a deep copy / an unpickled copy starts
    WITHOUT packed buffers (they are derived data, registered per tensor object, never pickled), packs again on first use
    and returns the same bits; the original keeps its buffers"""
    from bayeslim_amd import sph_harm, ops
    monkeypatch.setattr(ops, 'ALM_PACKED', True)
    monkeypatch.setattr(ops, 'ALM_PACKED_MIN_BYTES', 0)
    rng = np.random.default_rng(11)
    lmax, Npix, R = 30, 6000, 16
    l, m = sph_harm.gen_lm(lmax)
    l, m = np.asarray(l), np.asarray(m)
    colat = np.rad2deg(np.arccos(rng.uniform(-1, 1, Npix)))
    lon = rng.uniform(0, 360, Npix)
    A = sph_harm.AlmModel(l, m, real_output=True)
    A.device = DEV
    A.setup_Ylm(colat, lon, generate=True)
    assert A.Ylm.is_cuda and A.Ylm.dtype == torch.complex64

    def run(model):
        a = T(rng0.normal(size=(R, len(l))) + 1j * rng0.normal(size=(R, len(l)))).requires_grad_(True)
        out = model(a)
        (out * w).sum().backward()
        return out.detach(), a.grad.detach()

    w = T(rng.normal(size=(R, Npix)))
    rng0 = np.random.default_rng(3)
    out0, g0 = run(A)
    st = ops.ylm_packed_state(A.Ylm)
    assert st is not None and set(st[3]) == {0, 1} and all(v is not False for v in st[3].values())
    bufs = {d: st[3][d][0].data_ptr() for d in (0, 1)}
    for tag, B in (('deepcopy', copy.deepcopy(A)), ('pickle', pickle.loads(pickle.dumps(A, protocol=4)))):
        assert B.Ylm is not A.Ylm and B.Ylm.data_ptr() != A.Ylm.data_ptr() and torch.equal(B.Ylm, A.Ylm)
        assert ops.ylm_packed_state(B.Ylm) is None, tag
        rng0 = np.random.default_rng(3)
        out, gr = run(B)
        assert torch.equal(out, out0) and torch.equal(gr, g0), tag
        sb = ops.ylm_packed_state(B.Ylm)
        assert sb is not None and set(sb[3]) == {0, 1} and sb[3][0][0].data_ptr() != bufs[0]
    st2 = ops.ylm_packed_state(A.Ylm)
    assert st2 is st and {d: st2[3][d][0].data_ptr() for d in (0, 1)} == bufs
    # the registry lets go of a matrix when its tensor object dies
    n = len(ops._YLM_PACKED)
    del B, sb
    import gc
    gc.collect()
    assert len(ops._YLM_PACKED) < n
