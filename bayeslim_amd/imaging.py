"""
Map-making arithmetic of the reference's imaging.py on the fused fringe
kernels: the imaging matrix A = conj(fringe) * beam of VisMapper.build_A (imaging.py:251-296) is never
materialised -- (Nbl, Nf, Npix) complex, C4: 1.6 TB per time step.  `make_map` is the adjoint of the
RIME fringe sum (the backward kernels, including the antenna-factored matrix-core path), `compute_Am`
its forward, `compute_Pm` their composition.

`VisMapper` (imaging.py:12-714) is the reference's container around them -- time / baseline / channel selection,
the three diagonal normalisations, the PSF contractions -- with the per-time-step A replaced by a fused operator
(FringeGeometry + beam + FoV cut); `build_A` still materialises A for callers that want the matrix of a small
problem.  The module-level functions (imaging.py:717-736, 755-774, 777-815) take an ops.FringeGeometry instead of A.
"""
import numpy as np
import torch

from . import ops, telescope_model, utils
from .dataset import VisData, MapData


def geometry(blvecs, zen, az, freqs, antpos=None, bl_ants=None):
    """FringeGeometry of ONE time step at pointing angles (zen, az) [deg] (all pixels kept: apply the
    FoV cut before, as build_A does); antpos / bl_ants enable the antenna-factored kernels"""
    zen = torch.as_tensor(zen, dtype=torch.float64, device=blvecs.device)
    az = torch.as_tensor(az, dtype=torch.float64, device=blvecs.device)
    P = zen.numel()
    Ps = ops.pad_to_tile(P)
    sdir = torch.zeros(1, 3, Ps, dtype=torch.float64, device=blvecs.device)
    sdir[0, :, :P] = telescope_model.pointing_vectors(zen, az)
    geom = ops.FringeGeometry(blvecs, sdir, freqs, npix=[P], antpos=antpos, bl_ants=bl_ants)
    geom.P = P
    return geom


def _pad(x, Ps):
    return x if x.shape[-1] == Ps else torch.nn.functional.pad(x, (0, Ps - x.shape[-1]))


def make_map(v, w, geom, beam=None):
    """
    dirty map  m[..., f, p] = beam[f, p] * Re sum_b conj(F[b, f, p]) (v w)[..., b, f]   (imaging.py:717-736
    with A = conj(fringe) * beam).  v (..., Nbl, Nf) complex, w (Nbl, Nf) real -> (..., Nf, P) real.
    """
    lead = v.shape[:-2]
    g = (v * w).reshape((-1,) + tuple(v.shape[-2:]))               # (Nmaps, Nbl, Nf)
    out = ops.fringe_adjoint(g[:, :, None, :], geom)[0, 0]         # (Nmaps, Nf, Ps)
    out = out[..., :geom.P]
    if beam is not None:
        out = out * beam
    return out.reshape(lead + tuple(out.shape[-2:]))


def compute_Am(geom, m, beam=None):
    """
    conj(A) @ m = sum_p F[b, f, p] beam[f, p] m[..., f, p]: the RIME forward of a map (imaging.py:755-774).
    m (..., Nf, P) real -> (..., Nbl, Nf) complex.
    """
    lead = m.shape[:-2]
    x = m if beam is None else m * beam
    x = _pad(x.reshape((-1,) + tuple(m.shape[-2:])), geom.Pstride)  # (Nmaps, Nf, Ps)
    vis = ops.fringe_sum(x[None, None].contiguous(), geom)          # psky (1, 1, Nmaps, Nf, Ps) -> (Nmaps, Nbl, 1, Nf)
    return vis[:, :, 0].reshape(lead + (geom.Nbl, geom.Nf))


def compute_Pm(geom, w, m, beam=None, D=None):
    """P m = D A^T w (conj(A) m) (imaging.py:777-815): (..., Nf, P) real"""
    Pm = make_map(compute_Am(geom, m, beam), w, geom, beam)
    return Pm if D is None else Pm * D


def compute_P(geom, w, beam=None, D=None, contract=None):
    """
    PSF matrix P = D A^T w conj(A) of one time step (imaging.py:818-861) without A:
      'diag'   : sum_b w |A|^2 = beam^2 sum_b w            (|fringe| = 1: closed form)
      'rowsum' : P @ 1 = compute_Pm of a map of ones
      None     : P[f, p, q] = (P @ e_q)[f, p]: the unit maps through compute_Pm -- (Nf, P, P), small problems only
    """
    Nf, P = geom.Nf, geom.P
    rdt = w.dtype
    if contract == 'diag':
        out = (w.sum(0) * torch.ones(Nf, dtype=rdt, device=w.device))[:, None].expand(Nf, P)
        out = out * beam ** 2 if beam is not None else out.clone()
    elif contract == 'rowsum':
        out = compute_Pm(geom, w, torch.ones(Nf, P, dtype=rdt, device=w.device), beam)
    elif contract is None:
        eye = torch.eye(P, dtype=rdt, device=w.device)[:, None, :].expand(P, Nf, P)
        out = compute_Pm(geom, w, eye, beam).permute(1, 2, 0).contiguous()          # [q, f, p] -> [f, p, q]
    else:
        raise ValueError("contract must be None, 'diag' or 'rowsum'")
    if D is not None:
        out = out * (D[:, :, None] if contract is None else D)
    return out


def get_visdata(vd, bl_inds=None, time_inds=None, freq_inds=None, squeeze=False, **kwargs):
    """tensor (..., Npol, Npol, Nbls, Ntimes, Nfreqs) from a tensor, a VisData or a list of either (imaging.py:933-972)"""
    if isinstance(vd, torch.Tensor):
        sl = lambda x: slice(None) if x is None else x
        data = vd[..., sl(bl_inds), sl(time_inds), sl(freq_inds)]
        return data.squeeze() if squeeze else data
    if isinstance(vd, VisData):
        return vd.get_data(bl_inds=bl_inds, time_inds=time_inds, freq_inds=freq_inds, squeeze=squeeze, **kwargs)
    return torch.stack([get_visdata(v, bl_inds=bl_inds, time_inds=time_inds, freq_inds=freq_inds, squeeze=squeeze,
                                    **kwargs) for v in vd])


class VisMapper:
    """
    Images interferometric visibilities held in a VisData: y = A x, dirty map m = D A^T W y, PSF P = D A^T W conj(A)
    (imaging.py:12-714; single-pol imaging, antenna-independent beam).  Same constructor, selection setters,
    normalisation methods and outputs as the reference; the per-time-step products run on the fused fringe kernels.
    """
    def __init__(self, vd, ra, dec, beam=None, fov=180, dtype=None, cache_A=False, **kwargs):
        self.vd = vd.copy(copydata=False, copymeta=True)
        self.telescope = self.vd.telescope
        self.array = telescope_model.ArrayModel(self.vd.antpos, self.vd.freqs, device=self.vd.data.device,
                                                skip_reds=True, **kwargs)
        self.ra, self.dec, self.Npix = ra, dec, len(ra)
        self.device = self.vd.data.device
        self.dtype = dtype
        self.beam = beam
        self.fov = beam.fov if beam is not None else fov
        self._freqs = self.vd.freqs
        self.set_freq_inds()
        self._times = np.asarray(self.vd.times.cpu())            # numpy: the telescope cache keys on the float
        self.set_time_inds()
        self._blnums = self.vd.blnums
        self.set_bl_inds()
        self.cache_A = cache_A
        self.clear_cache()
        self.set_normalization()

    def clear_cache(self):
        self.A = {}               # materialised (A, cut) per time index: filled by build_A callers only
        self._ops = {}            # fused operators (geom, geom2, beam, cut) per time index
        self.D = None

    # ---- selections (imaging.py:103-231)
    def set_freq_inds(self, freq_inds=None, freqs=None):
        assert not ((freqs is not None) and (freq_inds is not None))
        fidx = lambda f: torch.where(torch.isclose(self._freqs, torch.as_tensor(f, dtype=self._freqs.dtype,
                                                                                device=self._freqs.device), atol=1e-10))[0]
        if freqs is not None:
            many = isinstance(freqs, (list, np.ndarray)) or (isinstance(freqs, torch.Tensor) and freqs.ndim == 1)
            freq_inds = torch.stack([fidx(f) for f in freqs]).ravel().tolist() if many else fidx(freqs).tolist()
        if freq_inds is None:
            freq_inds = slice(None)
        self.freq_inds = utils._list2slice(freq_inds)
        self.freqs = self._freqs[self.freq_inds]
        self.Nfreqs = len(self.freqs)
        self.clear_cache()

    def set_time_inds(self, time_inds=None, times=None):
        """self.time_inds index self.vd.times; the `time_ind` of build_v / build_w and of the caches index
        self.times = self.vd.times[self.time_inds]"""
        assert not ((times is not None) and (time_inds is not None))
        tidx = lambda t: np.where(np.isclose(self._times, t, atol=1e-10, rtol=1e-13))[0]
        if times is not None:
            many = isinstance(times, list) or (isinstance(times, (torch.Tensor, np.ndarray)) and times.ndim == 1)
            time_inds = np.concatenate([tidx(float(t)) for t in times]).tolist() if many else tidx(float(times)).tolist()
            if many:
                assert len(time_inds) == len(times)
        if time_inds is None:
            time_inds = list(range(len(self._times)))
        elif isinstance(time_inds, slice):
            time_inds = utils._slice2tensor(slice(time_inds.start, time_inds.stop if time_inds.stop is not None
                                                   else len(self._times), time_inds.step)).tolist()
        elif isinstance(time_inds, (np.ndarray, torch.Tensor)):
            time_inds = [time_inds.tolist()] if time_inds.ndim == 0 else time_inds.tolist()
        elif isinstance(time_inds, (int, np.integer)):
            time_inds = [int(time_inds)]
        self.time_inds = time_inds
        self.times = self._times[time_inds]
        self.Ntimes = len(self.times)
        self.clear_cache()

    def set_bl_inds(self, bl_inds=None, blnums=None):
        assert not ((blnums is not None) and (bl_inds is not None))
        blidx = lambda bl: np.where(self._blnums == bl)[0]
        if blnums is not None:
            many = isinstance(blnums, list) or (isinstance(blnums, (torch.Tensor, np.ndarray)) and blnums.ndim == 1)
            bl_inds = np.concatenate([blidx(int(b)) for b in blnums]).tolist() if many else blidx(int(blnums)).tolist()
        if bl_inds is None:
            bl_inds = slice(None)
        self.bl_inds = utils._list2slice(bl_inds)
        self.blnums = self._blnums[self.bl_inds]
        self.Nbls = len(self.blnums)
        self.bls = utils.blnum2ants(self.blnums)
        self.blvecs = self.array.get_blvecs(self.bls)
        self.clear_cache()

    def set_normalization(self, method='A2w', icov=None, clip=1e-8):
        """D = 1 / (1 @ w) ('w'), 1 / (w @ |A|) ('Aw') or 1 / (w @ |A|^2) ('A2w', least squares)"""
        assert method in ['w', 'Aw', 'A2w']
        self.method, self.icov, self.D, self.clip = method, icov, None, clip

    # ---- per-time-step pieces
    def _angles(self, time):
        zen, az = self.telescope.eq2top(time, self.ra, self.dec, store=True)
        return torch.as_tensor(zen, device=self.device), torch.as_tensor(az, device=self.device)

    def _beam_cut(self, time):
        """(beam (Nf, P) or None, cut, zen, az) as build_A derives them (imaging.py:268-283)"""
        zen, az = self._angles(time)
        if self.beam is not None:
            beam, cut, zen, az = self.beam.gen_beam(zen, az)
            beam = beam[:, :, :, self.freq_inds].to(self.device)[0, 0, 0]
            if not self.beam.powerbeam:
                beam = beam ** 2
            return beam.detach(), cut, zen, az
        cut = torch.where(zen <= self.fov / 2)[0]
        return None, cut, zen[cut], az[cut]

    @torch.no_grad()
    def build_A(self, time):
        """the imaging matrix conj(fringe) * beam of ONE time, materialised: (Nbls, Nfreqs, P) complex, and the
        pixel cut (imaging.py:251-296).  The mapper's own products never build it."""
        beam, cut, zen, az = self._beam_cut(time)
        self.array.set_freq_index(self.freq_inds)
        A = self.array.gen_fringe(self.blvecs, zen, az, conj=True)
        self.array.set_freq_index(None)
        if beam is not None:
            A = A * beam
        return A, cut

    @torch.no_grad()
    def _op(self, i):
        """fused operator of time index i: geometry of the selected baselines / channels over the cut pixels (and the
        same with doubled baselines, see make_map), beam (Nf, P) or None, cut"""
        if i in self._ops:
            return self._ops[i]
        beam, cut, zen, az = self._beam_cut(self.times[i])
        ants = self.array.ants
        idx = {a: k for k, a in enumerate(ants)}
        bl_ants = [(idx[a], idx[b]) for a, b in self.bls]
        antvecs = self.array.antvecs.to(self.device, torch.float64)
        freqs = torch.as_tensor(self.freqs, dtype=torch.float64)
        geom = geometry(self.blvecs.to(self.device, torch.float64), zen, az, freqs, antpos=antvecs, bl_ants=bl_ants)
        geom2 = None
        if self.method == 'A2w':
            geom2 = geometry(2.0 * self.blvecs.to(self.device, torch.float64), zen, az, freqs, antpos=2.0 * antvecs,
                             bl_ants=bl_ants)
        op = (geom, geom2, beam, cut)
        if self.cache_A:
            self._ops[i] = op
        return op

    @torch.no_grad()
    def build_v(self, time_ind, vd=None):
        """visibilities (..., Nbls, Nfreqs) of self.times[time_ind] (imaging.py:298-325)"""
        vd = self.vd if vd is None else vd
        t = self.time_inds[time_ind]
        return get_visdata(vd, bl_inds=self.bl_inds, time_inds=slice(t, t + 1), freq_inds=self.freq_inds,
                           squeeze=False, try_view=True)[..., 0, 0, :, 0, :]

    def build_w(self, time_ind):
        """weights (Nbls, Nfreqs) from self.icov, else self.vd.icov, else ones (Nbls, 1) (imaging.py:327-358)"""
        icov = self.icov if self.icov is not None else self.vd.icov
        t = self.time_inds[time_ind]
        if icov is not None:
            return self.vd.get_icov(bl_inds=self.bl_inds, time_inds=t, icov=icov, freq_inds=self.freq_inds,
                                    squeeze=False)[0, 0, :, 0]
        return torch.ones(self.Nbls, 1, device=self.device)

    def _real(self):
        return torch.float64 if self.vd.data.dtype == torch.complex128 else torch.float32

    def _norm_term(self, w, op, quirk):
        """this time step's addend to the normalisation sum Aw, over the cut pixels (or (Nf, 1) for 'w').
        quirk: VisMapper.make_map sums w Re(A^2) for 'A2w' (imaging.py:446-447) where compute_Pm / compute_P sum
        w |A|^2 (:619, :694); Re(A^2) = beam^2 cos(2 phase) is the dirty map of unit visibilities on DOUBLED
        baselines -- one more adjoint pass, no A."""
        geom, geom2, beam, _ = op
        Nf = self.Nfreqs
        wsum = (w.sum(0) * torch.ones(Nf, dtype=w.dtype, device=w.device))[:, None]      # (Nf, 1)
        if self.method == 'w':
            return wsum
        if self.method == 'Aw':
            return wsum.expand(Nf, geom.P) * (beam.abs() if beam is not None else 1.0)
        if quirk:
            ones = torch.ones(self.Nbls, Nf, dtype=torch.complex128 if w.dtype == torch.float64 else torch.complex64,
                              device=w.device)
            t = make_map(ones, w.expand(self.Nbls, Nf), geom2)
            return t * beam ** 2 if beam is not None else t
        return wsum.expand(Nf, geom.P) * (beam ** 2 if beam is not None else 1.0)

    def _init_Aw(self, rdt):
        return torch.zeros(self.Nfreqs, 1 if self.method == 'w' else self.Npix, dtype=rdt, device=self.device)

    def _add_Aw(self, Aw, term, cut):
        if self.method == 'w':
            Aw += term
        else:
            Aw[..., cut] += term

    @torch.no_grad()
    def make_map(self, vd=None, return_P=True, contract='diag'):
        """
        dirty maps of every selected time, summed and normalised: (maps (..., Nfreqs, Npix), P) with P the PSF diagonal /
        row sum (Nfreqs, Npix), the full matrix (contract None) or None (imaging.py:360-466).
        """
        assert self.method is not None, "First run set_normalization()"
        vd = self.vd if vd is None else vd
        rdt = self._real()
        Nmaps = len(vd) if (isinstance(vd, list) or (isinstance(vd, torch.Tensor) and vd.ndim > 5)) else 1
        maps = torch.zeros(Nmaps, self.Nfreqs, self.Npix, dtype=rdt, device=self.device)
        if isinstance(vd, VisData):
            maps = maps[0]
        Aw = self._init_Aw(rdt)
        P = None
        if return_P:
            P = torch.zeros((self.Nfreqs, self.Npix) + (() if contract is not None else (self.Npix,)), dtype=rdt,
                            device=self.device)
        for i in range(self.Ntimes):
            op = self._op(i)
            geom, _, beam, cut = op
            v = self.build_v(i, vd=vd)
            w = self.build_w(i).to(rdt)
            maps[..., cut] += make_map(v, w, geom, beam)
            if return_P:
                _P = compute_P(geom, w, beam, contract=contract)
                if contract is not None:
                    P[:, cut] += _P
                else:
                    P[:, cut[:, None], cut[None, :]] += _P
            self._add_Aw(Aw, self._norm_term(w, op, quirk=True), cut)
        self.D = 1 / Aw.clip(self.clip)
        maps *= self.D
        if return_P:
            P *= self.D if contract is not None else self.D[:, :, None]
        return maps, P

    @staticmethod
    def _map_tensor(maps):
        m2t = lambda m: m.data if isinstance(m, MapData) else m
        if isinstance(maps, list):
            return torch.stack([m2t(m) for m in maps])
        return m2t(maps)

    @torch.no_grad()
    def compute_Am(self, maps):
        """A-bar @ maps: the RIME forward of the maps over the selected times, (Nmaps, Nbls, Ntimes, Nfreqs), the
        leading axis dropped for one map (imaging.py:468-525)"""
        maps = self._map_tensor(maps)
        many = maps.ndim > 2
        m = maps if many else maps[None]
        cdt = torch.complex128 if m.dtype == torch.float64 else torch.complex64
        v = torch.zeros(len(m), self.Nbls, self.Ntimes, self.Nfreqs, dtype=cdt, device=self.device)
        for i in range(self.Ntimes):
            geom, _, beam, cut = self._op(i)
            v[..., i, :] = compute_Am(geom, m[..., cut], beam)
        return v if many else v[0]

    @torch.no_grad()
    def compute_Pm(self, maps, D=None):
        """P @ maps summed over the selected times and normalised by D (self's method when None) (imaging.py:527-606)"""
        maps = self._map_tensor(maps)
        rdt = maps.dtype
        Pm = torch.zeros(tuple(maps.shape[:-2]) + (self.Nfreqs, self.Npix), dtype=rdt, device=self.device)
        Aw = self._init_Aw(rdt) if D is None else None
        for i in range(self.Ntimes):
            op = self._op(i)
            geom, _, beam, cut = op
            w = self.build_w(i).to(rdt)
            Pm[..., cut] += compute_Pm(geom, w, maps[..., cut], beam)
            if D is None:
                self._add_Aw(Aw, self._norm_term(w, op, quirk=False), cut)
        if D is None:
            D = 1 / Aw.clip(self.clip)
        return Pm * D

    @torch.no_grad()
    def compute_P(self, D=None, contract='diag'):
        """PSF matrix over all pixels, summed over the selected times: (Nfreqs, Npix[, Npix]) (imaging.py:608-687)"""
        rdt = self._real()
        P = torch.zeros((self.Nfreqs, self.Npix) + (() if contract is not None else (self.Npix,)), dtype=rdt,
                        device=self.device)
        Aw = self._init_Aw(rdt) if D is None else None
        for i in range(self.Ntimes):
            op = self._op(i)
            geom, _, beam, cut = op
            w = self.build_w(i).to(rdt)
            _P = compute_P(geom, w, beam, contract=contract)
            if contract is not None:
                P[:, cut] += _P
            else:
                P[:, cut[:, None], cut[None, :]] += _P
            if D is None:
                self._add_Aw(Aw, self._norm_term(w, op, quirk=False), cut)
        if D is None:
            D = 1 / Aw.clip(self.clip)
        return P * (D if contract is not None else D[:, :, None])

    def push(self, device):
        """move the mapper and what hangs off it to a device / dtype (imaging.py:689-714)"""
        for k, v in self.A.items():
            self.A[k] = (utils.push(v[0], device), utils.push(v[1], device))
        self._ops = {}
        if self.D is not None:
            self.D = utils.push(self.D, device)
        if not isinstance(device, torch.dtype):
            self.device = device
        if self.beam is not None:
            self.beam.push(device)
        self.array.push(device)
        self.vd.push(device)
        self.telescope.push(device)
        self.blvecs = utils.push(self.blvecs, device)
