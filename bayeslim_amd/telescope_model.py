"""
Telescope / array models with the reference's API (telescope_model.py), for the RIME path:
`TelescopeModel.eq2top` (cached per key, :89-131) and `ArrayModel` (antenna positions,
baseline vectors, redundancy bookkeeping, `gen_fringe`, :142-460).

Differences that matter:
  * eq2top: the reference calls astropy's ICRS->AltAz on a cache miss (:498-502).  When astropy is
    importable that is what runs here too (`_astropy_eq2top`); it is not available in the build image, where
    a miss is served by the built-in chain of `bayeslim_amd/astrometry.py`
    (IAU 2006 precession + frame bias, truncated IAU 1980 nutation, GAST, annual + diurnal
    aberration; float64; per-direction work in the HIP kernel `rime_eq2top` for sky angles on the
    GPU) and announced ONCE with a warning that names what is not modelled (polar motion, UT1-UTC
    unless `TelescopeModel.dut1` is set, light deflection).  Pinned to SOFA known answers, PARITY
    UNPINNED against astropy itself.  Pre-populating `conv_cache` with (zen, az) from any source
    gives the reference's behaviour exactly (the cache-hit path is identical).
  * the module-level `eq2top` / `eq2top_device` / `JD2LST` are the PLAIN hour-angle rotation
    (SURVEY.md App. B.1): generators of synthetic (zen, az) inputs for benchmarks and tests, where
    the same arrays feed the CPU baseline and the GPU path.  The models never call them.
  * gen_fringe: the (Nbl, Nf, P) fringe tensor is produced by a HIP kernel, in float64 phase
    arithmetic; RIME itself never calls it (the fringe is fused into the sum).
"""
import copy
import itertools
import math

import numpy as np
import torch

import warnings

from . import utils, ops, astrometry
from .utils import _float, _cfloat, D2R


_WARNED = False


class TelescopeModel:
    def __init__(self, location, tloc=None, device=None, dtype=None, iers_file=None):
        """location = (lon, lat[, alt]) in degrees, as the reference (:22-53).
        iers_file (not in the reference, whose astropy reads its own IERS tables): an IERS Earth-orientation table
        (finals2000A / EOP C04 / `MJD xp yp UT1-UTC` text, astrometry.EarthOrientation) supplying UT1-UTC and the polar
        motion to the astrometry chain that serves a conv_cache miss; without one they are 0 (or `self.dut1` [s])."""
        self.location = location
        self.tloc = tloc
        self.dtype = dtype
        self.conv_cache = {}
        self.device = device
        self.eop = astrometry.EarthOrientation.from_file(iers_file) if iers_file is not None else None

    def earth_orientation(self, time):
        """(UT1-UTC [s], xp, yp [rad]) at a UTC Julian date: from the IERS table when one was given, else (self.dut1 | 0, 0, 0)"""
        if getattr(self, 'eop', None) is not None:
            return self.eop.at(float(time))
        return float(getattr(self, 'dut1', 0.0) or 0.0), 0.0, 0.0

    def hash(self, time, ra):
        return (time, len(ra))

    def clear_cache(self, key=None):
        if key is None:
            self.conv_cache = {}
        else:
            del self.conv_cache[key]

    def eq2top(self, time, ra, dec, store=False, key=None):
        """(zen, az) [deg] stacked as a (2, N) tensor; cached under `key` (:89-131).  A cache miss runs
        the astrometry chain (see the module header) instead of astropy and says so once."""
        key = key if key is not None else self.hash(time, ra)
        if key in self.conv_cache:
            return self.conv_cache[key]
        if getattr(self, 'use_astropy', True):
            angs = _astropy_eq2top(self.location, float(time), ra, dec)
            if angs is not None:                             # astropy present: exactly what the reference computes
                angs = torch.as_tensor(np.stack(angs), device=self.device, dtype=self.dtype)
                if store:
                    self.conv_cache[key] = angs
                return angs
        global _WARNED
        if not _WARNED:
            _WARNED = True
            warnings.warn("TelescopeModel.eq2top: (zen, az) not in conv_cache; computed by bayeslim_amd.astrometry "
                          "(IAU 2006 precession, truncated IAU 1980 nutation, annual + diurnal aberration) instead of "
                          "astropy's ICRS->AltAz. UT1-UTC and polar motion are 0 unless TelescopeModel(iers_file=...) supplies an IERS "
                          "table (or .dut1 [s] is set); light deflection is not modelled; expected agreement ~10 mas with "
                          "the table, a few 0.1 arcsec (+ 15 arcsec per second of UT1-UTC) without.", stacklevel=2)
        dut1, xp, yp = self.earth_orientation(time)
        M, vb, vd = astrometry.observation_frame(self.location, float(time), dut1, xp, yp)
        if isinstance(ra, torch.Tensor) and ra.is_cuda:
            angs = ops.eq2top(ra, torch.as_tensor(dec, device=ra.device), M, vb, vd)
            if self.device is not None:
                angs = angs.to(self.device)
        else:
            zen, az = astrometry.icrs_to_topo(self.location, float(time), utils.tensor2numpy(ra), utils.tensor2numpy(dec),
                                              dut1, xp, yp)
            angs = torch.as_tensor(np.stack([zen, az]), device=self.device, dtype=self.dtype)
        if store:
            self.conv_cache[key] = angs
        return angs

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
            for k, v in self.conv_cache.items():
                self.conv_cache[k] = v.to(device)


def _astropy_eq2top(location, time, ra, dec):
    """the reference's own transformation (telescope_model.py:469-502: ICRS -> AltAz of astropy, IERS tables and
    all) when astropy is importable in the user's environment, else None.  Host-side setup work, cached by the caller;
    the image this was built in has no astropy, so this branch is untested here and the astrometry chain serves a miss."""
    try:
        from astropy import units
        from astropy.time import Time
        from astropy.coordinates import EarthLocation, AltAz, ICRS
    except Exception:
        return None
    loc = location
    if not isinstance(loc, EarthLocation):
        loc = EarthLocation(lon=location[0] * units.deg, lat=location[1] * units.deg,
                            height=(location[2] if len(location) > 2 else 0.0) * units.m)
    out = ICRS(ra=utils.tensor2numpy(ra) * units.deg, dec=utils.tensor2numpy(dec) * units.deg).transform_to(
        AltAz(location=loc, obstime=Time(time, format='jd')))
    return out.zen.deg, out.az.deg


def JD2LST(jd, longitude):
    """local apparent-ish sidereal time [deg]: GMST (IAU 1982 polynomial) + east longitude"""
    d = np.asarray(jd, dtype=np.float64) - 2451545.0
    T = d / 36525.0
    gmst = 280.46061837 + 360.98564736629 * d + 0.000387933 * T ** 2 - T ** 3 / 38710000.0
    return np.mod(gmst + longitude, 360.0)


def eq2top(location, time, ra, dec):
    """
    Equatorial (ra, dec) [deg] -> topocentric (zen, az) [deg], az East of North, by a pure
    hour-angle rotation at the telescope latitude: a generator of SYNTHETIC inputs (benchmarks,
    tests).  Ignores precession / nutation / aberration, which astropy's ICRS->AltAz
    (telescope_model.py:469-502) and TelescopeModel.eq2top include.
    """
    lon, lat = location[0], location[1]
    H = np.deg2rad(JD2LST(time, lon) - np.asarray(ra, dtype=np.float64))
    d = np.deg2rad(np.asarray(dec, dtype=np.float64))
    p = np.deg2rad(lat)
    x = -np.cos(d) * np.sin(H)
    y = np.sin(d) * np.cos(p) - np.cos(d) * np.sin(p) * np.cos(H)
    z = np.sin(d) * np.sin(p) + np.cos(d) * np.cos(p) * np.cos(H)
    zen = np.rad2deg(np.arccos(np.clip(z, -1.0, 1.0)))
    az = np.mod(np.rad2deg(np.arctan2(x, y)), 360.0)
    return zen, az


def eq2top_device(location, time, ra, dec):
    """eq2top() with torch float64 ops on the device of `ra`: (2, N) tensor (zen, az) [deg]"""
    lon, lat = float(location[0]), float(location[1])
    lst = float(JD2LST(time, lon))
    H = (lst - ra.to(torch.float64)) * D2R
    d = dec.to(torch.float64) * D2R
    p = lat * D2R
    cd, sd_ = torch.cos(d), torch.sin(d)
    cH = torch.cos(H)
    x = -cd * torch.sin(H)
    y = sd_ * math.cos(p) - cd * math.sin(p) * cH
    z = sd_ * math.sin(p) + cd * math.cos(p) * cH
    zen = torch.acos(z.clamp(-1.0, 1.0)) / D2R
    az = torch.remainder(torch.atan2(x, y) / D2R, 360.0)
    return torch.stack([zen, az])


def pointing_vectors(zen, az):
    """s = (sin z sin a, sin z cos a, cos z), float64, (3, P) (telescope_model.py:337-343)"""
    z = zen.to(torch.float64) * D2R
    a = az.to(torch.float64) * D2R
    sz = torch.sin(z)
    return torch.stack([sz * torch.sin(a), sz * torch.cos(a), torch.cos(z)])


class ArrayModel(utils.Module, utils.AntposDict):
    """antenna layout + fringe model (telescope_model.py:142-460)"""
    def __init__(self, antpos, freqs=None, device=None, cache_s=True, cache_depth=None,
                 redtol=1.0, name=None, **kwargs):
        utils.Module.__init__(self, name=name)
        if isinstance(antpos, utils.AntposDict):
            ants, antvecs = antpos.ants, antpos.antvecs
        else:
            ants, antvecs = list(antpos.keys()), list(antpos.values())
        utils.AntposDict.__init__(self, ants, antvecs)
        self.cache_s = cache_s
        self.clear_cache()
        self.redtol = redtol
        self.device = device
        self.cache_depth = cache_depth
        self.set_freqs(freqs)
        (self.reds, self.redvecs, self.bl2red, self.bls, self.redlens, self.redangs,
         self.redtags) = build_reds(self, redtol=redtol, **kwargs)
        if device:
            self.push(device)

    def get_antpos(self, ant):
        return utils.AntposDict.__getitem__(self, ant)

    def __getitem__(self, key):
        if isinstance(key, str):
            return utils.Module.__getitem__(self, key)
        return utils.AntposDict.__getitem__(self, key)

    def get_blvecs(self, bls):
        """baseline vectors antpos[j] - antpos[i] in ENU metres, (Nbl, 3) (:221-239)"""
        if isinstance(bls, tuple) or isinstance(bls[0], (int, np.integer)):
            bls = [bls]
        idx = self._ant_idx
        i1 = torch.as_tensor([idx[b[0]] for b in bls], device=self.antvecs.device)
        i2 = torch.as_tensor([idx[b[1]] for b in bls], device=self.antvecs.device)
        return self.antvecs[i2] - self.antvecs[i1]

    def set_freqs(self, freqs):
        self.freqs = freqs
        if freqs is not None:
            self.freqs = torch.as_tensor(freqs, dtype=_float(), device=self.device)

    def set_freq_index(self, idx=None):
        self._freq_idx = idx

    def clear_cache(self, depth=None):
        if depth is None:
            self.cache = {}
        else:
            utils.clear_cache_depth(self.cache, depth)

    def _freqs_active(self):
        f = self.freqs
        if getattr(self, '_freq_idx', None) is not None:
            f = f[self._freq_idx]
        return f

    def get_s(self, zen, az):
        """cached float64 pointing vectors (3, P) keyed by arr_hash(zen) (:332-348)"""
        key = utils.arr_hash(zen)
        if self.cache_s and key in self.cache:
            return self.cache[key]
        s = pointing_vectors(torch.as_tensor(zen), torch.as_tensor(az)).to(self.device)
        if self.cache_s:
            self.cache[key] = s
            if self.cache_depth is not None:
                self.clear_cache(depth=self.cache_depth)
        return s

    def gen_fringe(self, blvecs, zen, az, conj=False):
        """exp(+-2 pi i nu/c b.s) materialised as (Nbl, Nf, P) complex (:310-358)"""
        s = self.get_s(zen, az)
        if not s.is_cuda:
            raise RuntimeError('ArrayModel.gen_fringe needs the model on a GPU device')
        return ops.gen_fringe(blvecs.to(s.device), s, self._freqs_active(), conj=conj,
                              dtype=_float())

    def push(self, device):
        utils.AntposDict.push(self, device)
        if self.freqs is not None:
            self.freqs = utils.push(self.freqs, device)
        if not isinstance(device, torch.dtype):
            self.device = device
            for k, v in self.cache.items():
                if isinstance(v, torch.Tensor):
                    self.cache[k] = v.to(device)

    def get_bls(self, uniq_bls=False, keep_autos=True, min_len=None, max_len=None, min_EW=None,
                max_EW=None, min_NS=None, max_NS=None, min_deg=None, max_deg=None, xants=None):
        """baseline query over the redundant groups (:373-460)"""
        keep = np.ones(len(self.reds), dtype=bool)
        lens = np.asarray(self.redlens)
        angs = np.asarray(self.redangs)
        vecs = np.abs(np.asarray([utils.tensor2numpy(v) for v in self.redvecs]))
        if not keep_autos:
            autos = np.where(np.isclose(lens, 0, atol=self.redtol))[0]
            if len(autos):
                keep[autos[0]] = False
        if min_len is not None:
            keep &= lens >= min_len
        if max_len is not None:
            keep &= lens <= max_len
        if min_EW is not None:
            keep &= vecs[:, 0] >= min_EW
        if max_EW is not None:
            keep &= vecs[:, 0] <= max_EW
        if min_NS is not None:
            keep &= vecs[:, 1] >= min_NS
        if max_NS is not None:
            keep &= vecs[:, 1] <= max_NS
        if min_deg is not None:
            keep &= angs >= min_deg
        if max_deg is not None:
            keep &= angs <= max_deg
        reds = [self.reds[i] for i in np.where(keep)[0]]
        if uniq_bls:
            reds = [r[:1] for r in reds]
        bls = utils.flatten(reds)
        if xants is not None:
            bls = [b for b in bls if b[0] not in xants and b[1] not in xants]
        return bls

    def to_antpos(self):
        return utils.AntposDict(self.ants, self.antvecs)


def build_reds(antpos, bls=None, redtol=1.0, min_len=None, max_len=None, skip_reds=False, **kw):
    """
    Group baselines (autos + all i<j pairs unless `bls` is given) into redundant sets by
    baseline vector within `redtol` metres; groups sorted by length + angle*redtol/180 and
    baselines sorted inside a group, as the reference does (telescope_model.py:693-942).
    Returns (reds, redvecs, bl2red, bls, redlens, redangs, redtags).
    """
    ants = list(antpos.keys())
    if bls is None:
        bls = [(a, a) for a in ants] + list(itertools.combinations(ants, 2))
    av = utils.tensor2numpy(antpos.antvecs if hasattr(antpos, 'antvecs') else
                            torch.stack([torch.as_tensor(v) for v in antpos.values()]))
    idx = {a: i for i, a in enumerate(ants)}
    vecs = np.asarray([av[idx[b[1]]] - av[idx[b[0]]] for b in bls], dtype=np.float64)
    lens = np.linalg.norm(vecs, axis=1)
    keep = np.ones(len(bls), dtype=bool)
    if min_len is not None:
        keep &= lens >= min_len
    if max_len is not None:
        keep &= lens <= max_len
    reds, rvecs = [], []
    # hash on a redtol grid to avoid the O(Nbl^2) scan; neighbours cells are checked so the
    # grouping equals the first-match-within-redtol rule of the reference
    cells = {}
    for i, bl in enumerate(bls):
        if not keep[i]:
            continue
        v = vecs[i]
        g = None
        if not skip_reds:
            c = tuple(np.floor(v / redtol).astype(np.int64))
            best = None
            for off in itertools.product((-1, 0, 1), repeat=3):
                for k in cells.get((c[0] + off[0], c[1] + off[1], c[2] + off[2]), ()):
                    if np.linalg.norm(rvecs[k] - v) < redtol and (best is None or k < best):
                        best = k
            g = best
        if g is None:
            reds.append([bl])
            rvecs.append(v)
            if not skip_reds:
                cells.setdefault(tuple(np.floor(v / redtol).astype(np.int64)), []).append(len(reds) - 1)
        else:
            reds[g].append(bl)
    rl = [float(np.linalg.norm(v)) for v in rvecs]
    ra = []
    for v in rvecs:
        ang = math.degrees(math.atan2(v[1], v[0]))
        if v[1] < 0:
            ang += 180.0
        if abs(v[1]) < redtol:
            ang = 0.0
        ra.append(ang)
    order = np.argsort(np.array(rl) + np.array(ra) * redtol / 180.0, kind='stable') if len(rl) else []
    reds = [sorted(reds[i]) for i in order]
    redvecs = [torch.as_tensor(rvecs[i]) for i in order]
    redlens = [rl[i] for i in order]
    redangs = [ra[i] for i in order]
    redtags = ['{:03.0f}_{:03.0f}'.format(a, b) for a, b in zip(redlens, redangs)]
    all_bls = utils.flatten(reds)
    bl2red = {}
    if not skip_reds:
        for i, r in enumerate(reds):
            for bl in r:
                bl2red[bl] = i
    return reds, redvecs, bl2red, all_bls, redlens, redangs, redtags
