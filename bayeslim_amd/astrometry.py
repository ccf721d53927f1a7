"""
ICRS (ra, dec) -> topocentric (zenith angle, azimuth) without astropy: the transformation
`telescope_model.eq2top` (telescope_model.py:469-502 of the reference) obtains from
astropy's `SkyCoord(..., frame='icrs').transform_to(AltAz(location, obstime))`.

Per observation time the host builds, in float64,
    M = L(lat) . R3(GAST + lon) . N(dpsi, deps) . P(IAU 2006) . B(frame bias)        (3 x 3)
and the observer's velocity (Earth's orbital velocity, annual aberration; the rotation of the
Earth, diurnal aberration); per direction the device (HIP kernel `rime_eq2top`, or the numpy
restatement `icrs_to_topo` for host tensors) applies aberration and M and converts to angles.

Terms and their sizes (what the round-1 stand-in, a bare LST rotation, left out):
    precession since J2000      ~ 0.3 deg in 2022        IAU 2006 (P03) polynomials + frame bias
    nutation                    ~ 17 arcsec              31 largest terms of the IAU 1980 series
                                                         (truncation ~ 0.003 arcsec)
    annual aberration           ~ 20.5 arcsec            Earth velocity from the Sun's true longitude
                                                         (Meeus ch. 25; ~ 0.02 arcsec)
    diurnal aberration          ~ 0.3 arcsec             exact
Earth orientation parameters (round 4): UT1-UTC (<= 0.9 s = 13 arcsec of hour angle) and the polar motion
(xp, yp ~ 0.3 arcsec) are INPUTS -- `dut1` [s], `xp`, `yp` [rad] of observation_frame(), interpolated from an IERS
table (`EarthOrientation`, finals2000A / EOP C04 / plain 4-column text: what astropy reads from its IERS files)
when `TelescopeModel(iers_file=...)` is given one, 0 otherwise.  NOT modelled: light deflection by the Sun
(<= 4 mas beyond 45 deg elongation), atmospheric refraction (astropy's AltAz default pressure is 0:
none), the TIO locator s' (< 0.1 mas).  Expected agreement with astropy given the same table: ~10 mas.  PARITY UNPINNED
against astropy itself (absent here); pinned instead to SOFA's published known-answer values
(tests/golden/sofa_vectors.json, tests/test_host_logic.py).
"""
import math
import re

import numpy as np

AS2R = math.pi / (180.0 * 3600.0)
D2R = math.pi / 180.0
TWOPI = 2.0 * math.pi
C_AUDAY = 173.1446326846693          # speed of light [AU / day]

# TAI - UTC [s] from the given UTC date (year, month) on
_LEAP = [(1972, 1, 10), (1972, 7, 11), (1973, 1, 12), (1974, 1, 13), (1975, 1, 14), (1976, 1, 15), (1977, 1, 16),
         (1978, 1, 17), (1979, 1, 18), (1980, 1, 19), (1981, 7, 20), (1982, 7, 21), (1983, 7, 22), (1985, 7, 23),
         (1988, 1, 24), (1990, 1, 25), (1991, 1, 26), (1992, 7, 27), (1993, 7, 28), (1994, 7, 29), (1996, 1, 30),
         (1997, 7, 31), (1999, 1, 32), (2006, 1, 33), (2009, 1, 34), (2012, 7, 35), (2015, 7, 36), (2017, 1, 37)]


def _cal2jd(y, m, d=1):
    a = (14 - m) // 12
    yy, mm = y + 4800 - a, m + 12 * a - 3
    return d + (153 * mm + 2) // 5 + 365 * yy + yy // 4 - yy // 100 + yy // 400 - 32045 - 0.5


_LEAP_JD = [(_cal2jd(y, m), s) for y, m, s in _LEAP]


def tai_minus_utc(jd_utc):
    out = 10.0
    for jd0, s in _LEAP_JD:
        if jd_utc >= jd0:
            out = float(s)
    return out


def tt_centuries(jd_utc):
    """Julian centuries of TT since J2000.0 for a UTC Julian date"""
    return (jd_utc + (tai_minus_utc(jd_utc) + 32.184) / 86400.0 - 2451545.0) / 36525.0


def rx(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, s], [0, -s, c]], dtype=np.float64)


def ry(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=np.float64)


def rz(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float64)


def frame_bias():
    """GCRS -> mean J2000.0 (IAU 2000): R1(-eta0) R2(xi0) R3(da0)"""
    da0, xi0, eta0 = -0.0146 * AS2R, -0.041775 * AS2R * math.sin(84381.448 * AS2R), -0.0068192 * AS2R
    return rx(-eta0) @ ry(xi0) @ rz(da0)


def precession_matrix(T):
    """mean J2000.0 -> mean equator and equinox of date, IAU 2006 (Capitaine et al. 2003 P03) equinox-based angles"""
    zeta = (2.650545 + (2306.083227 + (0.2988499 + (0.01801828 + (-0.000005971 - 0.0000003173 * T) * T) * T) * T) * T) * AS2R
    z = (-2.650545 + (2306.077181 + (1.0927348 + (0.01826837 + (-0.000028596 - 0.0000002904 * T) * T) * T) * T) * T) * AS2R
    theta = ((2004.191903 + (-0.4294934 + (-0.04182264 + (-0.000007089 - 0.0000001274 * T) * T) * T) * T) * T) * AS2R
    return rz(-z) @ ry(theta) @ rz(-zeta)


def mean_obliquity(T):
    """IAU 2006 mean obliquity of the ecliptic [rad]"""
    return (84381.406 + (-46.836769 + (-0.0001831 + (0.00200340 + (-0.000000576 - 0.0000000434 * T) * T) * T) * T) * T) * AS2R


# (D, M, M', F, Omega multipliers; dpsi, dpsi_T, deps, deps_T in 0.0001 arcsec): the 31 largest terms of the
# IAU 1980 nutation series (Seidelmann 1982; Meeus, Astronomical Algorithms, table 22.A)
_NUT = [
    (0, 0, 0, 0, 1, -171996, -174.2, 92025, 8.9), (-2, 0, 0, 2, 2, -13187, -1.6, 5736, -3.1),
    (0, 0, 0, 2, 2, -2274, -0.2, 977, -0.5), (0, 0, 0, 0, 2, 2062, 0.2, -895, 0.5),
    (0, 1, 0, 0, 0, 1426, -3.4, 54, -0.1), (0, 0, 1, 0, 0, 712, 0.1, -7, 0.0),
    (-2, 1, 0, 2, 2, -517, 1.2, 224, -0.6), (0, 0, 0, 2, 1, -386, -0.4, 200, 0.0),
    (0, 0, 1, 2, 2, -301, 0.0, 129, -0.1), (-2, -1, 0, 2, 2, 217, -0.5, -95, 0.3),
    (-2, 0, 1, 0, 0, -158, 0.0, 0, 0.0), (-2, 0, 0, 2, 1, 129, 0.1, -70, 0.0),
    (0, 0, -1, 2, 2, 123, 0.0, -53, 0.0), (2, 0, 0, 0, 0, 63, 0.0, 0, 0.0),
    (0, 0, 1, 0, 1, 63, 0.1, -33, 0.0), (2, 0, -1, 2, 2, -59, 0.0, 26, 0.0),
    (0, 0, -1, 0, 1, -58, -0.1, 32, 0.0), (0, 0, 1, 2, 1, -51, 0.0, 27, 0.0),
    (-2, 0, 2, 0, 0, 48, 0.0, 0, 0.0), (0, 0, -2, 2, 1, 46, 0.0, -24, 0.0),
    (2, 0, 0, 2, 2, -38, 0.0, 16, 0.0), (0, 0, 2, 2, 2, -31, 0.0, 13, 0.0),
    (0, 0, 2, 0, 0, 29, 0.0, 0, 0.0), (-2, 0, 1, 2, 2, 29, 0.0, -12, 0.0),
    (0, 0, 0, 2, 0, 26, 0.0, 0, 0.0), (-2, 0, 0, 2, 0, -22, 0.0, 0, 0.0),
    (0, 0, -1, 2, 1, 21, 0.0, -10, 0.0), (0, 2, 0, 0, 0, 17, -0.1, 0, 0.0),
    (2, 0, -1, 0, 1, 16, 0.0, -8, 0.0), (-2, 2, 0, 2, 2, -16, 0.1, 7, 0.0),
    (0, 1, 0, 0, 1, -15, 0.0, 9, 0.0),
]


def nutation(T):
    """(dpsi, deps) [rad]: truncated IAU 1980 series, fundamental arguments of Meeus ch. 22"""
    D = (297.85036 + 445267.111480 * T - 0.0019142 * T * T + T ** 3 / 189474.0) * D2R
    M = (357.52772 + 35999.050340 * T - 0.0001603 * T * T - T ** 3 / 300000.0) * D2R
    Mp = (134.96298 + 477198.867398 * T + 0.0086972 * T * T + T ** 3 / 56250.0) * D2R
    F = (93.27191 + 483202.017538 * T - 0.0036825 * T * T + T ** 3 / 327270.0) * D2R
    Om = (125.04452 - 1934.136261 * T + 0.0020708 * T * T + T ** 3 / 450000.0) * D2R
    dpsi = deps = 0.0
    for d, m, mp, f, om, ps, pst, ep, ept in _NUT:
        arg = d * D + m * M + mp * Mp + f * F + om * Om
        dpsi += (ps + pst * T) * math.sin(arg)
        deps += (ep + ept * T) * math.cos(arg)
    return dpsi * 1e-4 * AS2R, deps * 1e-4 * AS2R


def nutation_matrix(eps0, dpsi, deps):
    """mean equator and equinox of date -> true: R1(-(eps0 + deps)) R3(-dpsi) R1(eps0)"""
    return rx(-(eps0 + deps)) @ rz(-dpsi) @ rx(eps0)


def era(jd_ut1):
    """Earth rotation angle (IAU 2000) [rad]"""
    d = jd_ut1 - 2451545.0
    f = math.fmod(jd_ut1, 1.0)
    return math.fmod(TWOPI * (f + 0.7790572732640 + 0.00273781191135448 * d), TWOPI) % TWOPI


def gmst(jd_ut1, T):
    """Greenwich mean sidereal time, IAU 2006 [rad]"""
    poly = (0.014506 + (4612.156534 + (1.3915817 + (-0.00000044 + (-0.000029956 - 0.0000000368 * T) * T) * T) * T) * T) * AS2R
    return (era(jd_ut1) + poly) % TWOPI


def gast(jd_ut1, T, dpsi, eps0):
    """Greenwich apparent sidereal time: GMST + equation of the equinoxes (with the two largest complementary terms)"""
    Om = (125.04452 - 1934.136261 * T) * D2R
    ee = dpsi * math.cos(eps0) + (0.00264 * math.sin(Om) + 0.000063 * math.sin(2 * Om)) * AS2R
    return (gmst(jd_ut1, T) + ee) % TWOPI


def earth_velocity(T):
    """
    Earth's heliocentric velocity / c in the mean equator and equinox of date, from the Sun's true
    longitude (Meeus ch. 25) and the constant of aberration (ch. 23): v = kappa (sin L - e sin pi,
    -(cos L - e cos pi), 0) in ecliptic axes.  ~1e-3 relative (planetary perturbations, Sun-barycentre).
    """
    L0 = 280.46646 + 36000.76983 * T + 0.0003032 * T * T
    M = (357.52911 + 35999.05029 * T - 0.0001537 * T * T) * D2R
    e = 0.016708634 - 0.000042037 * T - 0.0000001267 * T * T
    C = ((1.914602 - 0.004817 * T - 0.000014 * T * T) * math.sin(M) + (0.019993 - 0.000101 * T) * math.sin(2 * M)
         + 0.000289 * math.sin(3 * M))
    lon = (L0 + C) * D2R
    peri = (102.93735 + 1.71946 * T + 0.00046 * T * T) * D2R
    kappa = 20.49552 * AS2R
    vx = kappa * (math.sin(lon) - e * math.sin(peri))
    vy = -kappa * (math.cos(lon) - e * math.cos(peri))
    eps = mean_obliquity(T)
    return np.array([vx, vy * math.cos(eps), vy * math.sin(eps)], dtype=np.float64)


def local_matrix(lat_deg):
    """true-of-date hour-angle frame (x to the meridian, z to the pole) -> local (East, North, Up)"""
    p = lat_deg * D2R
    return np.array([[0.0, 1.0, 0.0], [-math.sin(p), 0.0, math.cos(p)], [math.cos(p), 0.0, math.sin(p)]], dtype=np.float64)


def observation_frame(location, jd_utc, dut1=0.0, xp=0.0, yp=0.0):
    """
    Everything that depends on the observation time only: returns (M, vbary, vdiurnal) with
    M (3, 3) float64 ICRS -> local (East, North, Up); vbary (3,) the observer's barycentric velocity / c
    in ICRS axes (annual aberration); vdiurnal the eastward velocity / c of the site (diurnal aberration).
    location = (lon, lat[, alt]) geodetic degrees (metres), as the reference's TelescopeModel.
    dut1 = UT1 - UTC [s]; xp, yp = coordinates of the celestial intermediate pole in the terrestrial frame [rad]
    (polar motion, W = R1(-yp) R2(-xp) between the Earth rotation and the site's longitude).
    """
    lon, lat = float(location[0]), float(location[1])
    alt = float(location[2]) if len(location) > 2 else 0.0
    T = tt_centuries(jd_utc)
    eps0 = mean_obliquity(T)
    dpsi, deps = nutation(T)
    PB = precession_matrix(T) @ frame_bias()
    NPB = nutation_matrix(eps0, dpsi, deps) @ PB
    theta = gast(jd_utc + dut1 / 86400.0, T, dpsi, eps0)
    if xp == 0.0 and yp == 0.0:
        M = local_matrix(lat) @ rz(theta + lon * D2R) @ NPB
    else:
        M = local_matrix(lat) @ rz(lon * D2R) @ rx(-yp) @ ry(-xp) @ rz(theta) @ NPB
    vbary = PB.T @ earth_velocity(T)                      # mean-of-date -> ICRS axes
    # site velocity from the Earth's rotation: omega * distance from the axis (WGS84)
    a, f = 6378137.0, 1.0 / 298.257223563
    p = lat * D2R
    n = a / math.sqrt(1.0 - (2 * f - f * f) * math.sin(p) ** 2)
    vdiurnal = 7.292115855306589e-5 * (n + alt) * math.cos(p) / 299792458.0
    return M, vbary, vdiurnal


def icrs_to_topo(location, jd_utc, ra, dec, dut1=0.0, xp=0.0, yp=0.0):
    """numpy float64 restatement of the device path: (zen, az) [deg], az East of North"""
    M, vb, vd = observation_frame(location, jd_utc, dut1, xp, yp)
    a, d = np.deg2rad(np.atleast_1d(np.asarray(ra, dtype=np.float64))), np.deg2rad(np.atleast_1d(np.asarray(dec, dtype=np.float64)))
    p = np.stack([np.cos(d) * np.cos(a), np.cos(d) * np.sin(a), np.sin(d)])
    p = aberrate(p, vb)
    s = M @ p
    s[0] = s[0] + vd                                      # diurnal aberration: first order, eastward
    s /= np.linalg.norm(s, axis=0, keepdims=True)
    zen = np.rad2deg(np.arctan2(np.hypot(s[0], s[1]), s[2]))
    az = np.mod(np.rad2deg(np.arctan2(s[0], s[1])), 360.0)
    return zen, az


def aberrate(p, v):
    """relativistic aberration of unit vectors p (3, N) for an observer velocity v / c (3,) (no solar potential term)"""
    v = np.asarray(v, dtype=np.float64).reshape(3, 1)
    pdv = (p * v).sum(0, keepdims=True)
    bm1 = math.sqrt(1.0 - float((v * v).sum()))
    q = (bm1 * p + (1.0 + pdv / (1.0 + bm1)) * v) / (1.0 + pdv)
    return q / np.linalg.norm(q, axis=0, keepdims=True)


class EarthOrientation:
    """
    UT1-UTC and polar motion from an IERS table, linearly interpolated in time: what astropy's ICRS -> AltAz takes from
    its IERS-A / IERS-B files (telescope_model.py:469-502 of the reference goes through astropy.time / astropy.coordinates).
    Accepted text formats, recognised per line:
      * IERS `finals2000A.all` / `.daily` / `.data` (fixed columns: MJD 8-15, PM-x 19-27, PM-y 38-46 [arcsec],
        UT1-UTC 59-68 [s]; Bulletin A values, predictions included);
      * IERS EOP C04 (`year month day MJD x y UT1-UTC ...`, arcsec / s);
      * plain `MJD xp yp UT1-UTC` (arcsec, arcsec, s); `#` starts a comment.
    A leap second makes UT1-UTC jump by +1 s between two rows: the interpolation is done on the continuous UT1-TAI.
    Outside the table the nearest row is used and `extrapolated` is set.
    """
    def __init__(self, mjd, xp_arcsec, yp_arcsec, dut1_s):
        order = np.argsort(np.asarray(mjd, dtype=np.float64))
        self.mjd = np.asarray(mjd, dtype=np.float64)[order]
        self.xp = np.asarray(xp_arcsec, dtype=np.float64)[order] * AS2R
        self.yp = np.asarray(yp_arcsec, dtype=np.float64)[order] * AS2R
        self.dut1 = np.asarray(dut1_s, dtype=np.float64)[order]
        assert len(self.mjd) >= 1 and len(np.unique(self.mjd)) == len(self.mjd), 'IERS table: empty or repeated dates'
        # continuous quantity across leap seconds: UT1 - TAI = (UT1 - UTC) - (TAI - UTC)
        self._ut1_tai = self.dut1 - np.array([tai_minus_utc(m + 2400000.5) for m in self.mjd])
        self.extrapolated = False

    @classmethod
    def from_file(cls, path):
        rows = []
        with open(path) as f:
            for line in f:
                row = cls._parse(line)
                if row is not None:
                    rows.append(row)
        if not rows:
            raise ValueError('no Earth orientation rows recognised in %s' % path)
        a = np.array(rows)
        return cls(a[:, 0], a[:, 1], a[:, 2], a[:, 3])

    # a finals2000A row starts 'yymmdd MJD' in fixed columns (year, month, day two characters each, blank-padded): '17 5 1 57874.00'
    _FINALS_DATE = re.compile(r'^[ \d]\d[ \d]\d[ \d]\d [ \d]\d{4}\.\d\d(?:\s|$)')

    @staticmethod
    def _plausible(mjd, xp, yp, dut1):
        """a row of Earth orientation data: a date of the space age, a pole within 2 arcsec, |UT1-UTC| within a second"""
        return 30000.0 <= mjd <= 100000.0 and abs(xp) <= 2.0 and abs(yp) <= 2.0 and abs(dut1) <= 1.0

    @classmethod
    def _parse(cls, line):
        text = line.split('#')[0].rstrip('\n')
        if not text.strip():
            return None
        row = None
        if cls._FINALS_DATE.match(text):                                                      # finals2000A
            # the date-only rows that follow the last prediction carry no I / P flags and no values: no data (read as
            # plain text their four tokens 'yy m d MJD' used to become a row at MJD 17, or made from_file fail: ADVICE r04)
            if len(text) >= 68 and text[16:17] in ('I', 'P') and text[57:58] in ('I', 'P'):
                try:
                    row = float(text[7:15]), float(text[18:27]), float(text[37:46]), float(text[58:68])
                except ValueError:
                    row = None
        else:
            tok = text.split()
            try:
                if len(tok) >= 7 and all(t.lstrip('-').isdigit() for t in tok[:3]):            # EOP C04
                    row = float(tok[3]), float(tok[4]), float(tok[5]), float(tok[6])
                elif len(tok) >= 4:
                    row = float(tok[0]), float(tok[1]), float(tok[2]), float(tok[3])
            except ValueError:
                row = None
        return row if row is not None and cls._plausible(*row) else None

    def at(self, jd_utc):
        """(dut1 [s], xp [rad], yp [rad]) at a UTC Julian date"""
        mjd = float(jd_utc) - 2400000.5
        if mjd < self.mjd[0] or mjd > self.mjd[-1]:
            self.extrapolated = True
        xp = float(np.interp(mjd, self.mjd, self.xp))
        yp = float(np.interp(mjd, self.mjd, self.yp))
        dut1 = float(np.interp(mjd, self.mjd, self._ut1_tai)) + tai_minus_utc(float(jd_utc))
        return dut1, xp, yp
