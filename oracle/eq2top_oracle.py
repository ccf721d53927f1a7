"""
TEST INFRASTRUCTURE -- not part of the product.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Independent float64 restatement of what `telescope_model.eq2top` (reference telescope_model.py:469-502:
astropy `SkyCoord(icrs).transform_to(AltAz(location, obstime))`, pressure 0 = no refraction) computes:
ICRS (ra, dec) -> topocentric (zenith angle, azimuth East of North) at a UTC Julian date for a geodetic site.

PARITY STATUS: astropy is absent from the reference checkout and from this image, so the parity with astropy ITSELF
stays unpinned.  This oracle is pinned instead, END TO END, to the published known answers of the IAU SOFA
library's validation program t_sofa_c.c (SOFA is the engine under astropy's transformation: erfa): `atci13`
(ICRS -> CIRS), `atio13` (CIRS -> observed), `atco13` (ICRS -> observed), `apco13` (the refraction constants of
that site), `eors`-type equation-of-the-origins values and `hd2ae` (tests/golden/sofa_vectors.json;
tests/test_oracle_golden.py).  The quoted cases carry effects this chain (and the product's) does not model; the
`sofa_case_*` adaptors below put them on the KNOWN-ANSWER side (space motion and parallax of the test star, light
deflection near the Sun, refraction removal with the published A, B constants), and what remains unmodelled (polar
motion 0.38 arcsec in that case, IAU 2000A vs the truncated 1980 nutation) is the stated tolerance.

Deliberately a DIFFERENT factorisation from the product's (bayeslim_amd/astrometry.py: equinox based,
M = L(lat) R3(GAST + lon) N P(zeta, z, theta) B, analytic Earth velocity, matrices to the end):
  * CIO based: Fukushima-Williams angles (IAU 2006, frame bias included) + nutation -> NPB -> CIP (X, Y) ->
    CIO locator s -> GCRS-to-CIRS matrix; Earth rotation angle, no sidereal time anywhere;
  * fundamental arguments from the IERS 2003 expressions (Simon et al. 1994), not Meeus's;
  * Earth velocity by numerical differentiation of a truncated VSOP87 heliocentric position, not from the constant
    of aberration; one aberration for the total (orbital + diurnal) observer velocity, applied classically
    (unit(p + v/c): second-order terms <= 0.5 mas);
  * the local direction from the hour angle by spherical trigonometry, not by a rotation matrix.
The 31-term nutation table is the same PUBLISHED table the product uses (IAU 1980, Seidelmann 1982), typed here a
second time in arcseconds; SOFA's nut80 known answer pins it.  Nothing here imports the product.
"""
import math

import numpy as np

AS = math.pi / 648000.0            # arcsecond in radians
TAU = 2.0 * math.pi
C_KMS = 299792.458
AU_KM = 149597870.7
DAY_S = 86400.0

# (UTC JD from which it holds, TAI - UTC seconds)
_DAT = [(2441317.5, 10), (2441499.5, 11), (2441683.5, 12), (2442048.5, 13), (2442413.5, 14), (2442778.5, 15),
        (2443144.5, 16), (2443509.5, 17), (2443874.5, 18), (2444239.5, 19), (2444786.5, 20), (2445151.5, 21),
        (2445516.5, 22), (2446247.5, 23), (2447161.5, 24), (2447892.5, 25), (2448257.5, 26), (2448804.5, 27),
        (2449169.5, 28), (2449534.5, 29), (2450083.5, 30), (2450630.5, 31), (2451179.5, 32), (2453736.5, 33),
        (2454832.5, 34), (2456109.5, 35), (2457204.5, 36), (2457754.5, 37)]


def dat(jd_utc):
    s = 10
    for jd0, v in _DAT:
        if jd_utc >= jd0:
            s = v
    return float(s)


def tt_century(jd_utc):
    return (jd_utc - 2451545.0 + (dat(jd_utc) + 32.184) / DAY_S) / 36525.0


def rot(axis, a):
    """passive rotation of the frame about axis 1 / 2 / 3 by angle a (the r1 / r2 / r3 of the IAU papers)"""
    c, s = math.cos(a), math.sin(a)
    i, j = {1: (1, 2), 2: (2, 0), 3: (0, 1)}[axis]
    m = np.eye(3)
    m[i, i], m[i, j], m[j, i], m[j, j] = c, s, -s, c
    return m


# ---- precession (Fukushima-Williams, IAU 2006; Hilton et al. 2006 eqs 37-40) --------------------------------
def fw_angles(t):
    gamb = (-0.052928 + (10.556378 + (0.4932044 + (-0.00031238 + (-0.000002788 + 0.0000000260 * t) * t) * t) * t) * t) * AS
    phib = (84381.412819 + (-46.811016 + (0.0511268 + (0.00053289 + (-0.000000440 - 0.0000000176 * t) * t) * t) * t) * t) * AS
    psib = (-0.041775 + (5038.481484 + (1.5584175 + (-0.00018522 + (-0.000026452 - 0.0000000148 * t) * t) * t) * t) * t) * AS
    epsa = (84381.406 + (-46.836769 + (-0.0001831 + (0.00200340 + (-0.000000576 - 0.0000000434 * t) * t) * t) * t) * t) * AS
    return gamb, phib, psib, epsa


def fw_matrix(gamb, phib, psi, eps):
    return rot(1, -eps) @ rot(3, -psi) @ rot(1, phib) @ rot(3, gamb)


# ---- fundamental arguments (IERS Conventions 2003, arcseconds) -----------------------------------------------
def fund_args(t):
    def poly(c):
        return math.fmod(c[0] + (c[1] + (c[2] + (c[3] + c[4] * t) * t) * t) * t, 1296000.0) * AS
    el = poly((485868.249036, 1717915923.2178, 31.8792, 0.051635, -0.00024470))        # Moon's mean anomaly
    elp = poly((1287104.79305, 129596581.0481, -0.5532, 0.000136, -0.00001149))        # Sun's mean anomaly
    f = poly((335779.526232, 1739527262.8478, -12.7512, -0.001037, 0.00000417))        # L - Omega
    d = poly((1072260.70369, 1602961601.2090, -6.3706, 0.006593, -0.00003169))         # elongation
    om = poly((450160.398036, -6962890.5431, 7.4722, 0.007702, -0.00005939))           # node
    return el, elp, f, d, om


# multipliers of (l, l', F, D, Om); longitude sin coefficient + t * rate, obliquity cos coefficient + t * rate [arcsec]
_NUT80 = [
    ((0, 0, 0, 0, 1), -17.1996, -0.01742, 9.2025, 0.00089), ((0, 0, 2, -2, 2), -1.3187, -0.00016, 0.5736, -0.00031),
    ((0, 0, 2, 0, 2), -0.2274, -0.00002, 0.0977, -0.00005), ((0, 0, 0, 0, 2), 0.2062, 0.00002, -0.0895, 0.00005),
    ((0, 1, 0, 0, 0), 0.1426, -0.00034, 0.0054, -0.00001), ((1, 0, 0, 0, 0), 0.0712, 0.00001, -0.0007, 0.0),
    ((0, 1, 2, -2, 2), -0.0517, 0.00012, 0.0224, -0.00006), ((0, 0, 2, 0, 1), -0.0386, -0.00004, 0.0200, 0.0),
    ((1, 0, 2, 0, 2), -0.0301, 0.0, 0.0129, -0.00001), ((0, -1, 2, -2, 2), 0.0217, -0.00005, -0.0095, 0.00003),
    ((1, 0, 0, -2, 0), -0.0158, 0.0, 0.0, 0.0), ((0, 0, 2, -2, 1), 0.0129, 0.00001, -0.0070, 0.0),
    ((-1, 0, 2, 0, 2), 0.0123, 0.0, -0.0053, 0.0), ((0, 0, 0, 2, 0), 0.0063, 0.0, 0.0, 0.0),
    ((1, 0, 0, 0, 1), 0.0063, 0.00001, -0.0033, 0.0), ((-1, 0, 2, 2, 2), -0.0059, 0.0, 0.0026, 0.0),
    ((-1, 0, 0, 0, 1), -0.0058, -0.00001, 0.0032, 0.0), ((1, 0, 2, 0, 1), -0.0051, 0.0, 0.0027, 0.0),
    ((2, 0, 0, -2, 0), 0.0048, 0.0, 0.0, 0.0), ((-2, 0, 2, 0, 1), 0.0046, 0.0, -0.0024, 0.0),
    ((0, 0, 2, 2, 2), -0.0038, 0.0, 0.0016, 0.0), ((2, 0, 2, 0, 2), -0.0031, 0.0, 0.0013, 0.0),
    ((2, 0, 0, 0, 0), 0.0029, 0.0, 0.0, 0.0), ((1, 0, 2, -2, 2), 0.0029, 0.0, -0.0012, 0.0),
    ((0, 0, 2, 0, 0), 0.0026, 0.0, 0.0, 0.0), ((0, 0, 2, -2, 0), -0.0022, 0.0, 0.0, 0.0),
    ((-1, 0, 2, 0, 1), 0.0021, 0.0, -0.0010, 0.0), ((0, 2, 0, 0, 0), 0.0017, -0.00001, 0.0, 0.0),
    ((-1, 0, 0, 2, 1), 0.0016, 0.0, -0.0008, 0.0), ((0, 2, 2, -2, 2), -0.0016, 0.00001, 0.0007, 0.0),
    ((0, 1, 0, 0, 1), -0.0015, 0.0, 0.0009, 0.0),
]


def nutation(t):
    """(dpsi, deps) radians: the 31 largest terms of the IAU 1980 series"""
    args = fund_args(t)
    dp = de = 0.0
    for mult, ps, pst, ep, ept in _NUT80:
        a = sum(m * x for m, x in zip(mult, args))
        dp += (ps + pst * t) * math.sin(a)
        de += (ep + ept * t) * math.cos(a)
    return dp * AS, de * AS


def npb_matrix(t):
    """GCRS -> true equator and equinox of date"""
    gamb, phib, psib, epsa = fw_angles(t)
    dp, de = nutation(t)
    return fw_matrix(gamb, phib, psib + dp, epsa + de)


# ---- CIO locator and the GCRS -> CIRS matrix (Capitaine et al. 2003; IERS Conventions 2010 table 5.2d) -------
def cio_s(t, x, y):
    el, elp, f, d, om = fund_args(t)
    uas = 1e-6 * AS
    s = (94.00 + (3808.65 + (-122.68 + (-72574.11 + (27.98 + 15.62 * t) * t) * t) * t) * t) * uas
    s += (-2640.73 * math.sin(om) - 63.53 * math.sin(2 * om) - 11.75 * math.sin(2 * f - 2 * d + 3 * om)
          - 11.21 * math.sin(2 * f - 2 * d + om) + 4.57 * math.sin(2 * f - 2 * d + 2 * om) - 2.02 * math.sin(2 * f + 3 * om)
          - 1.98 * math.sin(2 * f + om) + 1.72 * math.sin(3 * om) + 1.41 * math.sin(elp + om) + 1.26 * math.sin(elp - om)
          + 0.63 * math.sin(el + om) + 0.63 * math.sin(el - om)) * uas
    s += t * (1.73 * math.sin(om) + 3.57 * math.cos(2 * om)) * uas
    s += t * t * (743.52 * math.sin(om) + 56.91 * math.sin(2 * f - 2 * d + 2 * om) + 9.84 * math.sin(2 * f + 2 * om)
                  - 8.85 * math.sin(2 * om)) * uas
    return s - 0.5 * x * y


def c2i_matrix(t):
    """GCRS -> CIRS from the CIP coordinates of the NPB matrix and the CIO locator; also returns the equation of the
    origins EO = ERA - GAST (from the same two objects)"""
    npb = npb_matrix(t)
    x, y = npb[2, 0], npb[2, 1]
    s = cio_s(t, x, y)
    r2 = x * x + y * y
    e = math.atan2(y, x) if r2 > 0 else 0.0
    dd = math.atan(math.sqrt(r2 / (1.0 - r2)))
    c2i = rot(3, -(e + s)) @ rot(2, dd) @ rot(3, e)
    # the equinox seen from the CIO along the true equator (Wallace & Capitaine 2006)
    ax = x / (1.0 + npb[2, 2])
    xs, ys, zs = 1.0 - ax * x, -ax * y, -x
    p = npb[0, 0] * xs + npb[0, 1] * ys + npb[0, 2] * zs
    q = npb[1, 0] * xs + npb[1, 1] * ys + npb[1, 2] * zs
    eo = s - math.atan2(q, p)
    return c2i, eo, npb


def earth_rotation_angle(jd_utc, dut1=0.0):
    d1 = math.floor(jd_utc - 0.5) + 0.5                    # 0h of the day, exact in float64
    frac = (jd_utc - d1) + dut1 / DAY_S
    tu = (d1 - 2451545.0) + frac                           # days of UT1 since J2000.0 (d1 is ...5: half a turn)
    turns = 0.5 + frac + 0.7790572732640 + 0.00273781191135448 * tu
    return math.fmod(turns, 1.0) * TAU % TAU


# ---- Earth: truncated VSOP87 (heliocentric, ecliptic and equinox of date; Meeus, Astronomical Algorithms, app. III) ---
_L = [[(175347046, 0, 0), (3341656, 4.6692568, 6283.0758500), (34894, 4.62610, 12566.15170), (3497, 2.7441, 5753.3849),
       (3418, 2.8289, 3.5231), (3136, 3.6277, 77713.7715), (2676, 4.4181, 7860.4194), (2343, 6.1352, 3930.2097),
       (1324, 0.7425, 11506.7698), (1273, 2.0371, 529.6910), (1199, 1.1096, 1577.3435), (990, 5.233, 5884.927),
       (902, 2.045, 26.298), (857, 3.508, 398.149), (780, 1.179, 5223.694), (753, 2.533, 5507.553),
       (505, 4.583, 18849.228), (492, 4.205, 775.523), (357, 2.920, 0.067), (317, 5.849, 11790.629)],
      [(628331966747, 0, 0), (206059, 2.678235, 6283.075850), (4303, 2.6351, 12566.1517), (425, 1.590, 3.523),
       (119, 5.796, 26.298), (109, 2.966, 1577.344), (93, 2.59, 18849.23), (72, 1.14, 529.69), (68, 1.87, 398.15)],
      [(52919, 0, 0), (8720, 1.0721, 6283.0758), (309, 0.867, 12566.152)],
      [(289, 5.844, 6283.076), (35, 0, 0)]]
_B = [[(280, 3.199, 84334.662), (102, 5.422, 5507.553), (80, 3.88, 5223.69), (44, 3.70, 2352.87), (32, 4.00, 1577.34)],
      [(9, 3.90, 5507.55), (6, 1.73, 5223.69)]]
_R = [[(100013989, 0, 0), (1670700, 3.0984635, 6283.0758500), (13956, 3.05525, 12566.15170), (3084, 5.1985, 77713.7715),
       (1628, 1.1739, 5753.3849), (1576, 2.8469, 7860.4194), (925, 5.453, 11506.770), (542, 4.564, 3930.210),
       (472, 3.661, 5884.927), (346, 0.964, 5507.553), (329, 5.900, 5223.694), (307, 0.299, 5573.143),
       (243, 4.273, 11790.629), (212, 5.847, 1577.344), (186, 5.022, 10977.079), (175, 3.012, 18849.228)],
      [(103019, 1.107490, 6283.075850), (1721, 1.0644, 12566.1517), (702, 3.142, 0)],
      [(4359, 5.7846, 6283.0758), (124, 5.579, 12566.152)]]


def _vsop(series, tau):
    tot = 0.0
    for k, terms in enumerate(series):
        tot += sum(a * math.cos(b + c * tau) for a, b, c in terms) * tau ** k
    return tot * 1e-8


def earth_heliocentric_gcrs(t):
    """heliocentric position of the Earth [AU] on GCRS axes (Sun - barycentre offset ignored: <= 0.01 AU)"""
    tau = t / 10.0
    lon, lat, r = _vsop(_L, tau), _vsop(_B, tau), _vsop(_R, tau)
    # VSOP87 dynamical frame -> FK5 (Meeus eq. 32.3): -0.09033 arcsec in longitude
    lon -= 0.09033 * AS
    ecl = np.array([r * math.cos(lat) * math.cos(lon), r * math.cos(lat) * math.sin(lon), r * math.sin(lat)])
    gamb, phib, psib, epsa = fw_angles(t)
    mean_of_date = rot(1, -epsa) @ ecl
    pb = fw_matrix(gamb, phib, psib, epsa)                 # GCRS -> mean equator and equinox of date
    return pb.T @ mean_of_date


def earth_velocity_gcrs(t, h_days=0.05):
    """d/dt of the above by central differences [AU / day]"""
    dt = h_days / 36525.0
    return (earth_heliocentric_gcrs(t + dt) - earth_heliocentric_gcrs(t - dt)) / (2.0 * h_days)


# ---- the chain ---------------------------------------------------------------------------------------------
def observer_velocity_gcrs(lon_rad, lat_rad, height_m, theta, c2i, t):
    """observer's velocity / c on GCRS axes: orbital + rotation of the Earth (geodetic site, WGS84)"""
    a, f = 6378137.0, 1.0 / 298.257223563
    e2 = f * (2.0 - f)
    n = a / math.sqrt(1.0 - e2 * math.sin(lat_rad) ** 2)
    rxy = (n + height_m) * math.cos(lat_rad)               # distance from the rotation axis [m]
    om = 7.292115855306589e-5                              # rad / s
    v_site = om * rxy / 1000.0 / C_KMS * np.array([-math.sin(theta), math.cos(theta), 0.0])      # CIRS axes, theta = ERA + lon
    v_orb = earth_velocity_gcrs(t) * AU_KM / DAY_S / C_KMS
    return v_orb + c2i.T @ v_site


def unit_from_radec(ra, dec):
    ra, dec = np.asarray(ra, dtype=np.float64), np.asarray(dec, dtype=np.float64)
    return np.stack([np.cos(dec) * np.cos(ra), np.cos(dec) * np.sin(ra), np.sin(dec)])


def gcrs_to_cirs_radec(p, jd_utc, observer=None, dut1=0.0):
    """aberration (if an observer (lon, lat, h) in rad / m is given: orbital + diurnal; else orbital only) and the
    GCRS -> CIRS rotation of BCRS unit vectors p (3, N); returns (ri, di, eo)"""
    t = tt_century(jd_utc)
    c2i, eo, _ = c2i_matrix(t)
    if observer is not None:
        theta = earth_rotation_angle(jd_utc, dut1) + observer[0]
        v = observer_velocity_gcrs(observer[0], observer[1], observer[2], theta, c2i, t)
    else:
        v = earth_velocity_gcrs(t) * AU_KM / DAY_S / C_KMS
    q = p + v.reshape(3, 1)
    q = q / np.linalg.norm(q, axis=0, keepdims=True)
    w = c2i @ q
    return np.mod(np.arctan2(w[1], w[0]), TAU), np.arcsin(np.clip(w[2], -1, 1)), eo


def hadec_to_zenaz(ha, dec, lat_rad):
    """spherical triangle pole - zenith - star: (zenith distance, azimuth East of North) radians"""
    sp, cp = math.sin(lat_rad), math.cos(lat_rad)
    cz = sp * np.sin(dec) + cp * np.cos(dec) * np.cos(ha)
    # sin z from the other two sides (well conditioned at the zenith, unlike acos(cz))
    ye = -np.cos(dec) * np.sin(ha)
    xn = np.sin(dec) * cp - np.cos(dec) * np.cos(ha) * sp
    zen = np.arctan2(np.hypot(xn, ye), cz)
    return zen, np.mod(np.arctan2(ye, xn), TAU)


def cirs_to_zenaz(ri, di, jd_utc, lon_rad, lat_rad, dut1=0.0, xp=0.0, yp=0.0):
    """CIRS (ri, di) -> (zen, az) radians at a geodetic site: Earth rotation angle, polar motion (xp, yp radians; the
    TIO locator s' < 0.1 mas is dropped), hour angle from the terrestrial longitude, spherical triangle"""
    w = unit_from_radec(ri, di)
    v = rot(1, -yp) @ rot(2, -xp) @ rot(3, earth_rotation_angle(jd_utc, dut1)) @ w        # terrestrial frame
    ha = lon_rad - np.arctan2(v[1], v[0])
    return hadec_to_zenaz(ha, np.arcsin(np.clip(v[2], -1, 1)), lat_rad)


def eq2top(location, jd_utc, ra_deg, dec_deg, dut1=0.0, xp=0.0, yp=0.0):
    """
    ICRS (ra, dec) degrees -> (zen, az) degrees, azimuth East of North, at UTC Julian date `jd_utc` for
    location = (lon, lat[, height m]) geodetic degrees -- the arguments of the reference's eq2top
    (telescope_model.py:469-502).  No refraction (astropy AltAz pressure 0); dut1 seconds and polar motion xp, yp
    (radians) as given -- the product models neither polar motion nor a dut1 table, so comparisons with it pass 0.
    """
    lon, lat = math.radians(float(location[0])), math.radians(float(location[1]))
    h = float(location[2]) if len(location) > 2 else 0.0
    p = unit_from_radec(np.deg2rad(np.atleast_1d(ra_deg)), np.deg2rad(np.atleast_1d(dec_deg)))
    ri, di, _ = gcrs_to_cirs_radec(p, jd_utc, (lon, lat, h), dut1)
    zen, az = cirs_to_zenaz(ri, di, jd_utc, lon, lat, dut1, xp, yp)
    return np.rad2deg(zen), np.rad2deg(az)


# ---- adaptors that put the extra physics of SOFA's test cases on the known-answer side ----------------------------
def sofa_case_star_direction(rc, dc, pr, pd, px_arcsec, rv_kms, jd_tt):
    """BCRS direction at the date of a star with space motion and parallax (SOFA pmpx, first order): pr is dRA/dt
    [rad / yr], pd dDec/dt, parallax arcsec, radial velocity km/s; light deflection by the Sun added (SOFA ldsun)"""
    t = (jd_tt - 2451545.0) / 36525.0
    eh = earth_heliocentric_gcrs(t)
    p = unit_from_radec(rc, dc).reshape(3)
    dt = (jd_tt - 2451545.0) / 365.25 + float(eh @ p) * (AU_KM / C_KMS) / DAY_S / 365.25       # + Roemer delay [yr]
    x, y, z = p
    w = (365.25 * DAY_S / AU_KM) * rv_kms * px_arcsec * AS                                     # radial motion [rad / yr]
    pm = np.array([-pr * y - pd * math.cos(rc) * math.sin(dc) + w * x, pr * x - pd * math.sin(rc) * math.sin(dc) + w * y,
                   pd * math.cos(dc) + w * z])
    q = p + dt * pm - px_arcsec * AS * eh
    q /= np.linalg.norm(q)
    # light deflection by the Sun: e = Sun -> observer, deflection away from the Sun
    em = np.linalg.norm(eh)
    e = eh / em
    wdef = 1.97412574336e-8 / em / max(1.0 + float(q @ e), 1e-6 / max(em * em, 1.0))
    q = q + wdef * np.cross(q, np.cross(e, q))
    return (q / np.linalg.norm(q)).reshape(3, 1)


def sofa_case_remove_refraction(zob, refa, refb):
    """observed -> unrefracted zenith distance with the published constants: dZ = A tan Z + B tan^3 Z, Z observed"""
    tz = math.tan(zob)
    return zob + refa * tz + refb * tz ** 3
