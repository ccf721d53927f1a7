"""
HEALPix RING-scheme helpers (Gorski et al. 2005) used for sky-pixel centres and for
`pixtype='healpix'` beam interpolation.  The reference calls healpy (`healpy.pix2ang`,
`healpy.get_interp_weights`, utils.py:765-769); healpy is not available in this image, so these
are written from the published pixelisation.  PARITY UNPINNED against healpy;
checked by self-consistency tests (areas, ring structure, weights sum to 1, exact at centres).
Setup-time host code (numpy, float64).
"""
import numpy as np


def nside2npix(nside):
    return 12 * int(nside) ** 2


def nside2pixarea(nside):
    return 4.0 * np.pi / nside2npix(nside)


def pix2ang(nside, ipix=None):
    """(colatitude, longitude) [rad] of RING pixel centres"""
    nside = int(nside)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    p = np.arange(npix, dtype=np.int64) if ipix is None else np.asarray(ipix, dtype=np.int64)
    z = np.empty(p.shape)
    phi = np.empty(p.shape)
    north = p < ncap
    south = p >= npix - ncap
    belt = ~(north | south)
    if north.any():
        q = p[north]
        i = ((1 + np.sqrt(1 + 2 * q.astype(np.float64))) / 2).astype(np.int64)
        # guard the float sqrt at ring boundaries
        i = np.where(2 * i * (i - 1) > q, i - 1, i)
        i = np.where(2 * (i + 1) * i <= q, i + 1, i)
        j = q + 1 - 2 * i * (i - 1)
        z[north] = 1.0 - i * i / (3.0 * nside * nside)
        phi[north] = (j - 0.5) * np.pi / (2.0 * i)
    if belt.any():
        q = p[belt] - ncap
        i = q // (4 * nside) + nside
        j = q % (4 * nside) + 1
        s = np.where((i + nside) % 2 == 1, 1.0, 0.5)
        z[belt] = (2 * nside - i) * 2.0 / (3.0 * nside)
        phi[belt] = (j - s) * np.pi / (2.0 * nside)
    if south.any():
        q = npix - p[south]
        i = ((1 + np.sqrt(2 * q.astype(np.float64) - 1)) / 2).astype(np.int64)
        i = np.where(2 * i * (i - 1) >= q, i - 1, i)
        i = np.where(2 * (i + 1) * i < q, i + 1, i)
        j = 4 * i + 1 - (q - 2 * i * (i - 1))
        z[south] = -1.0 + i * i / (3.0 * nside * nside)
        phi[south] = (j - 0.5) * np.pi / (2.0 * i)
    return np.arccos(z), phi


def _ring_info(nside, ring):
    """start pixel, pixels in ring, colatitude, half-pixel shift flag for ring index 1..4nside-1"""
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    north = np.where(ring > 2 * nside, 4 * nside - ring, ring)
    cap = north < nside
    nr = np.where(cap, 4 * north, 4 * nside)
    tmp = north.astype(np.float64) ** 2 * (4.0 / npix)
    theta_cap = np.arctan2(np.sqrt(np.maximum(tmp * (2 - tmp), 0)), 1 - tmp)
    theta_belt = np.arccos(np.clip((2 * nside - north) * (2.0 / (3.0 * nside)), -1, 1))
    theta = np.where(cap, theta_cap, theta_belt)
    shifted = np.where(cap, True, ((north - nside) & 1) == 0)
    start = np.where(cap, 2 * north * (north - 1), ncap + (north - nside) * nr)
    flip = north != ring
    theta = np.where(flip, np.pi - theta, theta)
    start = np.where(flip, npix - start - nr, start)
    return start, nr, theta, shifted


def get_interp_weights(nside, theta, phi):
    """
    Bilinear interpolation on the RING scheme: 4 pixels and 4 weights per direction, arrays of
    shape (4, N) like healpy.get_interp_weights.  Two pixels in the ring above, two in the
    ring below, linear in longitude within each ring and in colatitude between rings; beyond
    the first / last ring the missing ring is replaced by the four polar pixels.
    """
    nside = int(nside)
    npix = 12 * nside * nside
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    phi = np.mod(np.atleast_1d(np.asarray(phi, dtype=np.float64)), 2 * np.pi)
    z = np.cos(theta)
    az = np.abs(z)
    ir_eq = (nside * (2.0 - 1.5 * z)).astype(np.int64)
    ir_cap = (nside * np.sqrt(3.0 * (1.0 - az))).astype(np.int64)
    ir1 = np.where(az <= 2.0 / 3.0, ir_eq, np.where(z > 0, ir_cap, 4 * nside - ir_cap - 1))
    ir2 = ir1 + 1
    N = len(theta)
    pix = np.zeros((4, N), dtype=np.int64)
    wgt = np.zeros((4, N))
    theta1 = np.zeros(N)
    theta2 = np.zeros(N)
    for (ir, slot, thstore) in ((ir1, 0, theta1), (ir2, 2, theta2)):
        ok = (ir > 0) & (ir < 4 * nside)
        irc = np.clip(ir, 1, 4 * nside - 1)
        sp, nr, th, sh = _ring_info(nside, irc)
        dphi = 2 * np.pi / nr
        tmp = phi / dphi - 0.5 * sh
        i1 = np.floor(tmp).astype(np.int64)
        w1 = (phi - (i1 + 0.5 * sh) * dphi) / dphi
        i2 = i1 + 1
        i1 = np.where(i1 < 0, i1 + nr, i1)
        i2 = np.where(i2 >= nr, i2 - nr, i2)
        pix[slot] = np.where(ok, sp + i1, 0)
        pix[slot + 1] = np.where(ok, sp + i2, 0)
        wgt[slot] = np.where(ok, 1 - w1, 0)
        wgt[slot + 1] = np.where(ok, w1, 0)
        thstore[:] = th
    npole = ir1 == 0
    spole = ir2 == 4 * nside
    mid = ~(npole | spole)
    wt = np.zeros(N)
    wt[mid] = (theta[mid] - theta1[mid]) / (theta2[mid] - theta1[mid])
    wgt[0, mid] *= 1 - wt[mid]
    wgt[1, mid] *= 1 - wt[mid]
    wgt[2, mid] *= wt[mid]
    wgt[3, mid] *= wt[mid]
    if npole.any():
        w = theta[npole] / theta2[npole]
        fac = (1 - w) * 0.25
        wgt[2, npole] = wgt[2, npole] * w + fac
        wgt[3, npole] = wgt[3, npole] * w + fac
        wgt[0, npole] = fac
        wgt[1, npole] = fac
        pix[0, npole] = (pix[2, npole] + 2) & 3
        pix[1, npole] = (pix[3, npole] + 2) & 3
    if spole.any():
        w = (theta[spole] - theta1[spole]) / (np.pi - theta1[spole])
        fac = w * 0.25
        wgt[0, spole] = wgt[0, spole] * (1 - w) + fac
        wgt[1, spole] = wgt[1, spole] * (1 - w) + fac
        wgt[2, spole] = fac
        wgt[3, spole] = fac
        pix[2, spole] = ((pix[0, spole] + 2) & 3) + npix - 4
        pix[3, spole] = ((pix[1, spole] + 2) & 3) + npix - 4
    return pix, wgt
