// eq2top.hip -- ICRS (ra, dec) -> topocentric (zenith angle, azimuth), float64, one thread per direction.
//
// Replaces the per-direction part of telescope_model.eq2top (telescope_model.py:469-502: astropy's
// SkyCoord(icrs).transform_to(AltAz)).  Everything that depends on the observation time only -- the
// 3 x 3 matrix M = L(lat) R3(GAST + lon) N P B and the observer's velocity -- is computed on the host
// (bayeslim_amd/astrometry.py); a thread forms the ICRS unit vector, applies the (relativistic) annual
// aberration, rotates by M, adds the diurnal aberration (eastward, first order) and converts to
// zen = atan2(hypot(E, N), U), az = atan2(E, N) mod 360 (East of North, as gen_fringe's pointing
// vectors expect: telescope_model.py:337-343).  HBM traffic: 16 B in, 16 B out per direction.
#include <hip/hip_runtime.h>
#include "rime_common.h"

namespace rime {

struct Eq2TopArgs {
    const double* ra; const double* dec;      // [N] degrees
    double* zen; double* az;                  // [N] degrees
    double M[9];                              // ICRS -> (East, North, Up)
    double v[3];                              // observer velocity / c, ICRS axes
    double bm1;                               // sqrt(1 - |v|^2)
    double vd;                                // eastward site velocity / c
    int N;
};

__global__ void __launch_bounds__(256) eq2top_kernel(Eq2TopArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.N) return;
    constexpr double D2R = 0.017453292519943295769, R2D = 57.295779513082320877;
    double sa, ca, sd, cd;
    sincos(A.ra[i] * D2R, &sa, &ca);
    sincos(A.dec[i] * D2R, &sd, &cd);
    double p0 = cd * ca, p1 = cd * sa, p2 = sd;
    // aberration: q = (bm1 p + (1 + p.v / (1 + bm1)) v) / (1 + p.v), renormalised
    const double pdv = p0 * A.v[0] + p1 * A.v[1] + p2 * A.v[2];
    const double w = 1.0 + pdv / (1.0 + A.bm1), inv = 1.0 / (1.0 + pdv);
    p0 = (A.bm1 * p0 + w * A.v[0]) * inv;
    p1 = (A.bm1 * p1 + w * A.v[1]) * inv;
    p2 = (A.bm1 * p2 + w * A.v[2]) * inv;
    const double rn = rsqrt(p0 * p0 + p1 * p1 + p2 * p2);
    p0 *= rn; p1 *= rn; p2 *= rn;
    double e = A.M[0] * p0 + A.M[1] * p1 + A.M[2] * p2 + A.vd;
    double n = A.M[3] * p0 + A.M[4] * p1 + A.M[5] * p2;
    double u = A.M[6] * p0 + A.M[7] * p1 + A.M[8] * p2;
    A.zen[i] = atan2(hypot(e, n), u) * R2D;
    double az = atan2(e, n) * R2D;
    if (az < 0.0) az += 360.0;
    if (az >= 360.0) az -= 360.0;
    A.az[i] = az;
}

} // namespace rime

extern "C" int rime_eq2top(const double* ra_deg, const double* dec_deg, int N, const double* M_host,
                           const double* vbary_host, double vdiurnal, double* zen_deg, double* az_deg, void* stream)
{
    if (!ra_deg || !dec_deg || !M_host || !vbary_host || !zen_deg || !az_deg || N < 0) return RIME_EINVAL;
    if (N == 0) return RIME_OK;
    rime::Eq2TopArgs A{};
    A.ra = ra_deg; A.dec = dec_deg; A.zen = zen_deg; A.az = az_deg; A.N = N; A.vd = vdiurnal;
    for (int k = 0; k < 9; ++k) A.M[k] = M_host[k];
    double v2 = 0.0;
    for (int k = 0; k < 3; ++k) { A.v[k] = vbary_host[k]; v2 += vbary_host[k] * vbary_host[k]; }
    if (!(v2 < 1.0)) return RIME_EINVAL;
    A.bm1 = sqrt(1.0 - v2);
    hipLaunchKernelGGL(rime::eq2top_kernel, dim3((N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A);
    return rime::check_launch();
}
