// fringe_mfma_lab.hip -- correctness + timing of the antenna-factored MFMA forward kernel against
// the VALU baseline-formulation kernel on the same inputs (development tool).
// RIME_MF_FWD_V2=1 in the environment selects the interleaved forward kernel (fringe_mfma.hip, second form)
#define RIME_BUILD_FWD_V2 1
#include "../bayeslim_amd/csrc/fringe.hip"
#include "bin/lab_src/fringe_mfma.hip"      // the LAB source: tools/lab/make_lab_source.sh (product file + tools/lab/fringe_mfma_lab.patch); build with -Ibayeslim_amd/csrc
#include <vector>
#include <random>
#include <complex>
namespace rime { char g_last_error[256]; }
using namespace rime;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    int Nant = argc > 1 ? atoi(argv[1]) : 128;
    int Nt = 2, Nf = argc > 2 ? atoi(argv[2]) : 256, P = argc > 3 ? atoi(argv[3]) : 98304;
    int Nbl = Nant * (Nant - 1) / 2;
    std::mt19937 rng(0);
    std::normal_distribution<double> nd(0, 60.0);
    std::uniform_real_distribution<double> ud(0, 1);
    std::vector<double> ant(3 * Nant), bl(3 * (size_t)Nbl), sd((size_t)Nt * 3 * P), fr(Nf);
    for (auto& v : ant) v = nd(rng);
    for (int a = 0; a < Nant; ++a) ant[3 * a + 2] *= 0.02;
    std::vector<int> pd(128 * 128, -1), pc(128 * 128, -1);
    {
        int b = 0;
        for (int i = 0; i < Nant; ++i)
            for (int j = i + 1; j < Nant; ++j, ++b) {
                // alternate orientation so both the direct and the conjugate table are exercised
                bool flip = (b % 3 == 0);
                int a1 = flip ? j : i, a2 = flip ? i : j;        // baseline (a1 -> a2): blvec = pos[a2] - pos[a1]
                for (int d = 0; d < 3; ++d) bl[3 * (size_t)b + d] = ant[3 * a2 + d] - ant[3 * a1 + d];
                // V[a1,a2] lives at computed element (i,j) if a1 < a2 ... upper tile always has row<=col tiles
                int ti = a1 / 32, tj = a2 / 32;
                if (ti <= tj) pd[a1 * 128 + a2] = b;             // element (a1,a2) is computed directly
                else pc[a2 * 128 + a1] = b;                      // only (a2,a1) is computed: conj
            }
    }
    for (int t = 0; t < Nt; ++t)
        for (int p = 0; p < P; ++p) {
            double cz = ud(rng), az = 6.283185307 * ud(rng), sz = sqrt(1 - cz * cz);
            sd[((size_t)t * 3 + 0) * P + p] = sz * sin(az);
            sd[((size_t)t * 3 + 1) * P + p] = sz * cos(az);
            sd[((size_t)t * 3 + 2) * P + p] = cz;
        }
    for (int f = 0; f < Nf; ++f) fr[f] = 120e6 + 60e6 * f / (Nf - 1);
    size_t npsky = (size_t)Nt * Nf * P;
    std::vector<float> ps(npsky), sc((size_t)Nt * Nf);
    std::normal_distribution<double> nd1(0, 1.0);
    for (size_t i = 0; i < npsky; ++i) {
        // beam-like dynamic range: random sky x an envelope spanning 6 decades
        double env = exp(-13.8 * ud(rng));
        ps[i] = (float)(nd1(rng) * env * 3e-5);
        if (argc > 4) ps[i] = fabsf(ps[i]);       // 4th argument: non-negative sky (sign-free fast path)
    }
    for (int tf = 0; tf < Nt * Nf; ++tf) {
        float amax = 0;
        for (int p = 0; p < P; ++p) amax = fmaxf(amax, fabsf(ps[(size_t)tf * P + p]));
        sc[tf] = amax > 0 ? exp2f(floorf(log2f(16384.0f / amax))) : 1.0f;
    }
    double *dant, *dbl, *dsd, *dfr; float *dps, *dsc, *dv1, *dv2, *dws; int *dpd, *dpc;
    size_t nvis = (size_t)Nbl * Nt * Nf * 2;
    CHK(hipMalloc(&dant, ant.size() * 8)); CHK(hipMalloc(&dbl, bl.size() * 8)); CHK(hipMalloc(&dsd, sd.size() * 8));
    CHK(hipMalloc(&dfr, fr.size() * 8)); CHK(hipMalloc(&dps, npsky * 4)); CHK(hipMalloc(&dsc, sc.size() * 4));
    CHK(hipMalloc(&dv1, nvis * 4)); CHK(hipMalloc(&dv2, nvis * 4)); size_t wsb = std::max(std::max(rime_fringe_sum_workspace(RIME_F32, Nbl, Nt, Nf, P, 1, 1, 0, 0), rime_fringe_sum_workspace(RIME_F32, Nbl, Nt, Nf, P, 1, 1, 0, 1)), std::max(rime_fringe_ant_workspace(Nbl, Nt, Nf, P), rime_fringe_ant_bwd_workspace(Nbl, Nt, Nf))) + 256;
    CHK(hipMalloc(&dws, wsb));
    CHK(hipMalloc(&dpd, pd.size() * 4)); CHK(hipMalloc(&dpc, pc.size() * 4));
    CHK(hipMemcpy(dant, ant.data(), ant.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dbl, bl.data(), bl.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dsd, sd.data(), sd.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dfr, fr.data(), fr.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dps, ps.data(), npsky * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dpd, pd.data(), pd.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dpc, pc.data(), pc.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(dv2, 0, nvis * 4));
    float* drm = nullptr;                     // per-row minimum of psky (rows without negatives skip the sign masks)
    {
        std::vector<float> rm((size_t)Nt * Nf);
        for (int tf = 0; tf < Nt * Nf; ++tf) { float m = 0; for (int p = 0; p < P; ++p) m = fminf(m, ps[(size_t)tf * P + p]); rm[tf] = m; }
        CHK(hipMalloc(&drm, rm.size() * 4)); CHK(hipMemcpy(drm, rm.data(), rm.size() * 4, hipMemcpyHostToDevice));
    }

    int off[2] = {0, Nbl};
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms1 = 0, ms2 = 0;
    for (int r = 0; r < 2; ++r) {
        CHK(hipEventRecord(e0));
        int rc = rime_fringe_sum_fwd(RIME_F32, dbl, dsd, dfr, dps, off, nullptr, Nbl, Nt, Nf, P, 1, 1, 0, 1,
                                     1, fr[0], fr[1] - fr[0], 300.0, nullptr, dv1, dws, wsb, 0);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        if (rc) { printf("valu rc=%d\n", rc); return 1; }
        CHK(hipEventElapsedTime(&ms1, e0, e1));
    }
    float ms2min = 1e9f;
    for (int r = 0; r < 6; ++r) {
        CHK(hipEventRecord(e0));
        int rc = rime_fringe_ant_fwd(dant, dsd, dfr, dps, dsc, drm, dpd, dpc, Nant, Nbl, Nt, Nf, P,
                                     (long long)Nf * P, (long long)P, 1LL, 1, dv2, dws, wsb, 0);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        if (rc) { printf("mfma rc=%d (%s)\n", rc, rime_last_error()); return 1; }
        CHK(hipEventElapsedTime(&ms2, e0, e1));
        if (r > 0) ms2min = fminf(ms2min, ms2);
    }
    printf("MFMA fwd min of 5: %.3f ms (last %.3f)\n", ms2min, ms2);
    std::vector<float> v1(nvis), v2(nvis);
    CHK(hipMemcpy(v1.data(), dv1, nvis * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(v2.data(), dv2, nvis * 4, hipMemcpyDeviceToHost));
    double vmax = 0, dmax = 0, d2 = 0, n2 = 0;
    for (size_t i = 0; i < nvis; ++i) {
        vmax = fmax(vmax, fabs(v1[i])); dmax = fmax(dmax, fabs((double)v1[i] - v2[i]));
        d2 += ((double)v1[i] - v2[i]) * ((double)v1[i] - v2[i]); n2 += (double)v1[i] * v1[i];
    }
    {   // ---------------- backward: random gvis, VALU vs MFMA gpsky ----------------
        std::vector<float> gv(nvis), gsc((size_t)Nt * Nf);
        for (auto& v : gv) v = (float)nd1(rng);
        for (int tf = 0; tf < Nt * Nf; ++tf) {
            int t = tf / Nf, f = tf % Nf;
            float amax = 0;
            for (int b = 0; b < Nbl; ++b)
                for (int c = 0; c < 2; ++c) amax = fmaxf(amax, fabsf(gv[(((size_t)b * Nt + t) * Nf + f) * 2 + c]));
            gsc[tf] = amax > 0 ? exp2f(floorf(log2f(16384.0f / amax))) : 1.0f;
        }
        float *dgv, *dgsc, *dg1, *dg2;
        CHK(hipMalloc(&dgv, nvis * 4)); CHK(hipMalloc(&dgsc, gsc.size() * 4));
        CHK(hipMalloc(&dg1, npsky * 4)); CHK(hipMalloc(&dg2, npsky * 4));
        CHK(hipMemcpy(dgv, gv.data(), nvis * 4, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dgsc, gsc.data(), gsc.size() * 4, hipMemcpyHostToDevice));
        float mb1 = 0, mb2 = 0;
        for (int r = 0; r < 2; ++r) {
            CHK(hipEventRecord(e0));
            int rc = rime_fringe_sum_bwd(RIME_F32, dbl, dsd, dfr, dgv, off, nullptr, Nbl, Nt, Nf, P, 1, 1, 0, 1,
                                         1, fr[0], fr[1] - fr[0], 300.0, nullptr, dg1, dws, wsb, 0);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            if (rc) { printf("valu bwd rc=%d\n", rc); return 1; }
            CHK(hipEventElapsedTime(&mb1, e0, e1));
        }
        for (int r = 0; r < 2; ++r) {
            CHK(hipEventRecord(e0));
            int rc = rime_fringe_ant_bwd(dant, dsd, dfr, dgv, dgsc, dpd, dpc, Nant, Nbl, Nt, Nf, P,
                                         (long long)Nf * P, (long long)P, 1LL, 1, dg2, dws, wsb, 0);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            if (rc) { printf("mfma bwd rc=%d (%s)\n", rc, rime_last_error()); return 1; }
            CHK(hipEventElapsedTime(&mb2, e0, e1));
        }
        std::vector<float> g1(npsky), g2(npsky);
        CHK(hipMemcpy(g1.data(), dg1, npsky * 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(g2.data(), dg2, npsky * 4, hipMemcpyDeviceToHost));
        double gmax = 0, dmx = 0, dd = 0, nn = 0;
        for (size_t i = 0; i < npsky; ++i) {
            gmax = fmax(gmax, fabs(g1[i])); dmx = fmax(dmx, fabs((double)g1[i] - g2[i]));
            dd += ((double)g1[i] - g2[i]) * ((double)g1[i] - g2[i]); nn += (double)g1[i] * g1[i];
        }
        printf("BWD VALU %8.3f ms | MFMA %8.3f ms (x%.2f) | max|g|=%.3e max|diff|=%.3e (%.2e of max) rms rel %.2e\n",
               mb1, mb2, mb1 / mb2, gmax, dmx, dmx / gmax, sqrt(dd / nn));
    }
    double E = (double)Nbl * Nf * P * Nt;
    printf("Nant=%d Nbl=%d Nt=%d Nf=%d P=%d\n", Nant, Nbl, Nt, Nf, P);
    printf("VALU kernel : %8.3f ms  %.3e elem/s\n", ms1, E / ms1 * 1e3);
    printf("MFMA kernel : %8.3f ms  %.3e elem/s  (x%.2f)\n", ms2, E / ms2 * 1e3, ms1 / ms2);
    printf("max|v|=%.4e  max|diff|=%.4e  (%.2e of max)  rms rel %.2e\n", vmax, dmax, dmax / vmax, sqrt(d2 / n2));
    return 0;
}
extern "C" const char* rime_last_error(void) { return rime::g_last_error; }
