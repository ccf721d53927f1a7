// fringe_lab.hip -- timing ablations of the fused forward fringe kernel (development tool).
// Includes the product kernel source and launches template variants directly.
#include "../bayeslim_amd/csrc/fringe.hip"
#include <vector>
#include <random>
namespace rime { char g_last_error[256]; }
using namespace rime;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Prob { int Nbl, Nt, Nf, P; };

template <int CH, int MODE, int ABL, int WPS>
static void run(const char* name, const FringeArgs& base, int S, int block)
{
    using G = Geom<float, 1, false, CH>;
    FringeArgs A = base;
    const int ntiles = A.Pstride / TP;
    A.S = S; A.tiles_per_split = (ntiles + S - 1) / S;
    dim3 grid((A.bl_cnt + block - 1) / block, (A.Nf + CH - 1) / CH, A.Nt * A.S);
    size_t lds = (3 * TP + CH) * sizeof(double) + (size_t)TP * G::ASTRIDE * sizeof(float);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((fringe_fwd_kernel<float, 1, false, CH, MODE, ABL, WPS>), grid, dim3(block), lds, 0, A);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    CHK(hipGetLastError());
    double E = (double)A.Nbl * A.Nf * A.Pstride * A.Nt;
    printf("%-34s CH=%2d S=%d blk=%3d waves=%6d  %8.3f ms  %.3e elem/s  %.1f TF\n", name, CH, S, block,
           (int)(grid.x * grid.y * grid.z * (block / 64)), best, E / best * 1e3, E * 10 / best * 1e-9);
}

template <int CH, int MODE, int PIX, int WPS>
static void run_bwd(const char* name, const FringeArgs& base, int S)
{
    using G = Geom<float, 1, false, CH>;
    FringeArgs A = base;
    const int ntiles = (A.bl_cnt + TB - 1) / TB;
    A.S = S; A.tiles_per_split = (ntiles + S - 1) / S;
    const int block = 256;
    dim3 grid((A.Pstride + block * PIX - 1) / (block * PIX), (A.Nf + CH - 1) / CH, A.Nt * A.S);
    size_t lds = (3 * TB + CH) * sizeof(double) + (size_t)TB * G::GSTRIDE * sizeof(float);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((fringe_bwd_kernel<float, 1, false, CH, MODE, PIX, WPS>), grid, dim3(block), lds, 0, A);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    CHK(hipGetLastError());
    double E = (double)A.Nbl * A.Nf * A.Pstride * A.Nt;
    printf("BWD %-30s CH=%2d PIX=%d WPS=%d S=%d waves=%6d  %8.3f ms  %.3e elem/s  %.1f TF\n", name, CH, PIX, WPS, S,
           (int)(grid.x * grid.y * grid.z * (block / 64)), best, E / best * 1e3, E * 10 / best * 1e-9);
}

int main()
{
    Prob pr{8128, 2, 256, 108032};
    std::mt19937 rng(0);
    std::normal_distribution<double> nd(0, 100.0);
    std::uniform_real_distribution<double> ud(0, 1);
    std::vector<double> bl(3 * pr.Nbl), sd((size_t)pr.Nt * 3 * pr.P), fr(pr.Nf);
    for (auto& v : bl) v = nd(rng);
    for (int t = 0; t < pr.Nt; ++t)
        for (int p = 0; p < pr.P; ++p) {
            double cz = ud(rng), az = 6.283185307 * ud(rng), sz = sqrt(1 - cz * cz);
            sd[((size_t)t * 3 + 0) * pr.P + p] = sz * sin(az);
            sd[((size_t)t * 3 + 1) * pr.P + p] = sz * cos(az);
            sd[((size_t)t * 3 + 2) * pr.P + p] = cz;
        }
    for (int f = 0; f < pr.Nf; ++f) fr[f] = 120e6 + 60e6 * f / (pr.Nf - 1);
    size_t npsky = (size_t)pr.Nt * pr.Nf * pr.P;
    std::vector<float> ps(npsky);
    for (auto& v : ps) v = (float)(ud(rng) - 0.5);
    double *dbl, *dsd, *dfr; float *dps, *dvis, *dws;
    size_t nvis = (size_t)pr.Nbl * pr.Nt * pr.Nf * 2;
    CHK(hipMalloc(&dbl, bl.size() * 8)); CHK(hipMalloc(&dsd, sd.size() * 8)); CHK(hipMalloc(&dfr, fr.size() * 8));
    CHK(hipMalloc(&dps, npsky * 4));
    CHK(hipMalloc(&dvis, nvis * 4)); CHK(hipMalloc(&dws, nvis * 4 * 8));
    CHK(hipMemcpy(dbl, bl.data(), bl.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dsd, sd.data(), sd.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dfr, fr.data(), fr.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dps, ps.data(), npsky * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dvis, ps.data(), std::min(nvis, npsky) * 4, hipMemcpyHostToDevice));
    FringeArgs A{};
    A.blvecs = dbl; A.sdir = dsd; A.freqs = dfr; A.in = dps; A.out = dvis; A.ws = dws; A.bl_order = nullptr;
    A.bl_off = 0; A.bl_cnt = pr.Nbl; A.mp = 0; A.Nbl = pr.Nbl; A.Nt = pr.Nt; A.Nf = pr.Nf; A.Pstride = pr.P; A.Nmp = 1;
    A.st_f = pr.P; A.st_pp = (long long)pr.Nf * pr.P; A.st_mp = A.st_pp; A.st_t = A.st_mp;
    A.sign = 1.0; A.freq0_c = fr[0] / 2.99792458e8; A.dfreq_c = (fr[1] - fr[0]) / 2.99792458e8;

    {   // backward: in = gvis [1,Nbl,Nt,Nf] complex, out = gpsky
        FringeArgs B = A;
        B.in = dvis; B.out = dps; B.ws = dws;
        run_bwd<32, MODE_LIFT, 1, 1>("pix1 wps1", B, 1);
        run_bwd<32, MODE_LIFT, 1, 4>("pix1 wps4", B, 1);
        run_bwd<32, MODE_LIFT, 1, 3>("pix1 wps3", B, 1);
        run_bwd<32, MODE_LIFT, 2, 1>("pix2 wps1", B, 1);
        run_bwd<32, MODE_LIFT, 2, 2>("pix2 wps2", B, 1);
        run_bwd<16, MODE_LIFT, 2, 4>("ch16 pix2 wps4", B, 1);
        run_bwd<16, MODE_LIFT, 4, 2>("ch16 pix4 wps2", B, 1);
        run_bwd<64, MODE_LIFT, 1, 2>("ch64 pix1 wps2", B, 1);
        run_bwd<32, MODE_ROT, 1, 1>("rot pix1 wps1", B, 1);
    }
    run<32, MODE_LIFT, 0, 1>("base lift", A, 1, 256);
    run<32, MODE_LIFT, 0, 1>("base lift", A, 2, 256);
    run<32, MODE_LIFT, 0, 1>("base lift", A, 4, 256);
    run<32, MODE_LIFT, 0, 1>("base lift", A, 8, 256);
    run<32, MODE_LIFT, 0, 5>("lift wps5 (<=96 vgpr)", A, 3, 256);
    run<32, MODE_ROT, 0, 1>("base rot", A, 2, 256);
    run<32, MODE_LIFT, 4, 1>("abl: no accumulate", A, 2, 256);
    run<32, MODE_LIFT, 8, 1>("abl: no rotation", A, 2, 256);
    run<32, MODE_LIFT, 12, 1>("abl: no rot, no acc (setup only)", A, 2, 256);
    run<64, MODE_LIFT, 0, 1>("CH64", A, 2, 256);
    run<64, MODE_LIFT, 0, 1>("CH64", A, 4, 256);
    run<64, MODE_LIFT, 0, 1>("CH64", A, 8, 256);
    return 0;
}
