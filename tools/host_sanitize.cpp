// host_sanitize.cpp -- AddressSanitizer / UBSan sweep of the HOST-side code of librime_hip.so (development tool, CPU only):
// the workspace-size and split-planning arithmetic and the argument validation that every entry point runs before it
// launches anything.  Built by tools/host_sanitize.sh against a sanitizer build of the library; no GPU is needed (calls that
// would launch are made with arguments that are rejected first, or return a launch error without a device).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../include/rime_hip.h"

int main()
{
    std::mt19937 rng(1);
    auto U = [&](int lo, int hi) { return std::uniform_int_distribution<int>(lo, hi)(rng); };
    size_t acc = 0;
    long calls = 0;
    for (int it = 0; it < 200000; ++it) {
        const int Nbl = U(1, 140000), Nt = U(1, 70), Nf = U(1, 600), P = 64 * U(1, 7000), Nmp = U(1, 9);
        const int Npp = (int[]){1, 2, 4}[U(0, 2)], cplx = U(0, 1), dt = U(0, 1), R = U(1, 300), Nc = U(1, 9000), Npix = U(1, 200000);
        acc += rime_fringe_sum_workspace(dt, Nbl, Nt, Nf, P, Nmp, Npp, cplx && Npp != 2, 0);
        acc += rime_fringe_sum_workspace(dt, Nbl, Nt, Nf, P, Nmp, Npp, cplx && Npp != 2, 1);
        acc += rime_fringe_ant_workspace(Nbl, Nt, Nf, P);
        acc += rime_fringe_ant_bwd_workspace(Nbl, Nt, Nf);
        acc += rime_alm2pix_fwd_workspace(dt, R, Nc, Npix);
        acc += rime_alm2pix_bwd_workspace(dt, R, Nc, Npix);
        acc += rime_chisq_workspace();
        acc += rime_alm2pix_packed_bytes(Nc, Npix, 0) + rime_alm2pix_packed_bytes(Nc, Npix, 1);
        calls += 9;
    }
    // argument validation: null pointers, bad shapes, bad flags must come back as error codes, never touch memory
    std::vector<double> d(64, 0.0);
    std::vector<float> f(64, 0.f);
    std::vector<int> tab(128 * 128, -1);
    int off[2] = {0, 1};
    int bad = 0;
    bad += rime_fringe_sum_fwd(0, nullptr, d.data(), d.data(), f.data(), off, nullptr, 1, 1, 1, 64, 1, 1, 0, 1, 1, 1e8, 1e6, 10.0, nullptr, f.data(), nullptr, 0, nullptr) != RIME_EINVAL;
    bad += rime_fringe_sum_fwd(0, d.data(), d.data(), d.data(), f.data(), off, nullptr, 1, 1, 1, 63, 1, 1, 0, 1, 1, 1e8, 1e6, 10.0, nullptr, f.data(), nullptr, 0, nullptr) != RIME_EINVAL;
    bad += rime_fringe_sum_fwd(0, d.data(), d.data(), d.data(), f.data(), off, nullptr, 1, 1, 1, 64, 1, 3, 0, 1, 1, 1e8, 1e6, 10.0, nullptr, f.data(), nullptr, 0, nullptr) != RIME_EINVAL;
    bad += rime_fringe_sum_fwd(0, d.data(), d.data(), d.data(), f.data(), off, nullptr, 1, 1, 1, 64, 1, 2, 1, 1, 1, 1e8, 1e6, 10.0, nullptr, f.data(), nullptr, 0, nullptr) != RIME_EUNSUPPORTED;
    bad += rime_fringe_ant_fwd_block(d.data(), 129, 0, 0, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 0, f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    bad += rime_fringe_ant_fwd_block(d.data(), 96, 64, 0, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 0, f.data(), 1 << 20, nullptr) != RIME_EINVAL;      // (64, 32): unsupported shape
    bad += rime_fringe_ant_fwd_block(d.data(), 64, 0, 0, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 2, 1, 1, f.data(), 1 << 20, nullptr) != RIME_EUNSUPPORTED;  // complex pass on a diagonal block
    bad += rime_fringe_ant_fwd_block(d.data(), 64, 32, 0, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 1, f.data(), 1 << 20, nullptr) != RIME_EINVAL;       // complex needs st_p == 2
    bad += rime_fringe_ant_bwd_block(d.data(), 64, 32, 0, d.data(), d.data(), f.data(), tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 2, 1, 1, 0, f.data(), f.data(), 0, nullptr) != RIME_EWORKSPACE;
    // mirror mask (round 5): bits at or beyond ceil(Nrows / 16), or a negative mask -> RIME_EINVAL, before anything is launched
    bad += rime_fringe_ant_fwd_block(d.data(), 40, 0, 8, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 0, f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    bad += rime_fringe_ant_fwd_block(d.data(), 40, 0, -1, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 0, f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    bad += rime_fringe_ant_bwd_block(d.data(), 128, 0, 256, d.data(), d.data(), f.data(), tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 0, 0, f.data(), f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    // conjugate-pair blocks (round 5): more than 64 rows, a pixel stride other than 1 or 2, missing tables -> RIME_EINVAL; no workspace
    bad += rime_fringe_pair_fwd_block(d.data(), 65, nullptr, 0, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    bad += rime_fringe_pair_fwd_block(d.data(), 64, nullptr, 0, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 3, 1, f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    bad += rime_fringe_pair_fwd_block(d.data(), 64, nullptr, 0, d.data(), d.data(), f.data(), f.data(), nullptr, nullptr, tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    bad += rime_fringe_pair_fwd_block(d.data(), 64, nullptr, 0, d.data(), d.data(), f.data(), f.data(), nullptr, tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, f.data(), 4, nullptr) != RIME_EWORKSPACE;
    bad += rime_fringe_pair_bwd_block(d.data(), 0, nullptr, 0, d.data(), d.data(), f.data(), tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 0, f.data(), f.data(), 1 << 20, nullptr) != RIME_EINVAL;
    bad += rime_fringe_pair_bwd_block(d.data(), 64, nullptr, 0, d.data(), d.data(), f.data(), tab.data(), tab.data(), 1, 1, 1, 64, 64, 64, 1, 1, 0, f.data(), f.data(), 0, nullptr) != RIME_EWORKSPACE;
    bad += rime_eq2top(d.data(), d.data(), -1, d.data(), d.data(), 0.0, d.data(), d.data(), nullptr) != RIME_EINVAL;
    bad += rime_eq2top(d.data(), d.data(), 0, d.data(), d.data(), 0.0, d.data(), d.data(), nullptr) != RIME_OK;
    bad += rime_interp_gather_fwd(0, 0, nullptr, nullptr, nullptr, 1, 1, 1, 1, nullptr, 1, nullptr) == RIME_OK;
    bad += rime_alm2pix_fwd(0, nullptr, nullptr, 1.0, 1, 1, 1, nullptr, nullptr, 0, nullptr) != RIME_EINVAL;
    bad += rime_alm2pix_pack(f.data(), 0.0, 8, 8, 0, f.data(), nullptr) != RIME_EINVAL;                  // y_scale must be > 0
    bad += rime_alm2pix_pack(f.data(), 1.0, 8, 8, 2, f.data(), nullptr) != RIME_EINVAL;                  // direction 0 | 1
    bad += rime_alm2pix_fwd_packed(f.data(), f.data(), 1.0, 4, 8, 8, f.data(), nullptr, 0, nullptr) != RIME_EWORKSPACE;
    bad += rime_alm2pix_bwd_packed(f.data(), f.data(), 1.0, 4, 8, 8, f.data(), nullptr, 0, nullptr) != RIME_EWORKSPACE;
    bad += rime_jones_apply_fwd(0, 0, f.data(), f.data(), f.data(), 10, 3, f.data(), nullptr) != RIME_EINVAL;        // Ns must divide N
    bad += rime_jones_apply_bwd(0, 0, f.data(), f.data(), f.data(), f.data(), 8, 4, nullptr, f.data(), f.data(), nullptr) != RIME_EINVAL;
    bad += rime_fringe_row_scale(f.data(), 1, 1, 1, 0, 0, 0, 0, 0, 8, f.data(), nullptr, nullptr) != RIME_EINVAL;
    bad += rime_comm_init(nullptr, 1, 0, nullptr) != RIME_EINVAL;
    bad += rime_comm_allgather_vis(nullptr, 0, nullptr, nullptr, 1, nullptr) != RIME_EINVAL;
    std::printf("host sanitizer sweep: %ld planning calls (checksum %zu), %d unexpected return codes, version %s\n",
                calls, acc % 1000003, bad, rime_version());
    return bad != 0;
}
