// chisq.hip -- likelihood epilogue of the RIME path (gfx950):
//     chi^2 = sum_i icov[i] |pred[i] - data[i]|^2          (optim.apply_icov, cov_axis=None,
//                                                            optim.py:1889-1894, as used by
//                                                            LogProb.forward_chisq :1019-1027)
//     gpred[i] = 2 g icov[i] (pred[i] - data[i])           (its autograd backward)
// One pass over the visibility tensor instead of five elementwise / reduction kernels; HBM-bound
// (pred + data + icov read once).  Deterministic: per-block partial sums in double, summed in block
// order by a second kernel.  data == NULL -> 0, icov == NULL -> 1.
#include <hip/hip_runtime.h>
#include "rime_common.h"

namespace rime {

constexpr int CHI_BLOCKS = 2048;

template <typename T>
__global__ void __launch_bounds__(256)
chisq_partial_kernel(const T* __restrict__ pred, const T* __restrict__ data, const T* __restrict__ icov,
                     size_t N, double* __restrict__ partial)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        T re = pred[2 * i], im = pred[2 * i + 1];
        if (data) { re -= data[2 * i]; im -= data[2 * i + 1]; }
        T v = re * re + im * im;
        if (icov) v *= icov[i];
        acc += (double)v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ void __launch_bounds__(256)
chisq_final_kernel(const double* __restrict__ partial, int nb, T* __restrict__ out)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];      // fixed assignment: deterministic
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (T)red[0];
}

template <typename T>
__global__ void __launch_bounds__(256)
chisq_bwd_kernel(const T* __restrict__ pred, const T* __restrict__ data, const T* __restrict__ icov,
                 const T* __restrict__ g, size_t N, T* __restrict__ gpred)
{
    const T two_g = T(2) * g[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        T re = pred[2 * i], im = pred[2 * i + 1];
        if (data) { re -= data[2 * i]; im -= data[2 * i + 1]; }
        const T w = icov ? two_g * icov[i] : two_g;
        gpred[2 * i] = w * re;
        gpred[2 * i + 1] = w * im;
    }
}

} // namespace rime

using namespace rime;

extern "C" size_t rime_chisq_workspace(void) { return CHI_BLOCKS * sizeof(double); }

extern "C" int rime_chisq_fwd(int dtype, const void* pred, const void* data, const void* icov, size_t N,
                              void* out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pred || !out || N == 0) return RIME_EINVAL;
    if (!workspace || workspace_bytes < rime_chisq_workspace()) return RIME_EWORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = (int)std::min<size_t>((N + 255) / 256, CHI_BLOCKS);
    double* part = (double*)workspace;
    if (dtype == RIME_F32) {
        hipLaunchKernelGGL((chisq_partial_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)pred,
                           (const float*)data, (const float*)icov, N, part);
        hipLaunchKernelGGL((chisq_final_kernel<float>), dim3(1), dim3(256), 0, st, part, nb, (float*)out);
    } else if (dtype == RIME_F64) {
        hipLaunchKernelGGL((chisq_partial_kernel<double>), dim3(nb), dim3(256), 0, st, (const double*)pred,
                           (const double*)data, (const double*)icov, N, part);
        hipLaunchKernelGGL((chisq_final_kernel<double>), dim3(1), dim3(256), 0, st, part, nb, (double*)out);
    } else return RIME_EINVAL;
    return check_launch();
}

extern "C" int rime_chisq_bwd(int dtype, const void* pred, const void* data, const void* icov, const void* g,
                              size_t N, void* gpred, void* stream)
{
    if (!pred || !g || !gpred || N == 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = (int)std::min<size_t>((N + 255) / 256, 8192);
    if (dtype == RIME_F32)
        hipLaunchKernelGGL((chisq_bwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)pred, (const float*)data,
                           (const float*)icov, (const float*)g, N, (float*)gpred);
    else if (dtype == RIME_F64)
        hipLaunchKernelGGL((chisq_bwd_kernel<double>), dim3(nb), dim3(256), 0, st, (const double*)pred,
                           (const double*)data, (const double*)icov, (const double*)g, N, (double*)gpred);
    else return RIME_EINVAL;
    return check_launch();
}
