// rime_common.h -- shared host-side helpers of librime_hip.so (not part of the public ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include "../../include/rime_hip.h"

namespace rime {

extern char g_last_error[256];

// Call after launching: maps a failed launch to RIME_ELAUNCH and records the HIP message.
inline int check_launch()
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return RIME_OK;
    std::snprintf(g_last_error, sizeof(g_last_error), "%s", hipGetErrorString(e));
    return RIME_ELAUNCH;
}

// Reads of matrix-core results by the vector ALU.  hipcc (ROCm 7.2) places the wait states of the ISA hazard table
// between an MFMA and the first VALU read of its destination.  Observed on gfx950 (round 2, complex-psky backward of
// the diagonal fringe blocks at 393 216 pixels): when that first reader was a PACKED f32 op (v_pk_fma_f32 with op_sel,
// the compiler's fusion of two accumulation chains) sitting at exactly that distance, its low half sporadically saw a
// stale accumulator (wrong imaginary-plane gradients in a few hundred of 12 288 pixel tiles, different from run to run;
// the high half -- the real plane -- was always right).  Epilogues that read accumulators therefore start with 16 more
// wait states, fenced so that the scheduler keeps them between the last MFMA and the reads.
#if defined(RIME_NO_SETTLE)          /* lab: measure what the margin costs */
#define RIME_MFMA_SETTLE() do { } while (0)
#else
#define RIME_MFMA_SETTLE()                                   \
    do {                                                     \
        __builtin_amdgcn_sched_barrier(0);                   \
        asm volatile("s_nop 15" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
#endif

template <typename T> __device__ __forceinline__ T tfma(T a, T b, T c);
template <> __device__ __forceinline__ float tfma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double tfma<double>(double a, double b, double c) { return fma(a, b, c); }

} // namespace rime
