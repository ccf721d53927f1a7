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

template <typename T> __device__ __forceinline__ T tfma(T a, T b, T c);
template <> __device__ __forceinline__ float tfma<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double tfma<double>(double a, double b, double c) { return fma(a, b, c); }

} // namespace rime
