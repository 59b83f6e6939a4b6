// Shared helpers for the gfx950 kernels of libumhs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "umhs_hip.h"

#define UMHS_WAVE 64

#define UMHS_CHECK_LAUNCH()                        \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return UMHS_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t umhs_s(umhs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- wave64 primitives ------------------------------------------------------------------------
__device__ __forceinline__ float wave_inclusive_scan(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// inclusive scan from the high lanes down (suffix sum)
__device__ __forceinline__ float wave_inclusive_scan_rev(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_down(v, d, 64);
    if (lane + d < 64) v += o;
  }
  return v;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
