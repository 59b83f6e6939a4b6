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

// ---- multires hash grid: corner indices + offsets of one (position, level) (R2) ------------------------------------------
#define HASH_P1 2654435761u
#define HASH_P2 805459861u

struct HashCorners {
  uint32_t idx[8];
  float ox, oy, oz;
  uint32_t fx, fy, fz, eqx, eqy, eqz;  // floor coordinates and ceil == floor flags (cell identity)
};

__device__ __forceinline__ HashCorners hash_corners(float px, float py, float pz, float s, uint32_t mask,
                                                    uint32_t base) {
#pragma clang fp contract(off)
  // round the scaled coordinate BEFORE subtracting its floor (as the reference's torch ops do): a fused
  // fma(px, s, -floor) would use the unrounded product and shift the offset by up to half an ulp of ~2047.
  // (__fmul_rn is plain '*' in HIP, so contraction is switched off here and the products are made opaque.)
  float sx = px * s, sy = py * s, sz = pz * s;
  asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sz));
  float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  uint32_t xf = (uint32_t)(int)fx, yf = (uint32_t)(int)fy * HASH_P1, zf = (uint32_t)(int)fz * HASH_P2;
  // ceil = floor + 1 unless the coordinate is an integer: the hashed ceil products are the floor products + the prime (the same
  // bits as (uint32_t)(int)ceilf(.) * P mod 2^32, without two more quarter-rate v_mul_lo_u32)
  const bool ex = fx == sx, ey = fy == sy, ez = fz == sz;
  uint32_t xc = ex ? xf : xf + 1u, yc = ey ? yf : yf + HASH_P1, zc = ez ? zf : zf + HASH_P2;
  HashCorners h;
  h.fx = xf, h.fy = (uint32_t)(int)fy, h.fz = (uint32_t)(int)fz;
  h.eqx = ex, h.eqy = ey, h.eqz = ez;
  h.ox = sx - fx, h.oy = sy - fy, h.oz = sz - fz;
  // corner order of nerfstudio HashEncoding.pytorch_fwd: 0 ccc, 1 cfc, 2 ffc, 3 fcc, 4 ccf, 5 cff, 6 fff, 7 fcf
  h.idx[0] = ((xc ^ yc ^ zc) & mask) + base;
  h.idx[1] = ((xc ^ yf ^ zc) & mask) + base;
  h.idx[2] = ((xf ^ yf ^ zc) & mask) + base;
  h.idx[3] = ((xf ^ yc ^ zc) & mask) + base;
  h.idx[4] = ((xc ^ yc ^ zf) & mask) + base;
  h.idx[5] = ((xc ^ yf ^ zf) & mask) + base;
  h.idx[6] = ((xf ^ yf ^ zf) & mask) + base;
  h.idx[7] = ((xf ^ yc ^ zf) & mask) + base;
  return h;
}

// The 8 corner features of one (position, level).  Corners that differ only in x hash to slots idx and idx ^ (xf ^ xc): for an even
// floor coordinate that is the neighbouring slot of the same 16-byte pair, so ONE 16-byte load serves both -- on average 6 requests
// per (sample, level) instead of 8, and the gather is bound by the L2 request rate, not by bytes.  The other half of the lanes
// (odd floor coordinate) fetches its four ceil-x corners separately.  Same values as eight 8-byte loads.
struct HashGather {
  float4 q[4];   // the 16-byte slot pairs holding the floor-x corner of the four (y, z) combinations
  float2 cv[4];  // their ceil-x corners, fetched only when those live in another pair
  bool apart;
};
__device__ __forceinline__ void hash_gather8_issue(const float2* __restrict__ table, const HashCorners& h, HashGather& g) {
  const float4* __restrict__ t4 = reinterpret_cast<const float4*>(table);
  constexpr int FI[4] = {3, 2, 7, 6}, CI[4] = {0, 1, 4, 5};  // floor-x / ceil-x corner of the four (y, z) combinations
#pragma unroll
  for (int p = 0; p < 4; ++p) g.q[p] = t4[h.idx[FI[p]] >> 1];
  g.apart = (h.idx[CI[0]] >> 1) != (h.idx[FI[0]] >> 1);  // (a property of the x coordinate: the same for all four pairs)
  if (g.apart) {
#pragma unroll
    for (int p = 0; p < 4; ++p) g.cv[p] = table[h.idx[CI[p]]];
  }
}
__device__ __forceinline__ void hash_gather8_select(const HashCorners& h, const HashGather& g, float2 (&f)[8]) {
  constexpr int FI[4] = {3, 2, 7, 6}, CI[4] = {0, 1, 4, 5};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float2 lo = make_float2(g.q[p].x, g.q[p].y), hi = make_float2(g.q[p].z, g.q[p].w);
    f[FI[p]] = (h.idx[FI[p]] & 1u) ? hi : lo;
    f[CI[p]] = g.apart ? g.cv[p] : ((h.idx[CI[p]] & 1u) ? hi : lo);
  }
}
__device__ __forceinline__ void hash_gather8(const float2* __restrict__ table, const HashCorners& h, float2 (&f)[8]) {
  HashGather g;
  hash_gather8_issue(table, h, g);
  hash_gather8_select(h, g, f);
}

// trilinear blend of the 8 corner features (corner order above), one expression tree shared by the stand-alone gather kernel and
// the fused density kernel so that both produce the same bits
__device__ __forceinline__ float2 hash_trilerp(const float2 (&f)[8], float ox, float oy, float oz) {
#pragma clang fp contract(off)  // separate multiplies and adds, as the reference's torch ops -- and the same bits in every kernel
  const float rx = 1.0f - ox, ry = 1.0f - oy, rz = 1.0f - oz;
  float out[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float f0 = k ? f[0].y : f[0].x, f1 = k ? f[1].y : f[1].x, f2 = k ? f[2].y : f[2].x, f3 = k ? f[3].y : f[3].x;
    float f4 = k ? f[4].y : f[4].x, f5 = k ? f[5].y : f[5].x, f6 = k ? f[6].y : f[6].x, f7 = k ? f[7].y : f[7].x;
    float f03 = f0 * ox + f3 * rx, f12 = f1 * ox + f2 * rx, f56 = f5 * ox + f6 * rx, f47 = f4 * ox + f7 * rx;
    float f0312 = f03 * oy + f12 * ry, f4756 = f47 * oy + f56 * ry;
    out[k] = f0312 * oz + f4756 * rz;
  }
  return make_float2(out[0], out[1]);
}
