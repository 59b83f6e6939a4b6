// Microbenchmark + exactness check: the residual of a bf16 split, r = x - float(bf16 piece), as ONE v_dot2_f32_bf16 on the packed pieces
// (D = A.lo * B.lo + A.hi * B.hi + C with B = {-1, 0} / {0, -1}) instead of unpack (shift / and) + subtract.
//   hipcc --offload-arch=gfx950 -O3 -w tools/mb_dot2_split.hip -o /tmp/mbd && /tmp/mbd
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf(float a, float b) {
  const v2f v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, v2bf));
}
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, a), __builtin_bit_cast(v2bf, b), c, false);
}
__device__ __forceinline__ void split_ref(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = cvt_pk_bf(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf(r0, r1);
  l = cvt_pk_bf(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void split_dot(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  // (the constants through an opaque move: hipcc 7.2 folds 0x0000bf80 into the inline constant "-1.0", which the instruction reads as the
  //  FLOAT -1.0 = 0xbf800000 -- the other half)
  uint32_t KLO, KHI;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(KLO));
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(KHI));
  h = cvt_pk_bf(x0, x1);
  const float r0 = dot2(h, KLO, x0), r1 = dot2(h, KHI, x1);
  m = cvt_pk_bf(r0, r1);
  l = cvt_pk_bf(dot2(m, KLO, r0), dot2(m, KHI, r1));
}
__global__ void check(const float* x, int n, uint32_t* out_ref, uint32_t* out_dot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  uint32_t h, m, l;
  split_ref(x[2 * i], x[2 * i + 1], h, m, l);
  out_ref[3 * i] = h, out_ref[3 * i + 1] = m, out_ref[3 * i + 2] = l;
  split_dot(x[2 * i], x[2 * i + 1], h, m, l);
  out_dot[3 * i] = h, out_dot[3 * i + 1] = m, out_dot[3 * i + 2] = l;
}
template <int MODE>
__global__ __launch_bounds__(64) void rate(float* io, long long* cyc) {
  float x[16];
  for (int k = 0; k < 16; ++k) x[k] = io[threadIdx.x * 16 + k];
  uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      uint32_t h, m, l;
      if (MODE == 0) split_ref(x[k], x[k + 1], h, m, l); else split_dot(x[k], x[k + 1], h, m, l);
      acc[k / 2] ^= h + m + l;
      x[k] += 1e-9f, x[k + 1] += 1e-9f;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  io[threadIdx.x] = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ acc[4] ^ acc[5] ^ acc[6] ^ acc[7]);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int n = 1 << 22;
  float* hx = new float[n];
  uint32_t s = 12345;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    uint32_t bits = s;
    if (i % 7 == 0) bits = (s & 0x807fffffu) | ((uint32_t)(100 + (s >> 9) % 60) << 23);  // moderate exponents
    if (((bits >> 23) & 0xff) == 0xff) bits &= 0xff7fffffu;  // no inf / nan
    memcpy(&hx[i], &bits, 4);
  }
  hx[0] = 0.f, hx[1] = -0.f, hx[2] = 1e-40f, hx[3] = -1e-39f, hx[4] = 1.17549435e-38f, hx[5] = 3.0e38f, hx[6] = 1.0f, hx[7] = 1.00390625f;
  float* dx; uint32_t *dr, *dd;
  hipMalloc(&dx, n * 4), hipMalloc(&dr, n / 2 * 12), hipMalloc(&dd, n / 2 * 12);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dr, dd);
  uint32_t* hr = new uint32_t[n / 2 * 3]; uint32_t* hd = new uint32_t[n / 2 * 3];
  hipMemcpy(hr, dr, n / 2 * 12, hipMemcpyDeviceToHost), hipMemcpy(hd, dd, n / 2 * 12, hipMemcpyDeviceToHost);
  long bad = 0, bad_big = 0;
  for (int i = 0; i < n / 2; ++i) {
    if (hr[3 * i] != hd[3 * i] || hr[3 * i + 1] != hd[3 * i + 1] || hr[3 * i + 2] != hd[3 * i + 2]) {
      ++bad;
      const float a = fabsf(hx[2 * i]), b = fabsf(hx[2 * i + 1]);
      const bool tiny = (a < 1e-30f && a != 0) || (b < 1e-30f && b != 0) || a > 1e38f || b > 1e38f;
      if (!tiny) ++bad_big;
      if (bad <= 12) printf("diff at %d: x = %a %a  ref %08x %08x %08x  dot %08x %08x %08x\n", i, hx[2 * i], hx[2 * i + 1], hr[3 * i], hr[3 * i + 1], hr[3 * i + 2], hd[3 * i], hd[3 * i + 1], hd[3 * i + 2]);
    }
  }
  printf("pairs %d, differing %ld (of which with both |x| in [1e-30, 1e38] or 0: %ld)\n", n / 2, bad, bad_big);
  long long* dc; hipMalloc(&dc, 8 * 1024);
  long long hc[4];
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, dx, dc); else hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, dx, dc);
      hipDeviceSynchronize();
    }
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.1f cycles per pair split (one wave alone, 8 independent pairs in flight, + 5 other VALU ops per pair)\n", mode ? "dot2" : "shift+sub", hc[0] / (256.0 * 8));
  }
  return 0;
}
