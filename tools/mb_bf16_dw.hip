// Layout check + timing for the transpose-free dW of the one-pass field backward (csrc/umhs_field_bwd.hip):
//   (1) an fp32 tile in "samples on lanes" layout (lane = (sample j = l&15, q = l>>4), reg r <-> feature 4q+r) is turned into
//       the swapped layout (lane = (feature c = l&15, q), reg r <-> sample 4q+r) by ONE v_mfma_f32_16x16x16_bf16 per bf16 piece
//       against an identity B operand that every lane builds from its own id -- no LDS, no cross-lane moves;
//   (2) dW[o][i] = sum_s Z[s][o] X[s][i] from two swapped tiles, as hi*hi + hi*lo + lo*hi on the bf16 MFMA, vs an fp64 sum;
//   (3) issue cost of v_mfma_f32_16x16x16_bf16 vs 16x16x32_bf16 vs 16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_bf16_dw.hip -o gpurun_out/mb_bf16_dw && gpurun_out/mb_bf16_dw
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// split 4 floats into bf16 hi / lo pieces (x ~ hi + lo, |x - hi - lo| <= 2^-17 |x|)
__device__ __forceinline__ void split4(const float (&x)[4], v4s& hi, v4s& lo) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned short h = bf16_rne(x[r]);
    hi[r] = (short)h;
    lo[r] = (short)bf16_rne(x[r] - bf16_f(h));
  }
}
__device__ __forceinline__ v4s pack4(const v4f& v) {  // values that ARE bf16 numbers
  v4s p;
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] = (short)(__float_as_uint(v[r]) >> 16);
  return p;
}
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

// N-layout tile (4 regs) -> swapped hi / lo tiles
__device__ __forceinline__ void transpose_tile(const float (&x)[4], const v4s& ident, v4f& shi, v4f& slo) {
  v4s hi, lo;
  split4(x, hi, lo);
  const v4f z = {0.f, 0.f, 0.f, 0.f};
  shi = MFMA16(hi, ident, z);
  slo = MFMA16(lo, ident, z);
}

__global__ void check_kernel(const float* __restrict__ X, const float* __restrict__ Z, float* __restrict__ XS, float* __restrict__ dW) {
  // X, Z: [16 samples][16 features] row-major.  One wave.
  const int lane = threadIdx.x, j = lane & 15, q = lane >> 4;
  float xn[4], zn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) xn[r] = X[j * 16 + 4 * q + r], zn[r] = Z[j * 16 + 4 * q + r];
  v4s ident;
#pragma unroll
  for (int u = 0; u < 4; ++u) ident[u] = ((j >> 2) == q && (j & 3) == u) ? (short)0x3F80 : (short)0;
  v4f xhi, xlo, zhi, zlo;
  transpose_tile(xn, ident, xhi, xlo);
  transpose_tile(zn, ident, zhi, zlo);
  // swapped: lane (feature c = j, q), reg r <-> sample 4q + r
#pragma unroll
  for (int r = 0; r < 4; ++r) XS[(4 * q + r) * 16 + j] = xhi[r] + xlo[r];
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  const v4s zh = pack4(zhi), zl = pack4(zlo), xh = pack4(xhi), xl = pack4(xlo);
  acc = MFMA16(zh, xh, acc);
  acc = MFMA16(zh, xl, acc);
  acc = MFMA16(zl, xh, acc);
  // D[o = 4q + r][i = j]
#pragma unroll
  for (int r = 0; r < 4; ++r) dW[(4 * q + r) * 16 + j] = acc[r];
}

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  v8bf pa, pb;
  for (int i = 0; i < 8; ++i) pa[i] = (__bf16)(a + i), pb[i] = (__bf16)(b + i);
  v4s qa, qb;
  for (int i = 0; i < 4; ++i) qa[i] = (short)(threadIdx.x + i), qb[i] = (short)(threadIdx.x * 3 + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(pa), "v"(pb));
      if (KIND == 2) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(qa), "v"(qb));
      // dependent chains: every MFMA accumulates onto the previous one's result (1 accumulator), or 2 alternating accumulators
      if (KIND == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(pa), "v"(pb));
      if (KIND == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(pa), "v"(pb));
      if (KIND == 5) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
      if (KIND == 6) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(qa), "v"(qb));
      if (KIND == 7) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(qa), "v"(qb));
      if (KIND == 8) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(qa), "v"(qb));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void rate(const char* name, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %6.1f cycles per MFMA (one wave per SIMD, 8 independent accumulators, 2.4 GHz assumed)\n", name,
         ms * 1e-3 * 2.4e9 / iters / 8);
}

int main() {
  std::vector<float> X(256), Z(256), XS(256), dW(256);
  srand(3);
  for (int i = 0; i < 256; ++i) X[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.0f, Z[i] = (rand() / (float)RAND_MAX - 0.3f) * 0.01f;
  float *dX, *dZ, *dXS, *ddW;
  hipMalloc(&dX, 1024), hipMalloc(&dZ, 1024), hipMalloc(&dXS, 1024), hipMalloc(&ddW, 1024);
  hipMemcpy(dX, X.data(), 1024, hipMemcpyHostToDevice), hipMemcpy(dZ, Z.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, dX, dZ, dXS, ddW);
  hipMemcpy(XS.data(), dXS, 1024, hipMemcpyDeviceToHost), hipMemcpy(dW.data(), ddW, 1024, hipMemcpyDeviceToHost);
  double e_t = 0, e_w = 0, m_w = 0;
  for (int s = 0; s < 16; ++s)
    for (int c = 0; c < 16; ++c) e_t = fmax(e_t, fabs((double)XS[s * 16 + c] - X[s * 16 + c]) / fabs((double)X[s * 16 + c]));
  for (int o = 0; o < 16; ++o)
    for (int i = 0; i < 16; ++i) {
      double ref = 0;
      for (int s = 0; s < 16; ++s) ref += (double)Z[s * 16 + o] * X[s * 16 + i];
      e_w = fmax(e_w, fabs(ref - dW[o * 16 + i]));
      m_w = fmax(m_w, fabs(ref));
    }
  printf("transpose via identity MFMA: max relative error of hi+lo vs x = %.3e (expect <= 2^-17 = 7.6e-6)\n", e_t);
  printf("dW via 3 bf16 products   : max |err| / max |ref| = %.3e\n", e_w / m_w);
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  rate<0>("v_mfma_f32_16x16x4_f32", out);
  rate<1>("v_mfma_f32_16x16x32_bf16", out);
  rate<2>("v_mfma_f32_16x16x16_bf16", out);
  rate<3>("16x16x32_bf16, 1 accumulator", out);
  rate<4>("16x16x32_bf16, 2 accumulators", out);
  rate<5>("16x16x4_f32, 1 accumulator", out);
  rate<6>("16x16x16_bf16, 1 accumulator", out);
  rate<7>("16x16x16_bf16, 1 acc (AGPR)", out);
  rate<8>("16x16x16_bf16, 4 acc (AGPR)", out);
  return (e_t < 1e-5 && e_w / m_w < 5e-5) ? 0 : 1;
}
