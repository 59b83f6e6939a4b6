// Micro-benchmark: the three-piece bf16 split of a 16x16 fp32 tile (x = h + m + l, the B operand of the next layer's bf16x3 products)
// with the residuals taken (A) on the VALU -- unpack + subtract, 11 instructions per value pair, what the zipped backward did up to
// round 3 -- or (B) on the MFMA pipe: r = x - h as v_mfma_f32_16x16x16_bf16(-I, h, x) (the packed pieces of a lane ARE its B-operand
// rows, and the accumulator layout is the C layout: D = -I h + x), 3 cvt_pk per pair + 2 MFMAs per four values.  F independent
// "filler" MFMAs per block of 16 values stand for the transposes / dW products the schedule zips in.  One wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/mb_split_mfma.hip -o gpurun_out/mb_split_mfma && gpurun_out/mb_split_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ uint32_t cvt_pk_bf(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int MODE, int F>
__global__ __launch_bounds__(256, 1) void k(float* out, uint32_t* pieces, int iters) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  // A operand -I: lane (i = lane & 15, q): elements k = 4q + r
  v4s nident;
  for (int r = 0; r < 4; ++r) nident[r] = (4 * q + r == j) ? (short)0xBF80 : (short)0;  // bf16(-1.0)
  v4f x[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 4; ++r) x[t][r] = 0.37f + 0.001f * threadIdx.x + 0.11f * t + 0.013f * r + blockIdx.x * 1e-4f;
  v4f fill[8];
  for (int i = 0; i < 8; ++i) fill[i] = v4f{0, 0, 0, 0};
  v4s fa = nident, fb = nident;
  uint32_t H[8], M[8], L[8];
  for (int i = 0; i < 8; ++i) H[i] = M[i] = L[i] = 0;
  constexpr int STEPS = MODE == 1 ? 32 : 40;  // pinned steps per block: F fillers spread over them
  int fdone = 0;
  auto filler = [&](int step) __attribute__((always_inline)) {
    // fillers due after this step: floor((step + 1) * F / STEPS)
    const int due = (step + 1) * F / STEPS;
    for (; fdone < due; ++fdone) fill[fdone & 7] = MFMA_BF(fa, fb, fill[fdone & 7]);
    SB();
  };
  for (int it = 0; it < iters; ++it) {
    fdone = 0;
    SB();
    if (MODE == 0) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {  // pair p = values (x[p>>1][2(p&1)], x[p>>1][2(p&1)+1])
        const float x0 = x[p >> 1][2 * (p & 1)], x1 = x[p >> 1][2 * (p & 1) + 1];
        H[p] = cvt_pk_bf(x0, x1);
        filler(5 * p);
        const float h0 = __uint_as_float(H[p] << 16), h1 = __uint_as_float(H[p] & 0xffff0000u);
        filler(5 * p + 1);
        const float r0 = x0 - h0, r1 = x1 - h1;
        M[p] = cvt_pk_bf(r0, r1);
        filler(5 * p + 2);
        const float m0 = __uint_as_float(M[p] << 16), m1 = __uint_as_float(M[p] & 0xffff0000u);
        filler(5 * p + 3);
        L[p] = cvt_pk_bf(r0 - m0, r1 - m1);
        filler(5 * p + 4);
      }
    } else if (MODE == 2) {
      // the VALU form with the steps of the eight pairs interleaved (phase-major): consecutive pinned steps are independent
      float r0[8], r1[8], h0[8], h1[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        H[p] = cvt_pk_bf(x[p >> 1][2 * (p & 1)], x[p >> 1][2 * (p & 1) + 1]);
        filler(p);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        h0[p] = __uint_as_float(H[p] << 16), h1[p] = __uint_as_float(H[p] & 0xffff0000u);
        filler(8 + p);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        r0[p] = x[p >> 1][2 * (p & 1)] - h0[p], r1[p] = x[p >> 1][2 * (p & 1) + 1] - h1[p];
        M[p] = cvt_pk_bf(r0[p], r1[p]);
        filler(16 + p);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        h0[p] = __uint_as_float(M[p] << 16), h1[p] = __uint_as_float(M[p] & 0xffff0000u);
        filler(24 + p);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        L[p] = cvt_pk_bf(r0[p] - h0[p], r1[p] - h1[p]);
        filler(32 + p);
      }
    } else {
      // two quads at a time, interleaved: A0 B0 C0 A1 B1 C1 | D0 E0 F0 D1 E1 F1 | G0 H0 G1 H1   (16 steps per two quads)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        v4f r[2], r2[2];
        int st = 16 * g;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * g + u;
          H[2 * t] = cvt_pk_bf(x[t][0], x[t][1]);
          filler(st++);
          H[2 * t + 1] = cvt_pk_bf(x[t][2], x[t][3]);
          filler(st++);
          r[u] = MFMA_BF(nident, __builtin_bit_cast(v4s, make_uint2(H[2 * t], H[2 * t + 1])), x[t]);
          filler(st++);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * g + u;
          M[2 * t] = cvt_pk_bf(r[u][0], r[u][1]);
          filler(st++);
          M[2 * t + 1] = cvt_pk_bf(r[u][2], r[u][3]);
          filler(st++);
          r2[u] = MFMA_BF(nident, __builtin_bit_cast(v4s, make_uint2(M[2 * t], M[2 * t + 1])), r[u]);
          filler(st++);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * g + u;
          L[2 * t] = cvt_pk_bf(r2[u][0], r2[u][1]);
          filler(st++);
          L[2 * t + 1] = cvt_pk_bf(r2[u][2], r2[u][3]);
          filler(st++);
        }
      }
    }
    // opaque to the optimiser: nothing hoists out of the loop, no instruction spent
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(x[t]));
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(H[i]), "v"(M[i]), "v"(L[i]));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += fill[i][0] + fill[i][3];
  for (int t = 0; t < 4; ++t) s += x[t][0] + x[t][1] + x[t][2] + x[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (pieces && blockIdx.x == 0)
    for (int i = 0; i < 8; ++i) {
      pieces[(0 * 8 + i) * 256 + threadIdx.x] = H[i];
      pieces[(1 * 8 + i) * 256 + threadIdx.x] = M[i];
      pieces[(2 * 8 + i) * 256 + threadIdx.x] = L[i];
    }
}

template <int MODE, int F>
double run(const char* name, float* out, uint32_t* pieces) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, F>), dim3(256), dim3(256), 0, 0, out, pieces, 1);  // (one iteration: the pieces to compare)
  hipLaunchKernelGGL((k<MODE, F>), dim3(256), dim3(256), 0, 0, out, (uint32_t*)nullptr, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, F>), dim3(256), dim3(256), 0, 0, out, (uint32_t*)nullptr, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us_per_it = ms * 1e3 / iters;
  printf("%-44s %7.1f ns per block of 16 values  (~%5.0f cycles at 2.1 GHz; %d filler MFMAs = %4d cycles of MFMA pipe)\n", name, us_per_it * 1e3,
         us_per_it * 2100.0, F, 16 * F + (MODE ? 16 * 8 : 0));
  return us_per_it;
}

int main() {
  float* out;
  uint32_t *pa, *pb;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&pa, 24 * 256 * 4), hipMalloc(&pb, 24 * 256 * 4);
  run<0, 0>("VALU residuals, no fillers", out, pa);
  run<1, 0>("MFMA residuals, no fillers", out, pb);
  std::vector<uint32_t> a(24 * 256), b(24 * 256);
  hipMemcpy(a.data(), pa, a.size() * 4, hipMemcpyDeviceToHost), hipMemcpy(b.data(), pb, b.size() * 4, hipMemcpyDeviceToHost);
  int diff = 0;
  for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
  printf("pieces that differ between the two forms: %d of %zu\n", diff, a.size());
  run<2, 0>("VALU residuals phase-major, no fillers", out, pb);
  hipMemcpy(b.data(), pb, b.size() * 4, hipMemcpyDeviceToHost);
  diff = 0;
  for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
  printf("pieces that differ (phase-major vs pair-major): %d\n", diff);
  run<2, 8>("VALU residuals phase-major, 8 fillers", out, nullptr);
  run<2, 16>("VALU residuals phase-major, 16 fillers", out, nullptr);
  run<2, 24>("VALU residuals phase-major, 24 fillers", out, nullptr);
  run<2, 40>("VALU residuals phase-major, 40 fillers", out, nullptr);
  run<0, 8>("VALU residuals, 8 fillers", out, nullptr);
  run<1, 8>("MFMA residuals, 8 fillers", out, nullptr);
  run<0, 16>("VALU residuals, 16 fillers", out, nullptr);
  run<1, 16>("MFMA residuals, 16 fillers", out, nullptr);
  run<0, 24>("VALU residuals, 24 fillers", out, nullptr);
  run<1, 24>("MFMA residuals, 24 fillers", out, nullptr);
  run<0, 40>("VALU residuals, 40 fillers", out, nullptr);
  run<1, 40>("MFMA residuals, 40 fillers", out, nullptr);
  run<1, 32>("MFMA residuals, 32 fillers", out, nullptr);
  return 0;
}
