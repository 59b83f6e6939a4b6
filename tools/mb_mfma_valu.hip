// Micro-benchmark: do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and bf16 MFMAs (v_mfma_f32_16x16x32_bf16) overlap with plain
// VALU work on a CDNA4 SIMD?  Each case runs W waves per SIMD (blocks of 4*W waves on every CU) of: M MFMAs and V v_fma's
// per loop iteration, all independent.  Prints cycles per iteration per wave as seen by s_memtime/wall clock.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_mfma_valu.hip -o gpurun_out/mb_mfma_valu && gpurun_out/mb_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int KIND, int M, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  v8bf pa, pb;
  for (int i = 0; i < 8; ++i) pa[i] = (__bf16)(a + i), pb[i] = (__bf16)(b + i);
  typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
  v4bf qa, qb;
  for (int i = 0; i < 4; ++i) qa[i] = (__bf16)(a + i), qb[i] = (__bf16)(b + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      // inline asm with "v" constraints: accumulators stay in architectural VGPRs (left alone, hipcc parks them in AGPRs and
      // copies them back and forth every iteration, which is what one would then be measuring)
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(pa), "v"(pb));
      if (KIND == 3) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(qa), "v"(qb));
#pragma unroll
      for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * V + v) & 7]) : "v"(b), "v"(a));
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int M, int V>
void run(const char* name, int waves_per_simd, float* out) {
  const int iters = 20000, blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, M, V>), dim3(blocks), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, M, V>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: waves_per_simd waves, each iters iterations
  const double cyc = ms * 1e-3 * 2.4e9 / iters;  // cycles per iteration (all co-resident waves advance together)
  printf("%-34s waves/SIMD %d: %8.1f cycles per iteration (M=%d mfma + %d valu per wave)  -> per-SIMD %.1f cyc/mfma-slot\n", name,
         waves_per_simd, cyc, M, M * V, cyc / (M * waves_per_simd));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0, 8, 0>("fp32 mfma only", w, out);
    run<2, 8, 8>("valu only (64 fma)", w, out);
    run<0, 8, 4>("fp32 mfma + 4 fma each", w, out);
    run<0, 8, 8>("fp32 mfma + 8 fma each", w, out);
    run<3, 8, 0>("bf16 16x16x16 mfma only", w, out);
    run<1, 8, 0>("bf16 16x16x32 mfma only", w, out);
    run<1, 8, 2>("bf16 mfma + 2 fma each", w, out);
    run<1, 8, 4>("bf16 mfma + 4 fma each", w, out);
    run<1, 8, 8>("bf16 mfma + 8 fma each", w, out);
  }
  return 0;
}
