// Microbenchmark: issue rate of v_pk_add_f32 against two v_sub_f32 (one wave per SIMD, independent chains).
//   hipcc --offload-arch=gfx950 -O3 -w tools/mb_pk_add.hip -o /tmp/mbp && /tmp/mbp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float* io, long long* cyc) {
  v2f x[8], h[8];
  for (int i = 0; i < 8; ++i) x[i] = v2f{io[threadIdx.x + 64 * i], io[threadIdx.x + 64 * i + 1]}, h[i] = v2f{1e-3f * i, 2e-3f * i};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {
        asm volatile("v_sub_f32 %0, %1, %2\n v_sub_f32 %3, %4, %5" : "=v"(x[i].x), "=v"(x[i].y) : "v"(x[i].x), "v"(h[i].x), "v"(x[i].y), "v"(h[i].y));
      } else {
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x[i]) : "v"(x[i]), "v"(h[i]));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  io[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* d; long long* c; long long h;
  hipMalloc(&d, 4096 * 4), hipMalloc(&c, 8);
  hipMemset(d, 0, 4096 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, c); else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, c);
      hipDeviceSynchronize();
    }
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles (s_memtime units) per pair of subtractions (8 independent chains, one wave)\n", mode ? "v_pk_add_f32" : "2 x v_sub_f32", h / (512.0 * 8));
  }
  return 0;
}
