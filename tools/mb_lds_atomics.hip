// Microbenchmark: throughput of LDS atomics (f32 add, u32 add, u64 add) and plain LDS RMW on gfx950.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/mb_lds_atomics.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ idx, int per_thread, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* f = reinterpret_cast<float*>(smem);
  uint32_t* u = reinterpret_cast<uint32_t*>(smem);
  unsigned long long* q = reinterpret_cast<unsigned long long*>(smem);
  for (int i = threadIdx.x; i < 16384; i += 256) u[i] = 0;
  __syncthreads();
  const uint32_t* p = idx + (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
  for (int j = 0; j < per_thread; ++j) {
    uint32_t s = p[(size_t)j * 256] & 8191;
    if (MODE == 0) atomicAdd(&f[s], 1.0f);
    if (MODE == 1) atomicAdd(&u[s], 1u);
    if (MODE == 2) atomicAdd(&q[s], 1ull);
    if (MODE == 3) f[s] += 1.0f;  // racy plain RMW: issue-rate reference only
    if (MODE == 4) atomicAdd(&f[(s & 63)], 1.0f);  // heavy same-address contention
    if (MODE == 5) atomicAdd(&u[(s & 63)], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[0];
}

int main() {
  const int blocks = 1024, per_thread = 256;
  size_t n = (size_t)blocks * 256 * per_thread;
  std::vector<uint32_t> h(n);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
  uint32_t* d; float* out;
  hipMalloc(&d, n * 4); hipMalloc(&out, blocks * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"ds_add_f32 random", "ds_add_u32 random", "ds_add_u64 random", "plain RMW f32", "ds_add_f32 64 addrs", "ds_add_u32 64 addrs"};
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      size_t lds = mode == 2 ? 8192 * 8 : 16384 * 4;
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), lds, 0, d, per_thread, out); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), lds, 0, d, per_thread, out); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), lds, 0, d, per_thread, out); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), lds, 0, d, per_thread, out); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), lds, 0, d, per_thread, out); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), lds, 0, d, per_thread, out); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    double ops = (double)n;
    printf("%-22s %8.3f ms  %7.2f G lane-ops/s  (%.2f cycles/lane-op/CU @2.4GHz)\n", names[mode], best, ops / best / 1e6,
           best * 1e-3 * 2.4e9 * 256 / ops);
  }
  return 0;
}
