// Microbenchmark 2: the LDS atomic unit's own rate on gfx950 -- indices preloaded into registers, atomics issued back to back with no
// global memory inside the loop (tools/mb_lds_atomics.hip had a dependent global load per atomic and measured the loop, not the unit).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/mb_lds_atomics2.hip -o /tmp/mb2 && /tmp/mb2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int NIDX = 16;

template <int MODE>
__global__ void k(const uint32_t* __restrict__ idx, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* f = reinterpret_cast<float*>(smem);
  double* d = reinterpret_cast<double*>(smem);
  uint32_t* u = reinterpret_cast<uint32_t*>(smem);
  unsigned long long* q = reinterpret_cast<unsigned long long*>(smem);
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) u[i] = 0;  // 128 KiB
  uint32_t s[NIDX];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NIDX; ++j) s[j] = idx[((size_t)blockIdx.x * NIDX + j) * blockDim.x + threadIdx.x] & 8191;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NIDX; ++j) {
      const uint32_t r = s[j], cf = (uint32_t)(lane + 64 * ((j + it) & 127)) & 8191;  // cf: lane-consecutive, conflict-free
      if (MODE == 0) atomicAdd(&q[2 * r], 1ull), atomicAdd(&q[2 * r + 1], 3ull);        // the reduce pass: 2 x u64 per record, random slot
      if (MODE == 1) atomicAdd(&q[2 * cf], 1ull), atomicAdd(&q[2 * cf + 1], 3ull);      // same, consecutive slots
      if (MODE == 2) atomicAdd(&u[r], 1u), atomicAdd(&u[r + 8192], 3u);                 // 2 x u32 random
      if (MODE == 3) atomicAdd(&u[cf], 1u), atomicAdd(&u[cf + 8192], 3u);               // 2 x u32 consecutive
      if (MODE == 4) atomicAdd(&f[r], 1.0f), atomicAdd(&f[r + 8192], 3.0f);             // 2 x f32 random
      if (MODE == 5) atomicAdd(&d[2 * r], 1.0), atomicAdd(&d[2 * r + 1], 3.0);          // 2 x f64 random
      if (MODE == 6) atomicAdd(&q[r], 1ull), atomicAdd(&q[r + 8192], 3ull);             // 2 x u64 random, components in separate planes
      if (MODE == 7) {                                                                  // x-pair: slots r and r^1, both components
        atomicAdd(&q[2 * r], 1ull), atomicAdd(&q[2 * r + 1], 3ull), atomicAdd(&q[2 * (r ^ 1)], 1ull), atomicAdd(&q[2 * (r ^ 1) + 1], 3ull);
      }
      if (MODE == 8) atomicAdd(&q[2 * r], 1ull);                                        // 1 x u64 random
      if (MODE == 9) atomicAdd(&u[r], 1u);                                              // 1 x u32 random
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[1];
}

int main() {
  const int blocks = 256, iters = 64;
  const char* names[] = {"2 x u64 random (reduce pass)", "2 x u64 consecutive", "2 x u32 random", "2 x u32 consecutive", "2 x f32 random",
                         "2 x f64 random", "2 x u64 random, planes", "4 x u64 x-pair", "1 x u64 random", "1 x u32 random"};
  const int per_rec[] = {2, 2, 2, 2, 2, 2, 2, 4, 1, 1};
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int threads : {256, 512, 1024}) {
    size_t n = (size_t)blocks * NIDX * threads;
    std::vector<uint32_t> h(n);
    uint32_t x = 12345;
    for (auto& v : h) x = x * 1664525u + 1013904223u, v = x >> 8;
    uint32_t* di;
    float* out;
    hipMalloc(&di, n * 4), hipMalloc(&out, blocks * 4);
    hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice);
    printf("threads per workgroup %d (one 128 KiB workgroup per CU, %d workgroups)\n", threads, blocks);
    for (int mode = 0; mode < 10; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        const size_t lds = 131072;
#define L(M)                                                                                              \
  case M:                                                                                                 \
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(threads), lds, 0, di, iters, out);                        \
    break;
        switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) }
        hipEventRecord(e1), hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double recs = (double)blocks * threads * NIDX * iters, winstr = recs * per_rec[mode] / 64.0;
      printf("  %-32s %8.3f ms  %7.1f ns per wave-instruction per CU  (%.2f ns per record per CU)\n", names[mode], best, best * 1e6 / (winstr / blocks),
             best * 1e6 / (recs / blocks));
    }
    hipFree(di), hipFree(out);
  }
  return 0;
}
