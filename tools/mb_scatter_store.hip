// Microbenchmark: the store pattern of the hash-grid backward's scatter pass, alone.  W workgroups (256 threads) per level, L levels; workgroup
// w writes a run of c 16-byte records at position w*c of each of NB bucket regions of its level (regions contiguous per level, as in
// csrc/umhs_kernels.hip), with consecutive lanes on consecutive records of a run.  Reports GB/s for run lengths c, for the XCD-contiguous
// vs round-robin workgroup mapping, plain vs nontemporal stores; a workgroup always writes 2048 records, in 2048 / c runs of c.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w tools/mb_scatter_store.hip -o /tmp/mbs && /tmp/mbs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int NT, int REMAP>
__global__ __launch_bounds__(256) void k(uint4* __restrict__ out, int W, int NB, int c, size_t level_stride) {
  int w = blockIdx.x;
  if (REMAP) {
    const int per = gridDim.x >> 3;
    w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  uint4* base = out + (size_t)blockIdx.y * level_stride;
  const int total = NB * c;  // records this workgroup writes
  for (int i = threadIdx.x; i < total; i += 256) {
    const int b = i / c, j = i - b * c;
    uint4* p = base + ((size_t)b * W + w) * c + j;
    const uint4 v = make_uint4(i, w, b, j);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    if (NT)
      __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(p));
    else
      *p = v;
  }
}

// unstaged: thread t's r-th record goes to a pseudo-random (bucket, place) of the workgroup's 64 x 32 slots (perm), i.e. the 64 lanes of a
// store instruction write 16 bytes each into ~40 different runs; the partial lines meet in L2
template <int REMAP>
__global__ __launch_bounds__(256) void kdirect(uint4* __restrict__ out, const unsigned short* __restrict__ perm, int W, size_t level_stride) {
  int w = blockIdx.x;
  if (REMAP) {
    const int per = gridDim.x >> 3;
    w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  uint4* base = out + (size_t)blockIdx.y * level_stride;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int slot = perm[threadIdx.x + 256 * r], b = slot >> 5, j = slot & 31;
    base[((size_t)b * W + w) * 32 + j] = make_uint4(r, w, b, j);
  }
}

// the write-out loop of hg_partition_kernel<true> as it is: records staged in LDS ordered by bucket, run lengths 24..40 (mean 32), per-record
// lookups of the run's LDS / global offsets, run starts at 640-byte strides (unaligned to lines); LDSPAD bytes of extra LDS set the occupancy
template <int LDSPAD>
__global__ __launch_bounds__(256) void kreal(uint4* __restrict__ out, int W, size_t level_stride) {
  __shared__ uint4 stage[2048 + 64 * 8];
  __shared__ uint32_t base[64], lbase[65];
  __shared__ char pad[LDSPAD > 0 ? LDSPAD : 1];
  const int per = gridDim.x >> 3, w = (blockIdx.x & 7) * per + (blockIdx.x >> 3), tid = threadIdx.x;
  if (LDSPAD > 0 && tid == 1000) pad[0] = 1;
  if (tid < 64) {
    const uint32_t c = 24 + ((tid * 13 + w * 5) % 17);
    uint32_t incl = c;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (tid >= d) incl += o;
    }
    lbase[tid] = incl - c, base[tid] = ((uint32_t)tid * W + w) * 40;
    if (tid == 63) lbase[64] = incl;
  }
  __syncthreads();
  const uint32_t total = lbase[64];
  for (uint32_t i = tid; i < total; i += 256) {  // (stand-in for the placement: every record knows its bucket)
    uint32_t b = 0;
    for (int st = 32; st >= 1; st >>= 1)
      if (b + st < 64 && lbase[b + st] <= i) b += st;
    stage[i] = make_uint4(i, w, 0, b << 17);
  }
  __syncthreads();
  uint4* o = out + (size_t)blockIdx.y * level_stride;
  for (uint32_t i = tid; i < total; i += 256) {
    const uint4 r = stage[i];
    const uint32_t b = (r.w >> 17) & 127u;
    o[base[b] + (i - lbase[b])] = r;
  }
}

// the same bytes as one contiguous stream per workgroup (what a store-bandwidth-bound kernel would get)
__global__ __launch_bounds__(256) void kseq(uint4* __restrict__ out, int W, int NB, int c, size_t level_stride) {
  uint4* base = out + (size_t)blockIdx.y * level_stride + (size_t)blockIdx.x * NB * c;
  for (int i = threadIdx.x; i < NB * c; i += 256) base[i] = make_uint4(i, 0, 0, 0);
}

int main() {
  const int L = 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  {
    const int W = 512;
    const size_t level_stride = (size_t)64 * W * 32;
    uint4* d;
    unsigned short hp[2048], *dp;
    for (int i = 0; i < 2048; ++i) hp[i] = (unsigned short)i;
    unsigned x = 1;
    for (int i = 2047; i > 0; --i) {
      x = x * 1664525u + 1013904223u;
      const int j = (x >> 8) % (i + 1);
      const unsigned short t = hp[i];
      hp[i] = hp[j], hp[j] = t;
    }
    hipMalloc(&d, L * level_stride * 16), hipMalloc(&dp, sizeof(hp));
    hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice);
    for (int remap = 0; remap < 2; ++remap) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (remap)
          hipLaunchKernelGGL(kdirect<1>, dim3(W, L), dim3(256), 0, 0, d, dp, W, level_stride);
        else
          hipLaunchKernelGGL(kdirect<0>, dim3(W, L), dim3(256), 0, 0, d, dp, W, level_stride);
        hipEventRecord(e1), hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("unstaged (every lane its own 16-byte record in a random run of 32), %s: %6.0f GB/s\n", remap ? "XCD-contiguous" : "round-robin", L * level_stride * 16 / best / 1e6);
    }
    hipFree(d), hipFree(dp);
  }
  {
    const int W = 512;
    const size_t level_stride = (size_t)64 * W * 40;
    uint4* d;
    hipMalloc(&d, L * level_stride * 16);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(kreal<0>, dim3(W, L), dim3(256), 0, 0, d, W, level_stride);
        if (mode == 1) hipLaunchKernelGGL(kreal<12000>, dim3(W, L), dim3(256), 0, 0, d, W, level_stride);   // 53 KB: 3 per CU
        if (mode == 2) hipLaunchKernelGGL(kreal<40000>, dim3(W, L), dim3(256), 0, 0, d, W, level_stride);   // 81 KB: 1 per CU
        hipEventRecord(e1), hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("real write-out loop (runs of 24..40 records from LDS, unaligned), LDS %s: %6.0f GB/s  (%.1f us for %.0f MB)\n",
             mode == 0 ? "41 KB (3 workgroups per CU)" : mode == 1 ? "53 KB (3 per CU)" : "81 KB (1 per CU)", L * 512.0 * 2048 * 16 / best / 1e6, best * 1e3,
             L * 512.0 * 2048 * 16 / 1e6);
    }
    hipFree(d);
  }
  for (int c : {1, 2, 4, 8, 16, 32, 64, 128}) {
    const int W = 512, NB = 2048 / c;  // every workgroup writes 2048 records (32 KiB), in 2048 / c runs: 512 x 32 KiB = 16.8 MB per level
    const size_t level_stride = (size_t)NB * W * c;
    uint4* d;
    hipMalloc(&d, L * level_stride * 16);
    const double bytes = (double)L * level_stride * 16;
    auto run = [&](int mode) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        dim3 g(W, L);
        if (mode == 0) hipLaunchKernelGGL((k<0, 0>), g, dim3(256), 0, 0, d, W, NB, c, level_stride);
        if (mode == 1) hipLaunchKernelGGL((k<0, 1>), g, dim3(256), 0, 0, d, W, NB, c, level_stride);
        if (mode == 2) hipLaunchKernelGGL((k<1, 1>), g, dim3(256), 0, 0, d, W, NB, c, level_stride);
        if (mode == 3) hipLaunchKernelGGL(kseq, g, dim3(256), 0, 0, d, W, NB, c, level_stride);
        hipEventRecord(e1), hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      return bytes / best / 1e6;
    };
    printf("run of %3d records (%5d B), %5d workgroups/level, %.0f MB: round-robin %6.0f GB/s | XCD-contiguous %6.0f | + nontemporal %6.0f | sequential %6.0f\n", c,
           c * 16, W, bytes / 1e6, run(0), run(1), run(2), run(3));
    hipFree(d);
  }
  return 0;
}
