// occupancy-grid ray marcher and visibility pruning for gfx950.
// Replaces nerfacc==0.5.2 OccGridEstimator.sampling (CUDA-only traverse_grids + render_visibility_from_density) reached
// through nerfstudio's VolumetricSampler at umhs_model.py:201-209,229-237.  nerfacc's source is not available offline: the
// traversal below restates its published behaviour (see oracle/torch_ref.py march_ray_ref, which this kernel reproduces
// bit for bit: float32, fixed operation order, fp contraction off).  One thread per ray; two passes (count, then write at
// the offsets of an exclusive scan) because the number of samples per ray is data dependent.
#include "umhs_common.h"

struct MarchArgs {
  const float* o;
  const float* d;
  int64_t n_rays;
  const uint8_t* bin;  // [levels][res][res][res], x-major like nerfacc's binaries
  float cx, cy, cz, hx, hy, hz;  // roi centre and half extent (level 0)
  int levels, res;
  float near, far, step, cone;
  const float *nears, *fars;   // optional per-ray planes [R] (override near / far)
  const float* jitter;         // optional [R]: the near plane of ray r is moved out by jitter[r] * jitter_step (stratified sampling)
  float jitter_step;
  const int64_t* packed_info;  // write pass: [R,2] (start, count)
  int64_t* counts;             // count pass
  float *t_starts, *t_ends;
  int64_t* ray_indices;
  int cap;  // single-pass mode (count kernel with scratch): the first `cap` samples of ray r go to t_starts/t_ends[r * cap + i]
};

#ifndef MARCH_RPW
#define MARCH_RPW 16
#endif
template <bool WRITE>
__global__ __launch_bounds__(64, 6) void march_kernel(MarchArgs a) {  // 6 waves/SIMD = at most 80 VGPRs (see the walk loop below)
#pragma clang fp contract(off)
  // MARCH_RPW rays per wave: a batch has a few thousand rays, the chip 1024 SIMDs -- a wave of 64 rays runs as long as its slowest
  // ray on one SIMD while 900 others idle; fewer rays per wave = less divergence, more SIMDs
  if (threadIdx.x >= MARCH_RPW) return;
  const int64_t r = (int64_t)blockIdx.x * MARCH_RPW + threadIdx.x;
  if (r >= a.n_rays) return;
  const float BIG = 1e30f;
  const float o[3] = {a.o[3 * r], a.o[3 * r + 1], a.o[3 * r + 2]};
  const float d[3] = {a.d[3 * r], a.d[3 * r + 1], a.d[3 * r + 2]};
  const float c[3] = {a.cx, a.cy, a.cz}, h[3] = {a.hx, a.hy, a.hz};
  float inv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) inv[k] = d[k] != 0.0f ? 1.0f / d[k] : (signbit(d[k]) ? -BIG : BIG);
  // per-axis constants of level 0; level l scales them by 2^l exactly (power-of-two factors commute with rounding), so every
  // voxel step is multiplications only: a float divide is a ~10-instruction dependent chain, and this kernel is one long
  // dependency chain per ray (64-256 waves in all: nothing else to overlap it with)
  float ih[3], vs0[3], ivs0[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ih[k] = 1.0f / h[k];
    vs0[k] = (h[k] * 2.0f) / (float)a.res;
    ivs0[k] = 1.0f / vs0[k];
  }
  const float top = (float)(1 << (a.levels - 1));
  float tn = -INFINITY, tf = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float ho = h[k] * top;
    const float t0 = (c[k] - ho - o[k]) * inv[k], t1 = (c[k] + ho - o[k]) * inv[k];
    tn = fmaxf(tn, fminf(t0, t1)), tf = fminf(tf, fmaxf(t0, t1));
  }
  float near_r = a.nears ? a.nears[r] : a.near;
  if (a.jitter) near_r = near_r + a.jitter[r] * a.jitter_step;  // = torch's nears + rand * step (contraction is off)
  float t = fmaxf(tn, near_r);
  const float t_end = fminf(tf, a.fars ? a.fars[r] : a.far);
  int64_t cnt = 0;
  int64_t w = WRITE ? a.packed_info[2 * r] : 0;
  int iters = 0;
  if (t < t_end) {
    bool continuous = false, left_grid = false;
    float t_last = t;
    constexpr int KB = 8;
    float tt = t;
    int guard = 0;
    // geometry of the next KB voxels from tt on: entry / clipped exit parameters and the cell index of each (slots past the end
    // of the walk keep cell 0, so that every occupancy fetch is unconditional: a load under "k < nb" compiles to a branch +
    // s_waitcnt vmcnt(0) each, KB serial L2 round trips -- exactly what the batch exists to avoid)
    auto walk = [&](float (&vt0)[KB], float (&vtc)[KB], uint32_t (&vcell)[KB]) -> int {
      int nb = 0;
#pragma unroll
      for (int k = 0; k < KB; ++k) vcell[k] = 0;
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        if (tt < t_end && guard < 100000 && !left_grid) {
          ++guard, ++iters;
          const float tm = tt + 1e-5f * fmaxf(1.0f, fabsf(tt));
          float p[3], m = 0.0f;
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            p[q] = o[q] + d[q] * tm;
            m = fmaxf(m, fabsf(p[q] - c[q]) * ih[q]);
          }
          if (!(m < top)) {
            left_grid = true;
          } else {
            int lvl = 0;
            if (!(m < 1.0f)) lvl = (int)((__float_as_uint(m) >> 23) & 0xffu) - 126;  // frexp exponent: m in [2^(e-1), 2^e) -> level e
            lvl = min(max(lvl, 0), a.levels - 1);
            const float sc = __uint_as_float((uint32_t)(127 + lvl) << 23), isc = __uint_as_float((uint32_t)(127 - lvl) << 23);
            int idx[3];
            float t_exit = BIG;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const float hl = h[q] * sc, vmin = c[q] - hl, vs = vs0[q] * sc, ivs = ivs0[q] * isc;
              int i = (int)floorf((p[q] - vmin) * ivs);
              i = min(max(i, 0), a.res - 1);
              idx[q] = i;
              const float lo = vmin + (float)i * vs, hi = vmin + (float)(i + 1) * vs;
              const float tx = ((d[q] >= 0.0f ? hi : lo) - o[q]) * inv[q];
              if (d[q] != 0.0f) t_exit = fminf(t_exit, tx);
            }
            if (!(t_exit > tt)) t_exit = nextafterf(tt, BIG);
            const float t_clip = fminf(t_exit, t_end);
            vt0[k] = tt, vtc[k] = t_clip;
            vcell[k] = (uint32_t)((((size_t)lvl * a.res + idx[0]) * a.res + idx[1]) * a.res + idx[2]);
            tt = t_clip;
            nb = k + 1;
          }
        }
      }
      return nb;
    };
    // The voxel sequence a ray crosses is pure geometry: it does not depend on what the occupancy grid says.  So the walk runs in
    // batches of KB voxels: geometry for KB steps, then the KB occupancy bytes fetched together (one L2 / MALL round trip instead
    // of KB), then the sample emission.  (Overlapping the fetches with the next batch's geometry as well bought nothing -- the
    // geometry is what the time goes to -- and cost 26 VGPRs: at 100 the kernel no longer fits on a SIMD beside two waves of the
    // field backward, whose workgroups then could not be placed on any CU that held a marcher wave: the trainer runs this kernel
    // one step ahead on a side stream, and the field backward ran at half speed while it was resident.  Keep it <= 80 VGPRs.)
    float at0[KB], atc[KB];
    uint32_t acell[KB];
    uint32_t aocc[KB];  // one register each: a uint8_t array is byte-packed by the compiler, which consumes (waits for) every fetch at once
    while (true) {
      const int na = walk(at0, atc, acell);
      if (na == 0) break;
#pragma unroll
      for (int k = 0; k < KB; ++k) aocc[k] = a.bin[acell[k]];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        if (k < na) {
          const float t_clip = atc[k];
          if (aocc[k]) {
            if (!continuous) t_last = at0[k];
            while (true) {
              const float dt = fminf(fmaxf(t_last * a.cone, a.step), BIG);
              if (!(t_last + dt * 0.5f < t_clip)) break;
              if (WRITE) {
                a.t_starts[w] = t_last, a.t_ends[w] = t_last + dt, a.ray_indices[w] = r;
                ++w;
              } else if (a.cap > 0 && cnt < a.cap) {  // single pass: park the sample in the ray's scratch row
                a.t_starts[r * a.cap + cnt] = t_last, a.t_ends[r * a.cap + cnt] = t_last + dt;
              }
              ++cnt;
              t_last = t_last + dt;
            }
            continuous = true;
          } else {
            continuous = false;
          }
        }
      }
    }
  }
#ifdef MARCH_DEBUG_ITERS  // diagnostic build only: voxel steps instead of sample counts
  if (!WRITE) a.counts[r] = iters;
#else
  if (!WRITE) a.counts[r] = cnt;
#endif
}

static int fill_march(MarchArgs* a, const float* o, const float* d, int64_t n_rays, const uint8_t* bin, const float* roi6, int levels,
                      int res, float near_plane, float far_plane, float step, float cone, const float* nears, const float* fars,
                      const float* jitter, float jitter_step) {
  if (n_rays < 0 || !roi6 || (n_rays > 0 && (!o || !d || !bin))) return UMHS_ERR_ARG;
  if (levels < 1 || levels > 8 || res < 1 || res > 512 || !(step > 0.0f)) return UMHS_ERR_UNSUPPORTED;
  a->o = o, a->d = d, a->n_rays = n_rays, a->bin = bin, a->levels = levels, a->res = res;
  a->cx = (roi6[0] + roi6[3]) / 2.0f, a->cy = (roi6[1] + roi6[4]) / 2.0f, a->cz = (roi6[2] + roi6[5]) / 2.0f;
  a->hx = (roi6[3] - roi6[0]) / 2.0f, a->hy = (roi6[4] - roi6[1]) / 2.0f, a->hz = (roi6[5] - roi6[2]) / 2.0f;
  a->near = near_plane, a->far = far_plane, a->step = step, a->cone = cone, a->nears = nears, a->fars = fars;
  a->jitter = jitter, a->jitter_step = jitter_step;
  a->packed_info = nullptr, a->counts = nullptr, a->t_starts = a->t_ends = nullptr, a->ray_indices = nullptr, a->cap = 0;
  return UMHS_OK;
}

extern "C" int umhs_march_count(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                                const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                                float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                                float jitter_step, int64_t* counts, umhs_stream_t stream) {
  MarchArgs a;
  int rc = fill_march(&a, origins, directions, n_rays, binaries, roi_aabb_host6, levels, resolution, near_plane, far_plane,
                      step_size, cone_angle, nears, fars, jitter, jitter_step);
  if (rc) return rc;
  if (!counts && n_rays > 0) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  a.counts = counts;
  hipLaunchKernelGGL(march_kernel<false>, dim3((unsigned)((n_rays + MARCH_RPW - 1) / MARCH_RPW)), dim3(64), 0, umhs_s(stream), a);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_march_write(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                                const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                                float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                                float jitter_step, const int64_t* packed_info, float* t_starts, float* t_ends,
                                int64_t* ray_indices, umhs_stream_t stream) {
  MarchArgs a;
  int rc = fill_march(&a, origins, directions, n_rays, binaries, roi_aabb_host6, levels, resolution, near_plane, far_plane,
                      step_size, cone_angle, nears, fars, jitter, jitter_step);
  if (rc) return rc;
  if (n_rays == 0) return UMHS_OK;
  if (!packed_info || !t_starts || !t_ends || !ray_indices) return UMHS_ERR_ARG;
  a.packed_info = packed_info, a.t_starts = t_starts, a.t_ends = t_ends, a.ray_indices = ray_indices;
  hipLaunchKernelGGL(march_kernel<true>, dim3((unsigned)((n_rays + MARCH_RPW - 1) / MARCH_RPW)), dim3(64), 0, umhs_s(stream), a);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// Single-pass marching: one walk of the grid both counts the samples of every ray and parks the first `cap` of them in a scratch
// row [R, cap]; umhs_march_compact then copies the rows to their packed places (one wave per ray, coalesced).  A ray with more
// than `cap` samples only shows in its count: the caller falls back to umhs_march_write for that batch.
extern "C" int umhs_march_scratch(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                                  const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                                  float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                                  float jitter_step, int cap, int64_t* counts, float* scratch_t0, float* scratch_t1,
                                  umhs_stream_t stream) {
  MarchArgs a;
  int rc = fill_march(&a, origins, directions, n_rays, binaries, roi_aabb_host6, levels, resolution, near_plane, far_plane,
                      step_size, cone_angle, nears, fars, jitter, jitter_step);
  if (rc) return rc;
  if (n_rays == 0) return UMHS_OK;
  if (!counts || !scratch_t0 || !scratch_t1 || cap < 1) return UMHS_ERR_ARG;
  a.counts = counts, a.t_starts = scratch_t0, a.t_ends = scratch_t1, a.cap = cap;
  hipLaunchKernelGGL(march_kernel<false>, dim3((unsigned)((n_rays + MARCH_RPW - 1) / MARCH_RPW)), dim3(64), 0, umhs_s(stream), a);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

__global__ __launch_bounds__(256) void march_compact_kernel(const int64_t* __restrict__ pinfo, int64_t n_rays, int cap,
                                                            const float* __restrict__ s0, const float* __restrict__ s1,
                                                            float* __restrict__ t0, float* __restrict__ t1,
                                                            int64_t* __restrict__ ri) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const int64_t start = pinfo[2 * r], cnt = pinfo[2 * r + 1];
  for (int64_t i = lane; i < cnt; i += 64) {
    t0[start + i] = s0[r * cap + i], t1[start + i] = s1[r * cap + i];
    ri[start + i] = r;
  }
}

extern "C" int umhs_march_compact(const int64_t* packed_info, int64_t n_rays, int cap, const float* scratch_t0,
                                  const float* scratch_t1, float* t_starts, float* t_ends, int64_t* ray_indices,
                                  umhs_stream_t stream) {
  if (n_rays < 0 || cap < 1) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  if (!packed_info || !scratch_t0 || !scratch_t1 || !t_starts || !t_ends || !ray_indices) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(march_compact_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), packed_info, n_rays, cap,
                     scratch_t0, scratch_t1, t_starts, t_ends, ray_indices);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// visibility mask of nerfacc.render_visibility_from_density: keep sample n iff T_n = exp(-sum_{m<n} sigma dt) >= early_stop_eps
// and alpha_n = 1 - exp(-sigma dt) >= alpha_thre.  One wave per ray, 64-lane scan with carry (as the compositing kernels).
__global__ __launch_bounds__(256) void visibility_kernel(const float* __restrict__ sigma, const float* __restrict__ t0,
                                                         const float* __restrict__ t1, const int64_t* __restrict__ pinfo,
                                                         int64_t n_rays, float eps, float thre, uint8_t* __restrict__ mask,
                                                         int64_t* __restrict__ kept) {
  const int lane = threadIdx.x & 63;
  const int64_t r = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (r >= n_rays) return;
  const int64_t start = pinfo[2 * r];
  const int cnt = (int)pinfo[2 * r + 1];
  float carry = 0.0f;
  int64_t nkept = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    const bool valid = i < cnt;
    const float x = valid ? sigma[start + i] * (t1[start + i] - t0[start + i]) : 0.0f;
    const float incl = wave_inclusive_scan(x, lane);
    const float T = expf(-(carry + (incl - x))), alpha = 1.0f - expf(-x);
    const bool keep = valid && T >= eps && (thre <= 0.0f || alpha >= thre);
    if (valid) mask[start + i] = keep ? 1 : 0;
    nkept += __popcll(__ballot(keep));
    carry += __shfl(incl, 63, 64);
  }
  if (kept && lane == 0) kept[r] = nkept;
}

static int run_visibility(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info, int64_t n_rays,
                          int64_t n, float early_stop_eps, float alpha_thre, uint8_t* mask, int64_t* kept, umhs_stream_t stream) {
  if (n_rays < 0 || n < 0 || !packed_info || (n > 0 && (!sigma || !t_starts || !t_ends || !mask))) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  if (n == 0 && !kept) return UMHS_OK;
  hipLaunchKernelGGL(visibility_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), sigma, t_starts, t_ends,
                     packed_info, n_rays, early_stop_eps, alpha_thre, mask, kept);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_visibility(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                               int64_t n_rays, int64_t n, float early_stop_eps, float alpha_thre, uint8_t* mask,
                               umhs_stream_t stream) {
  return run_visibility(sigma, t_starts, t_ends, packed_info, n_rays, n, early_stop_eps, alpha_thre, mask, nullptr, stream);
}

// umhs_visibility + the number of surviving samples of every ray (kept[R]): what umhs_ray_prefix turns into the survivors' packed_info
extern "C" int umhs_visibility_count(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                                     int64_t n_rays, int64_t n, float early_stop_eps, float alpha_thre, uint8_t* mask,
                                     int64_t* kept, umhs_stream_t stream) {
  if (n_rays > 0 && !kept) return UMHS_ERR_ARG;
  return run_visibility(sigma, t_starts, t_ends, packed_info, n_rays, n, early_stop_eps, alpha_thre, mask, kept, stream);
}

// packed_info[r] = (sum of counts[< r], counts[r]); stats = (sum of counts, max of counts).  One workgroup: R is a ray batch
// (thousands), and the point is ONE launch where torch needs cumsum + sub + stack + max + stack (7 launches, each ~10 us of host
// time in the launch-bound stretch between the sampler's host syncs).
__global__ __launch_bounds__(1024) void ray_prefix_kernel(const int64_t* __restrict__ counts, int64_t n_rays,
                                                          int64_t* __restrict__ pinfo, int64_t* __restrict__ stats) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s, max_s[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int64_t carry = 0, vmax = 0;
  for (int64_t base = 0; base < n_rays; base += 1024) {
    const int64_t r = base + tid;
    const int64_t c = r < n_rays ? counts[r] : 0;
    vmax = c > vmax ? c : vmax;
    int64_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int64_t off = carry;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (r < n_rays) pinfo[2 * r] = off + incl - c, pinfo[2 * r + 1] = c;
    if (tid == 1023) carry_s = off + incl;
    __syncthreads();
    carry = carry_s;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int64_t o = __shfl_xor(vmax, d, 64);
    vmax = o > vmax ? o : vmax;
  }
  if (lane == 0) max_s[wv] = vmax;
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < 16; ++k) vmax = max_s[k] > vmax ? max_s[k] : vmax;
    stats[0] = carry, stats[1] = vmax;
  }
}

extern "C" int umhs_ray_prefix(const int64_t* counts, int64_t n_rays, int64_t* packed_info, int64_t* stats, umhs_stream_t stream) {
  if (n_rays < 0 || !stats || (n_rays > 0 && (!counts || !packed_info))) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(ray_prefix_kernel, dim3(1), dim3(1024), 0, umhs_s(stream), counts, n_rays, packed_info, stats);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// pos[i] = o[ri[i]] + d[ri[i]] * ((t0[i] + t1[i]) / 2): the sample midpoints the sampler's density query runs on (VolumetricSampler's
// sigma_fn: origins[ray_indices] + directions[ray_indices] * (t_starts + t_ends)[:, None] / 2.0 -- six torch launches; same
// operation order, contraction off, so the positions are the same bits).
__global__ __launch_bounds__(256) void sample_midpoints_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                               const int64_t* __restrict__ ri, const float* __restrict__ t0,
                                                               const float* __restrict__ t1, int64_t n, float* __restrict__ pos) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t r = ri[i];
  const float s = (t0[i] + t1[i]) / 2.0f;
#pragma unroll
  for (int q = 0; q < 3; ++q) pos[3 * i + q] = o[3 * r + q] + d[3 * r + q] * s;
}

extern "C" int umhs_sample_midpoints(const float* origins, const float* directions, const int64_t* ray_indices, const float* t_starts,
                                     const float* t_ends, int64_t n, float* positions, umhs_stream_t stream) {
  if (n < 0) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  if (!origins || !directions || !ray_indices || !t_starts || !t_ends || !positions) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(sample_midpoints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, umhs_s(stream), origins, directions,
                     ray_indices, t_starts, t_ends, n, positions);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// Survivors of the visibility mask, moved to their packed places: one wave per ray, order within the ray kept (= torch.nonzero +
// index_select of ray_indices / t_starts / t_ends, origins[ray_indices], directions[ray_indices], camera_indices[ray_indices] and
// pack_info of the result: twelve launches).  packed_in / packed_out: (start, count) of every ray before / after pruning
// (umhs_ray_prefix of umhs_visibility_count's kept[]); sel[j] = index of survivor j among the candidates.
__global__ __launch_bounds__(256) void compact_samples_kernel(const uint8_t* __restrict__ mask, const int64_t* __restrict__ pin,
                                                              const int64_t* __restrict__ pout, int64_t n_rays,
                                                              const float* __restrict__ t0, const float* __restrict__ t1,
                                                              const float* __restrict__ o, const float* __restrict__ d,
                                                              const int64_t* __restrict__ cam, int64_t* __restrict__ o_ri,
                                                              float* __restrict__ o_t0, float* __restrict__ o_t1,
                                                              float* __restrict__ o_o, float* __restrict__ o_d,
                                                              int64_t* __restrict__ o_cam, int64_t* __restrict__ sel) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const int64_t start = pin[2 * r], cnt = pin[2 * r + 1];
  int64_t w = pout[2 * r];
  const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2], dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
  const int64_t cr = cam ? cam[r] : 0;
  for (int64_t base = 0; base < cnt; base += 64) {
    const int64_t i = base + lane;
    const bool keep = i < cnt && mask[start + i] != 0;
    const unsigned long long b = __ballot(keep);
    if (keep) {
      const int64_t j = w + __popcll(b & ((1ull << lane) - 1ull));
      o_ri[j] = r, o_t0[j] = t0[start + i], o_t1[j] = t1[start + i], sel[j] = start + i;
      o_o[3 * j] = ox, o_o[3 * j + 1] = oy, o_o[3 * j + 2] = oz;
      o_d[3 * j] = dx, o_d[3 * j + 1] = dy, o_d[3 * j + 2] = dz;
      if (o_cam) o_cam[j] = cr;
    }
    w += __popcll(b);
  }
}

extern "C" int umhs_compact_samples(const uint8_t* mask, const int64_t* packed_in, const int64_t* packed_out, int64_t n_rays,
                                    const float* t_starts, const float* t_ends, const float* origins, const float* directions,
                                    const int64_t* camera_indices, int64_t* out_ray_indices, float* out_t_starts, float* out_t_ends,
                                    float* out_origins, float* out_directions, int64_t* out_camera_indices, int64_t* out_sel,
                                    umhs_stream_t stream) {
  if (n_rays < 0) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  if (!mask || !packed_in || !packed_out || !t_starts || !t_ends || !origins || !directions || !out_ray_indices || !out_t_starts ||
      !out_t_ends || !out_origins || !out_directions || !out_sel || ((camera_indices == nullptr) != (out_camera_indices == nullptr)))
    return UMHS_ERR_ARG;
  hipLaunchKernelGGL(compact_samples_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), mask, packed_in,
                     packed_out, n_rays, t_starts, t_ends, origins, directions, camera_indices, out_ray_indices, out_t_starts, out_t_ends,
                     out_origins, out_directions, out_camera_indices, out_sel);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}
