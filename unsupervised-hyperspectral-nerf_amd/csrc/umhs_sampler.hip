// occupancy-grid ray marcher and visibility pruning for gfx950.
// Replaces nerfacc==0.5.2 OccGridEstimator.sampling (CUDA-only traverse_grids + render_visibility_from_density) reached
// through nerfstudio's VolumetricSampler at umhs_model.py:201-209,229-237.  nerfacc's source is not available offline: the
// traversal below restates its published behaviour (see oracle/torch_ref.py march_ray_ref, which this kernel reproduces
// bit for bit: float32, fixed operation order, fp contraction off).  Two kernels: the voxel walk (march_walk_kernel: one WAVE per
// ray, the lanes start inside the ray and fall onto the sequential walk at their first voxel face -- exact, see there) leaves the runs
// of occupied voxels in a per-ray list; the emission (march_kernel: one thread per ray, the form the oracle is written in -- it walks
// the grid itself when it is given no lists) turns them into samples, counted and parked in per-ray scratch rows in one pass, or
// counted and then written at the offsets of an exclusive scan in two (the number of samples per ray is data dependent).
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
  // lists of umhs_march_walk (all NULL: the kernel walks the grid itself): the RUNS of consecutive occupied voxels among the first
  // vox_nv[r] voxels of ray r in walk order -- entry parameter of the run's first voxel, clipped exit parameter of its last, whether it
  // goes on from the entry in front of it (a run cut by a window of the walk: no reset of the sample recurrence) --, the chain value
  // behind those voxels, flags (1: the walk ended there, 2: the last of them was occupied)
  const float* vox_a;
  const float* vox_b;
  const uint8_t* vox_c;
  const int32_t* vox_n;   // entries
  const int32_t* vox_nv;  // voxels walked
  const float* vox_cur;
  const uint8_t* vox_flags;
  int vcap;
};
struct WalkOut {
  float* a;        // [R][vcap]
  float* b;        // [R][vcap]
  uint8_t* c;      // [R][vcap]
  int32_t* n;      // [R]
  int32_t* nv;     // [R]
  float* cur;      // [R]
  uint8_t* flags;  // [R]
  int vcap;
};

// ---- geometry shared by the two marchers -----------------------------------------------------------------------------------
struct RayGeo {
  float o[3], d[3], inv[3], c[3], h[3], ih[3], vs0[3], ivs0[3];
  float top, t, t_end;  // first chain parameter, end of the walk
  int levels, res;
};

__device__ __forceinline__ void march_setup(const MarchArgs& a, int64_t r, RayGeo& g) {
#pragma clang fp contract(off)
  const float BIG = 1e30f;
#pragma unroll
  for (int k = 0; k < 3; ++k) g.o[k] = a.o[3 * r + k], g.d[k] = a.d[3 * r + k];
  g.c[0] = a.cx, g.c[1] = a.cy, g.c[2] = a.cz, g.h[0] = a.hx, g.h[1] = a.hy, g.h[2] = a.hz;
#pragma unroll
  for (int k = 0; k < 3; ++k) g.inv[k] = g.d[k] != 0.0f ? 1.0f / g.d[k] : (signbit(g.d[k]) ? -BIG : BIG);
  // per-axis constants of level 0; level l scales them by 2^l exactly (power-of-two factors commute with rounding), so every
  // voxel step is multiplications only: a float divide is a ~10-instruction dependent chain, and a voxel step is one long chain
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    g.ih[k] = 1.0f / g.h[k];
    g.vs0[k] = (g.h[k] * 2.0f) / (float)a.res;
    g.ivs0[k] = 1.0f / g.vs0[k];
  }
  g.levels = a.levels, g.res = a.res;
  g.top = (float)(1 << (a.levels - 1));
  float tn = -INFINITY, tf = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float ho = g.h[k] * g.top;
    const float t0 = (g.c[k] - ho - g.o[k]) * g.inv[k], t1 = (g.c[k] + ho - g.o[k]) * g.inv[k];
    tn = fmaxf(tn, fminf(t0, t1)), tf = fminf(tf, fmaxf(t0, t1));
  }
  float near_r = a.nears ? a.nears[r] : a.near;
  if (a.jitter) near_r = near_r + a.jitter[r] * a.jitter_step;  // = torch's nears + rand * step (contraction is off)
  g.t = fmaxf(tn, near_r);
  g.t_end = fminf(tf, a.fars ? a.fars[r] : a.far);
}

// One step of the walk: the voxel the ray is in just behind chain parameter tt (< t_end) -> its clipped exit parameter (the next
// chain value) and its cell in the occupancy grid; false: the ray has left the grid.  The chain tt -> t_clip is a pure function of
// tt (it never looks at the occupancy), which is what march_wave_kernel builds on.
__device__ __forceinline__ bool march_voxel(const RayGeo& g, const float tt, float& t_clip, uint32_t& cell) {
#pragma clang fp contract(off)
  const float BIG = 1e30f;
  const float tm = tt + 1e-5f * fmaxf(1.0f, fabsf(tt));
  float p[3], m = 0.0f;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    p[q] = g.o[q] + g.d[q] * tm;
    m = fmaxf(m, fabsf(p[q] - g.c[q]) * g.ih[q]);
  }
  if (!(m < g.top)) return false;
  int lvl = 0;
  if (!(m < 1.0f)) lvl = (int)((__float_as_uint(m) >> 23) & 0xffu) - 126;  // frexp exponent: m in [2^(e-1), 2^e) -> level e
  lvl = min(max(lvl, 0), g.levels - 1);
  const float sc = __uint_as_float((uint32_t)(127 + lvl) << 23), isc = __uint_as_float((uint32_t)(127 - lvl) << 23);
  int idx[3];
  float t_exit = BIG;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float hl = g.h[q] * sc, vmin = g.c[q] - hl, vs = g.vs0[q] * sc, ivs = g.ivs0[q] * isc;
    int i = (int)floorf((p[q] - vmin) * ivs);
    i = min(max(i, 0), g.res - 1);
    idx[q] = i;
    const float lo = vmin + (float)i * vs, hi = vmin + (float)(i + 1) * vs;
    const float tx = ((g.d[q] >= 0.0f ? hi : lo) - g.o[q]) * g.inv[q];
    if (g.d[q] != 0.0f) t_exit = fminf(t_exit, tx);
  }
  if (!(t_exit > tt)) t_exit = nextafterf(tt, BIG);
  t_clip = fminf(t_exit, g.t_end);
  cell = (uint32_t)((((size_t)lvl * g.res + idx[0]) * g.res + idx[1]) * g.res + idx[2]);
  return true;
}

// ---- sample emission + (without voxel lists) the walk itself: one thread per ray ---------------------------------------------------
// This is the form the oracle is written in.  With the lists of march_walk_kernel it only replays them -- t_last += max(t_last * cone,
// step) in float is sequential by nature -- and walks on by itself from where a list ends (a ray with more than vcap voxels).
template <bool WRITE>
__global__ __launch_bounds__(64, 4) void march_kernel(MarchArgs a, int rpw) {
#pragma clang fp contract(off)
  // rpw rays per wave: the lanes of a wave run in lockstep through their rays' voxels, each voxel as long as the ray with the most
  // samples in it; a batch has a few thousand rays, the chip 1024 SIMDs -- fewer rays per wave = less waiting, more SIMDs
  // (march_rpw())
  if ((int)threadIdx.x >= rpw) return;
  const int64_t r = (int64_t)blockIdx.x * rpw + threadIdx.x;
  if (r >= a.n_rays) return;
  const float BIG = 1e30f;
  RayGeo g;
  march_setup(a, r, g);
  const float t = g.t, t_end = g.t_end;
  int64_t cnt = 0;
  int64_t w = WRITE ? a.packed_info[2 * r] : 0;
  int iters = 0;
  if (t < t_end) {
    bool continuous = false, left_grid = false;
    float t_last = t;
    int cnt32 = 0;  // (count / scratch forms: a ray has at most 2^31 samples)
    float* __restrict__ row0 = a.t_starts + r * a.cap;
    float* __restrict__ row1 = a.t_ends + r * a.cap;
    // one voxel [entry, t_clip) of the walk: samples while their mid-point lies inside it, if it is occupied
    auto emit = [&](const float entry, const float t_clip, const bool occupied) __attribute__((always_inline)) {
      if (occupied) {
        if (!continuous) t_last = entry;
        // One compare -> exec mask -> branch round trip per sample and nothing else divergent inside: the scratch row takes every
        // sample unconditionally (a ray with more than `cap` keeps overwriting its row's last slot -- its count then sends the batch
        // to the write pass anyway).  (Four speculative samples per round without a branch measured SLOWER: 151 vs 127 us at 16 rays
        // per wave; four-sample store groups: neutral.)
        if (WRITE) {
          while (true) {
            const float dt = fminf(fmaxf(t_last * a.cone, a.step), BIG);
            if (!(t_last + dt * 0.5f < t_clip)) break;
            a.t_starts[w] = t_last, a.t_ends[w] = t_last + dt, a.ray_indices[w] = r;
            ++w, ++cnt;
            t_last = t_last + dt;
          }
        } else if (a.cap > 0) {
          while (true) {
            const float dt = fminf(fmaxf(t_last * a.cone, a.step), BIG);
            if (!(t_last + dt * 0.5f < t_clip)) break;
            const int at = min(cnt32, a.cap - 1);
            row0[at] = t_last, row1[at] = t_last + dt;
            ++cnt32;
            t_last = t_last + dt;
          }
        } else {
          while (true) {
            const float dt = fminf(fmaxf(t_last * a.cone, a.step), BIG);
            if (!(t_last + dt * 0.5f < t_clip)) break;
            ++cnt32;
            t_last = t_last + dt;
          }
        }
        continuous = true;
      } else {
        continuous = false;
      }
    };
    float tt = t;
    int guard = 0;
    if (a.vox_a) {  // replay the list: 8 occupied voxels per fetch, the next fetch in flight while these are consumed
      const int total = a.vox_n[r];
      const float4* va = reinterpret_cast<const float4*>(a.vox_a + (size_t)r * a.vcap);
      const float4* vb = reinterpret_cast<const float4*>(a.vox_b + (size_t)r * a.vcap);
      const uint2* vc = reinterpret_cast<const uint2*>(a.vox_c + (size_t)r * a.vcap);
      float4 qa[2], qb[2], na[2], nb[2];
      uint2 qc, nc;
      if (total > 0) qa[0] = va[0], qa[1] = va[1], qb[0] = vb[0], qb[1] = vb[1], qc = vc[0];
      for (int k0 = 0; k0 < total; k0 += 8) {
        if (k0 + 8 < total) {
          const int f = (k0 >> 2) + 2;
          na[0] = va[f], na[1] = va[f + 1], nb[0] = vb[f], nb[1] = vb[f + 1], nc = vc[(k0 >> 3) + 1];
        }
        const float ea[8] = {qa[0].x, qa[0].y, qa[0].z, qa[0].w, qa[1].x, qa[1].y, qa[1].z, qa[1].w};
        const float eb[8] = {qb[0].x, qb[0].y, qb[0].z, qb[0].w, qb[1].x, qb[1].y, qb[1].z, qb[1].w};
        const uint32_t ec[2] = {qc.x, qc.y};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u < total) {
            continuous = ((ec[u >> 2] >> (8 * (u & 3))) & 0xffu) != 0;
            emit(ea[u], eb[u], true);
          }
        }
        qa[0] = na[0], qa[1] = na[1], qb[0] = nb[0], qb[1] = nb[1], qc = nc;
      }
      const uint32_t fl = a.vox_flags[r];
      tt = a.vox_cur[r], guard = a.vox_nv[r], iters = guard;
      continuous = (fl & 2u) != 0;
      if (fl & 1u) left_grid = true;  // (nothing left to walk)
    }
    constexpr int KB = 8;
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
          float t_clip;
          uint32_t cell;
          if (!march_voxel(g, tt, t_clip, cell)) {
            left_grid = true;
          } else {
            vt0[k] = tt, vtc[k] = t_clip, vcell[k] = cell;
            tt = t_clip;
            nb = k + 1;
          }
        }
      }
      return nb;
    };
    // The voxel sequence a ray crosses is pure geometry: it does not depend on what the occupancy grid says.  So the walk runs in
    // batches of KB voxels: geometry for KB steps, then the KB occupancy bytes fetched together (one L2 / MALL round trip instead
    // of KB), then the sample emission.
    float at0[KB], atc[KB];
    uint32_t acell[KB];
    uint32_t aocc[KB];  // one register each: a uint8_t array is byte-packed by the compiler, which consumes (waits for) every fetch at once
    while (true) {
      const int na = walk(at0, atc, acell);
      if (na == 0) break;
#pragma unroll
      for (int k = 0; k < KB; ++k) aocc[k] = a.bin[acell[k]];
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if (k < na) emit(at0[k], atc[k], aocc[k] != 0);
    }
    if (!WRITE) cnt = cnt32;
  }
#ifdef MARCH_DEBUG_ITERS  // diagnostic build only: voxel steps instead of sample counts
  if (!WRITE) a.counts[r] = iters;
#else
  if (!WRITE) a.counts[r] = cnt;
#endif
}

// ---- the walk (umhs_march_walk): one WAVE per ray, split over the 64 lanes --------------------------------------------------------
// A voxel step is a ~150-instruction dependent chain (~1,200 cycles for a lone wave) and a ray crosses ~350 voxels of a 4-level
// 128^3 grid: walked by march_kernel that chain is 0.45 ms for ANY number of rays up to 16 k (0.87 ms beside other kernels), chip idle.
// But the chain tt -> t_clip is a function of tt alone and t_clip is a voxel-face crossing: a walk started ANYWHERE falls onto the
// true chain at its first face.  So lane j starts at its own parameter s_j, takes the exit b_j of the voxel it finds itself in as
// its chain start and walks until it reaches b_(j+1); it is "joined" when it lands on b_(j+1) EXACTLY -- then, by induction from lane
// 0 (which starts on the chain), the concatenation of the lanes' voxels IS the sequential walk, bit for bit.  Where a lane is not
// joined (a sliver voxel shorter than the 1e-5 look-ahead, a nextafter fix-up, more than MARCH_VSEG voxels in one lane) the window
// ends at that lane and the next window starts on the chain where it stopped: no fallback path, no tolerance.  One such case is
// systematic: a ray almost parallel to a grid axis (|d_k| < ~0.01) reaches a face of that axis with o + d * (t + 1e-5 |t|) still
// rounding to the old side, and the chain then CREEPS, t -> nextafter(t), a few hundred steps per face.  A lane that meets a creeping
// step stops; the following windows give lane j the chain value j ulps ahead and one step each: 64 creeping steps per window.  The s_j divide the
// remaining range evenly in VOXELS, not in t (level l voxels are 2^l wide: the ray's intervals inside the nested level boxes give a
// piecewise-linear voxel measure).  The occupancy bytes are fetched by the walking lanes (nothing waits for them).  The voxels go to
// runs of occupied ones go to the ray's list [vcap] (start, end; ~20 per ray); march_kernel replays it (emission on one lane of the walking wave was measured first: 15 k single-lane
// instructions per ray, 0.26 ms for 4,096 rays and no faster than the serial marcher at 32,768 -- issue slots, not latency).
#define MARCH_VSEG 12
__device__ __forceinline__ float march_segment_start(const RayGeo& g, const float cur, const int lane) {
  // A[l] <= B[l]: the part of [cur, t_end] inside the level-l box, nested (an inner box the ray misses gets zero length)
  float A[8], B[8];
  float lo = cur, hi = g.t_end;
#pragma unroll
  for (int l = 7; l >= 0; --l) {
    if (l < g.levels) {
      const float sc = __uint_as_float((uint32_t)(127 + l) << 23);
      float tn = -INFINITY, tf = INFINITY;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float ho = g.h[k] * sc;
        const float t0 = (g.c[k] - ho - g.o[k]) * g.inv[k], t1 = (g.c[k] + ho - g.o[k]) * g.inv[k];
        tn = fmaxf(tn, fminf(t0, t1)), tf = fminf(tf, fmaxf(t0, t1));
      }
      A[l] = fminf(fmaxf(tn, lo), hi), B[l] = fminf(fmaxf(tf, A[l]), hi);
      lo = A[l], hi = B[l];
    } else {
      A[l] = cur, B[l] = g.t_end;
    }
  }
  // voxel measure of the pieces in t order: [A[l], A[l-1]] at weight 2^-l (l = levels-1 .. 1), [A[0], B[0]] at 1, [B[l-1], B[l]] at 2^-l
  float total = B[0] - A[0];
#pragma unroll
  for (int l = 1; l < 8; ++l)
    if (l < g.levels) total += ((A[l - 1] - A[l]) + (B[l] - B[l - 1])) * __uint_as_float((uint32_t)(127 - l) << 23);
  const float want = total * ((float)lane * (1.0f / 64.0f));
  float acc = 0.0f, s = cur;
#pragma unroll
  for (int l = 7; l >= 1; --l) {
    if (l < g.levels) {
      const float wgt = __uint_as_float((uint32_t)(127 - l) << 23), len = A[l - 1] - A[l];
      if (want >= acc) s = A[l] + fminf((want - acc) * __uint_as_float((uint32_t)(127 + l) << 23), len);
      acc += len * wgt;
    }
  }
  if (want >= acc) s = A[0] + fminf(want - acc, B[0] - A[0]);
  acc += B[0] - A[0];
#pragma unroll
  for (int l = 1; l < 8; ++l) {
    if (l < g.levels) {
      const float wgt = __uint_as_float((uint32_t)(127 - l) << 23), len = B[l] - B[l - 1];
      if (want >= acc) s = B[l - 1] + fminf((want - acc) * __uint_as_float((uint32_t)(127 + l) << 23), len);
      acc += len * wgt;
    }
  }
  return fminf(fmaxf(s, cur), g.t_end);
}

// x advanced by k >= 0 floats towards +inf (k applications of nextafterf(x, BIG); finite x)
__device__ __forceinline__ float march_ulps_up(const float x, const int k) {
  const uint32_t u = __float_as_uint(x);
  int key = (u & 0x80000000u) ? -(int)(u & 0x7fffffffu) : (int)u;  // floats in order as integers (-0 and +0 are both 0)
  key += k;
  return __uint_as_float(key >= 0 ? (uint32_t)key : (0x80000000u | (uint32_t)(-key)));
}

__global__ __launch_bounds__(64) void march_walk_kernel(MarchArgs a, WalkOut out, int vseg) {
#pragma clang fp contract(off)
  const int64_t r = blockIdx.x;
  const int lane = threadIdx.x;
  RayGeo g;
  march_setup(a, r, g);
  const float t_end = g.t_end;
  float* __restrict__ va = out.a + (size_t)r * out.vcap;
  float* __restrict__ vb = out.b + (size_t)r * out.vcap;
  uint8_t* __restrict__ vc = out.c + (size_t)r * out.vcap;
  int total = 0, voxels = 0;  // uniform: entries in the list, voxels walked
  float cur = g.t;            // uniform: the chain value the next window starts on
  bool left = false;          // uniform: the walk has left the grid
  bool carry = false;         // uniform: the last voxel walked was occupied
  {
    bool creep = false;  // uniform: the chain is advancing one ulp per step here (see below)
    while (cur < t_end && !left && voxels + 64 * MARCH_VSEG <= 100000) {  // (beyond: the emission kernel's own walk, with its exact guard)
      // every lane's start: lane 0 on the chain, lane j at the first voxel face behind its share of the remaining voxels
      float b = cur;
      if (creep) {
        if (lane > 0) b = march_ulps_up(cur, lane);
        if (!(b < t_end)) b = INFINITY;
      } else if (lane > 0) {
        b = INFINITY;
        const float s = march_segment_start(g, cur, lane);
        float tc;
        uint32_t cell;
        if (s < t_end && march_voxel(g, s, tc, cell)) b = tc;
      }
      const uint64_t none = __builtin_amdgcn_ballot_w64(b == INFINITY);  // (started at / behind the end of the walk)
      if (none != 0 && lane > (int)__builtin_ctzll(none)) b = INFINITY;   // nothing behind the first such lane either
      float nb = __shfl_down(b, 1, 64);
      if (lane == 63) nb = INFINITY;
      const bool mine = b != INFINITY;
      float tt = b;
      bool lleft = false, crept = false;
      float tclip[MARCH_VSEG];
      uint32_t occ[MARCH_VSEG];
      int n = 0;
      const int budget = creep ? 1 : vseg;
#pragma unroll
      for (int i = 0; i < MARCH_VSEG; ++i) {
        uint32_t cell = 0;
        tclip[i] = 0.0f;
        if (mine && !lleft && !crept && i < budget && tt < nb && tt < t_end) {
          float tc;
          if (march_voxel(g, tt, tc, cell)) {
            crept = tc == march_ulps_up(tt, 1);  // (a lane that meets a creeping stretch stops: the window ends at it)
            tclip[i] = tc, tt = tc, n = i + 1;
          } else {
            lleft = true, cell = 0;
          }
        }
        occ[i] = a.bin[cell];  // unconditional (cell 0 when there is no voxel): nothing in the walk waits for it
      }
      // joined: landed exactly on the next lane's start, or (nobody behind) walked to the end
      const bool joined = !mine || tt == nb || (nb == INFINITY && (lleft || !(tt < t_end)));
      const uint64_t open = __builtin_amdgcn_ballot_w64(!joined);
      const int J = open ? (int)__builtin_ctzll(open) : 63;  // lanes 0 .. J are the sequential walk
      // Lanes 0 .. J in order are the next voxels of the walk.  The list takes one entry per RUN of consecutive occupied voxels (the
      // emission only looks at a run's first entry parameter and its last exit: within a run the sample recurrence goes on from voxel
      // to voxel and "mid-point < exit" with the growing exit stops at the same sample) -- a run that crosses lanes is followed with
      // ballots to the lane it ends in; one that began in an earlier window goes on as an entry flagged "continues".
      const int ne = lane <= J ? n : 0;
      uint32_t obits = 0;
#pragma unroll
      for (int i = 0; i < MARCH_VSEG; ++i) obits |= (i < ne && occ[i] != 0 ? 1u : 0u) << i;
      const bool F = (obits & 1u) != 0;                                                // first voxel occupied
      const bool L = ne > 0 && ((obits >> (ne - 1)) & 1u) != 0;                       // last voxel occupied
      const bool A = ne > 0 && obits == ((1u << ne) - 1u);                            // all of them
      const uint64_t nonempty = __builtin_amdgcn_ballot_w64(ne > 0), lastm = __builtin_amdgcn_ballot_w64(L), firstm = __builtin_amdgcn_ballot_w64(F);
      const uint64_t lower = nonempty & ((1ull << lane) - 1ull), higher = lane < 63 ? nonempty & ~((2ull << lane) - 1ull) : 0ull;
      const bool pred = lower ? ((lastm >> (63 - __builtin_clzll(lower))) & 1ull) != 0 : carry;  // the voxel in front of this lane's first
      const bool link = L && higher != 0 && ((firstm >> __builtin_ctzll(higher)) & 1ull) != 0;   // this lane's last run goes on in the next lane
      const uint64_t stops = __builtin_amdgcn_ballot_w64(ne > 0 && !(A && link));                 // lanes a run that enters them ends in
      float fse = tclip[0];  // exit of the run that starts at this lane's first voxel
#pragma unroll
      for (int i = 1; i < MARCH_VSEG; ++i)
        if ((obits & ((2u << i) - 1u)) == ((2u << i) - 1u)) fse = tclip[i];
      const uint64_t after = lane < 63 ? stops & ~((2ull << lane) - 1ull) : 0ull;
      const float chain_end = __shfl(fse, after ? (int)__builtin_ctzll(after) : lane, 64);  // (read only when link: then `after` is not empty)
      // this lane's entries: every run that starts inside it, and its first run if that one starts a run of the ray (no occupied voxel
      // in front of it) or goes on from the previous window (first lane with voxels: entry flagged "continues")
      const uint32_t starts = obits & ~(obits << 1);
      const bool first_emits = F && (!pred || lower == 0);
      const uint32_t emits = (starts & ~1u) | (first_emits ? 1u : 0u);
      const int no = __builtin_popcount(emits);
      int incl = no, vincl = ne;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const int up = __shfl_up(incl, dd, 64), vup = __shfl_up(vincl, dd, 64);
        if (lane >= dd) incl += up, vincl += vup;
      }
      const int sum = __shfl(incl, 63, 64);
      if (total + sum > out.vcap) break;  // (uniform) the list is full: the emission kernel walks on from cur by itself
      int at = total + incl - no;
      {
        bool open_run = false, emit_it = false, cont_it = false;
        float run_start = 0.0f;
#pragma unroll
        for (int i = 0; i < MARCH_VSEG; ++i) {
          const bool bit = ((obits >> i) & 1u) != 0;
          if (bit && !open_run) {
            open_run = true, run_start = i == 0 ? b : tclip[i > 0 ? i - 1 : 0];
            emit_it = ((emits >> i) & 1u) != 0, cont_it = i == 0 && pred;
          }
          if (open_run && !((obits >> (i + 1)) & 1u)) {  // the run ends with voxel i (obits has no bit at or above ne)
            if (emit_it) {
              va[at] = run_start;
              vb[at] = (i == ne - 1 && link) ? chain_end : tclip[i];
              vc[at] = cont_it ? 1 : 0;
              ++at;
            }
            open_run = false;
          }
        }
      }
      total += sum, voxels += __shfl(vincl, 63, 64);
      if (nonempty) carry = ((lastm >> (63 - __builtin_clzll(nonempty))) & 1ull) != 0;
      creep = open != 0 && __builtin_amdgcn_readlane((int)crept, J) != 0;
      cur = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tt), J));
      left = __builtin_amdgcn_readlane((int)lleft, J) != 0;
    }
  }
  if (lane == 0)
    out.n[r] = total, out.nv[r] = voxels, out.cur[r] = cur, out.flags[r] = ((left || !(cur < t_end)) ? 1 : 0) | (carry ? 2 : 0);
}

static int march_vseg() {  // (tests shrink the lanes' voxel budget to force many windows)
  const char* e = getenv("UMHS_MARCH_VSEG");
  const int v = e ? atoi(e) : MARCH_VSEG;
  return v < 1 ? 1 : (v > MARCH_VSEG ? MARCH_VSEG : v);
}
static int march_vcap() {  // occupied voxels per ray a list holds (tests shrink it: the emission kernel then walks on by itself)
  const char* e = getenv("UMHS_MARCH_VCAP");
  int v = e ? atoi(e) : 512;
  v = v < 16 ? 16 : (v > 65536 ? 65536 : v);
  return (v + 15) & ~15;
}
static size_t walk_align(size_t x) { return (x + 255) & ~(size_t)255; }

static int march_rpw(int64_t n_rays) {  // rays per wave of the emission kernel: about 8,192 waves (eight per SIMD), 16 rays at most
  // (with the scratch rows: 4,096 rays 16 / 4 / 2 / 1 rays per wave = 196 / 147 / 71 / 58 us; 32,768 consecutive pixels of an eval
  //  image 280 / 185 / 195 / 232 us -- lanes wait for the ray with the longest run at their position)
  const char* e = getenv("UMHS_MARCH_RPW");
  if (e && atoi(e) >= 1 && atoi(e) <= 64) return atoi(e);
  int rpw = 1;
  while (rpw < 16 && n_rays > 8192 * (int64_t)rpw) rpw *= 2;
  return rpw;
}
template <bool WRITE>
static void march_launch(const MarchArgs& a, umhs_stream_t stream) {
  const int rpw = march_rpw(a.n_rays);
  hipLaunchKernelGGL(march_kernel<WRITE>, dim3((unsigned)((a.n_rays + rpw - 1) / rpw)), dim3(64), 0, umhs_s(stream), a, rpw);
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
  a->vox_a = a->vox_b = nullptr, a->vox_c = nullptr, a->vox_n = a->vox_nv = nullptr, a->vox_cur = nullptr, a->vox_flags = nullptr, a->vcap = 0;
  return UMHS_OK;
}

extern "C" size_t umhs_march_walk_workspace_bytes(int64_t n_rays) {
  if (n_rays <= 0) return 0;
  const size_t R = (size_t)n_rays, vcap = (size_t)march_vcap();
  return 256 + 2 * walk_align(R * vcap * 4) + walk_align(R * vcap) + 3 * walk_align(R * 4) + walk_align(R);
}
static int walk_carve(void* ws, size_t bytes, int64_t n_rays, WalkOut* o) {
  if (!ws || bytes < umhs_march_walk_workspace_bytes(n_rays)) return UMHS_ERR_WORKSPACE;
  const size_t R = (size_t)n_rays, vcap = (size_t)march_vcap();
  uintptr_t p = ((uintptr_t)ws + 255) & ~(uintptr_t)255;
  o->a = reinterpret_cast<float*>(p), p += walk_align(R * vcap * 4);
  o->b = reinterpret_cast<float*>(p), p += walk_align(R * vcap * 4);
  o->c = reinterpret_cast<uint8_t*>(p), p += walk_align(R * vcap);
  o->n = reinterpret_cast<int32_t*>(p), p += walk_align(R * 4);
  o->nv = reinterpret_cast<int32_t*>(p), p += walk_align(R * 4);
  o->cur = reinterpret_cast<float*>(p), p += walk_align(R * 4);
  o->flags = reinterpret_cast<uint8_t*>(p);
  o->vcap = (int)vcap;
  return UMHS_OK;
}
// the lists of umhs_march_walk for the emission kernels (walked == NULL: they walk the grid themselves)
static int march_use_walked(MarchArgs* a, const void* walked, size_t walked_bytes) {
  if (!walked) return UMHS_OK;
  WalkOut o;
  int rc = walk_carve(const_cast<void*>(walked), walked_bytes, a->n_rays, &o);
  if (rc) return rc;
  a->vox_a = o.a, a->vox_b = o.b, a->vox_c = o.c, a->vox_n = o.n, a->vox_nv = o.nv, a->vox_cur = o.cur, a->vox_flags = o.flags, a->vcap = o.vcap;
  return UMHS_OK;
}

extern "C" int umhs_march_walk(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                               const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                               const float* nears, const float* fars, const float* jitter, float jitter_step, void* walked,
                               size_t walked_bytes, umhs_stream_t stream) {
  MarchArgs a;
  int rc = fill_march(&a, origins, directions, n_rays, binaries, roi_aabb_host6, levels, resolution, near_plane, far_plane, 1.0f, 0.0f,
                      nears, fars, jitter, jitter_step);
  if (rc) return rc;
  if (n_rays == 0) return UMHS_OK;
  WalkOut o;
  rc = walk_carve(walked, walked_bytes, n_rays, &o);
  if (rc) return rc;
  hipLaunchKernelGGL(march_walk_kernel, dim3((unsigned)n_rays), dim3(64), 0, umhs_s(stream), a, o, march_vseg());
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_march_count(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                                const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                                float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                                float jitter_step, int64_t* counts, const void* walked, size_t walked_bytes, umhs_stream_t stream) {
  MarchArgs a;
  int rc = fill_march(&a, origins, directions, n_rays, binaries, roi_aabb_host6, levels, resolution, near_plane, far_plane,
                      step_size, cone_angle, nears, fars, jitter, jitter_step);
  if (rc) return rc;
  if (!counts && n_rays > 0) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  a.counts = counts;
  rc = march_use_walked(&a, walked, walked_bytes);
  if (rc) return rc;
  march_launch<false>(a, stream);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_march_write(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                                const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                                float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                                float jitter_step, const int64_t* packed_info, float* t_starts, float* t_ends,
                                int64_t* ray_indices, const void* walked, size_t walked_bytes, umhs_stream_t stream) {
  MarchArgs a;
  int rc = fill_march(&a, origins, directions, n_rays, binaries, roi_aabb_host6, levels, resolution, near_plane, far_plane,
                      step_size, cone_angle, nears, fars, jitter, jitter_step);
  if (rc) return rc;
  if (n_rays == 0) return UMHS_OK;
  if (!packed_info || !t_starts || !t_ends || !ray_indices) return UMHS_ERR_ARG;
  a.packed_info = packed_info, a.t_starts = t_starts, a.t_ends = t_ends, a.ray_indices = ray_indices;
  rc = march_use_walked(&a, walked, walked_bytes);
  if (rc) return rc;
  march_launch<true>(a, stream);
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
                                  const void* walked, size_t walked_bytes, umhs_stream_t stream) {
  MarchArgs a;
  int rc = fill_march(&a, origins, directions, n_rays, binaries, roi_aabb_host6, levels, resolution, near_plane, far_plane,
                      step_size, cone_angle, nears, fars, jitter, jitter_step);
  if (rc) return rc;
  if (n_rays == 0) return UMHS_OK;
  if (!counts || !scratch_t0 || !scratch_t1 || cap < 1) return UMHS_ERR_ARG;
  a.counts = counts, a.t_starts = scratch_t0, a.t_ends = scratch_t1, a.cap = cap;
  rc = march_use_walked(&a, walked, walked_bytes);
  if (rc) return rc;
  march_launch<false>(a, stream);
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
