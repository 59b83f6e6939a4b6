// gfx950 kernels for the memory-bound rows of the UMHS hot path: sample positions (R1 prefix),
// multires hash-grid encode fwd/bwd (R2), packed transmittance + per-ray band accumulation fwd/bwd
// (R11-R13), spectrum->sRGB fwd/bwd (R14) and the fused Adam step.  The MFMA field kernels live in
// umhs_field.hip.  Reference citations are in include/umhs_hip.h.
#include <cstdlib>

#include "umhs_common.h"
#include <atomic>

// =============================================================================================
// R1 prefix: positions
// =============================================================================================
__global__ __launch_bounds__(256) void positions_kernel(const float* __restrict__ origins,
                                                        const float* __restrict__ directions,
                                                        const float* __restrict__ starts,
                                                        const float* __restrict__ ends,
                                                        const float* __restrict__ world_in, int64_t n,
                                                        int contraction, float ax, float ay, float az, float bx,
                                                        float by, float bz, float* __restrict__ world_out,
                                                        float* __restrict__ pos01, float* __restrict__ selector) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float p[3];
  if (world_in) {
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = world_in[3 * i + c];
  } else {
    float t = starts[i] + ends[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = origins[3 * i + c] + directions[3 * i + c] * t / 2.0f;
  }
  if (world_out) {
#pragma unroll
    for (int c = 0; c < 3; ++c) world_out[3 * i + c] = p[c];
  }
  float q[3];
  if (contraction) {
    float mag = fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2])));
    if (mag < 1.0f) {
      q[0] = p[0], q[1] = p[1], q[2] = p[2];
    } else {
      float sc = 2.0f - (1.0f / mag);
#pragma unroll
      for (int c = 0; c < 3; ++c) q[c] = sc * (p[c] / mag);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = (q[c] + 2.0f) / 4.0f;
  } else {
    q[0] = (p[0] - ax) / (bx - ax);
    q[1] = (p[1] - ay) / (by - ay);
    q[2] = (p[2] - az) / (bz - az);
  }
  bool sel = q[0] > 0.0f && q[0] < 1.0f && q[1] > 0.0f && q[1] < 1.0f && q[2] > 0.0f && q[2] < 1.0f;
  float sf = sel ? 1.0f : 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) pos01[3 * i + c] = q[c] * sf;
  if (selector) selector[i] = sf;
}

extern "C" int umhs_positions_fwd(const float* origins, const float* directions, const float* starts,
                                  const float* ends, const float* world_pos_in, int64_t n, int contraction,
                                  const float* aabb, float* world_pos_out, float* pos01_out, float* selector_out,
                                  umhs_stream_t stream) {
  if (n < 0 || !pos01_out) return UMHS_ERR_ARG;
  if (!world_pos_in && (!origins || !directions || !starts || !ends)) return UMHS_ERR_ARG;
  if (!contraction && !aabb) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  float a[6] = {-1, -1, -1, 1, 1, 1};
  if (aabb)
    for (int i = 0; i < 6; ++i) a[i] = aabb[i];
  dim3 grid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(positions_kernel, grid, dim3(256), 0, umhs_s(stream), origins, directions, starts, ends,
                     world_pos_in, n, contraction, a[0], a[1], a[2], a[3], a[4], a[5], world_pos_out, pos01_out,
                     selector_out);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// =============================================================================================
// R2: multires hash grid.  One thread per (sample, level); grid.y = level, so blocks are dispatched
// level-major and the resident waves of an XCD gather from one 4 MiB level slab (= one XCD L2) at
// a time instead of from the whole 64 MiB table.
// =============================================================================================
// (Two / four levels per thread -- shared position math, more gathers in flight -- were measured in round 2: 92 / 110 / 128 us; a
// thread that walks two level slabs undoes the one-slab-per-XCD locality.  Removed in round 3.)
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(const float* __restrict__ pos01,
                                                           const float2* __restrict__ table,
                                                           const float* __restrict__ scalings, int64_t n, int n_levels,
                                                           int log2_T, float* __restrict__ enc, int64_t stride_n,
                                                           int64_t stride_l) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float px = pos01[3 * i], py = pos01[3 * i + 1], pz = pos01[3 * i + 2];
  const int l = (int)blockIdx.y;
  HashCorners h = hash_corners(px, py, pz, scalings[l], (1u << log2_T) - 1u, (uint32_t)l << log2_T);
#ifdef HF_ABL_SAMEADDR  // (ablation builds of tools/alt_kernels.py only): every lane gathers the level's first slots -> what is left is not the gather
#pragma unroll
  for (int c = 0; c < 8; ++c) h.idx[c] = ((uint32_t)l << log2_T) + (h.idx[c] & 7u);
#endif
  float2 f[8];
  hash_gather8(table, h, f);
  const float2 r = hash_trilerp(f, h.ox, h.oy, h.oz);
  float* o = enc + i * stride_n + (int64_t)l * stride_l;
  if (((stride_n | stride_l) & 1) == 0) {
    *reinterpret_cast<float2*>(o) = r;
  } else {
    o[0] = r.x, o[1] = r.y;
  }
}

extern "C" int umhs_hashgrid_fwd(const float* pos01, const float* table, const float* scalings, int64_t n,
                                 int n_levels, int log2_T, float* enc, int64_t stride_n, int64_t stride_l,
                                 umhs_stream_t stream) {
  if (n < 0 || !table || !scalings) return UMHS_ERR_ARG;
  if (n_levels < 1 || n_levels > 32 || log2_T < 1 || log2_T > 24) return UMHS_ERR_UNSUPPORTED;
  if (n == 0) return UMHS_OK;  // (an empty batch has no per-sample arrays: torch hands out NULL for them)
  if (!pos01 || !enc) return UMHS_ERR_ARG;
  if (((uintptr_t)table & 15) || ((uintptr_t)enc & 7)) return UMHS_ERR_ARG;  // (16-byte slot pairs are fetched with one load)
  const float2* t2 = reinterpret_cast<const float2*>(table);
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)n_levels);
  hipLaunchKernelGGL(hashgrid_fwd_kernel, grid, dim3(256), 0, umhs_s(stream), pos01, t2, scalings, n, n_levels, log2_T, enc, stride_n, stride_l);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// Backward v1: memory-side float atomics, one (sample, level) per thread, level-major grid.
__global__ __launch_bounds__(256) void hashgrid_bwd_kernel(const float* __restrict__ pos01,
                                                           const float* __restrict__ d_enc, int64_t stride_n,
                                                           int64_t stride_l, const float* __restrict__ scalings,
                                                           int64_t n, int log2_T, float* __restrict__ d_table, int level0) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int l = level0 + blockIdx.y;
  const float* g = d_enc + i * stride_n + (int64_t)l * stride_l;
  float g0 = g[0], g1 = g[1];
  if (g0 == 0.0f && g1 == 0.0f) return;  // masked / zero-weight samples contribute exact zeros
  HashCorners h = hash_corners(pos01[3 * i], pos01[3 * i + 1], pos01[3 * i + 2], scalings[l],
                               (1u << log2_T) - 1u, (uint32_t)l << log2_T);
  float ox = h.ox, oy = h.oy, oz = h.oz, rx = 1.0f - ox, ry = 1.0f - oy, rz = 1.0f - oz;
  float w[8];
  w[0] = ox * oy * oz, w[3] = rx * oy * oz, w[1] = ox * ry * oz, w[2] = rx * ry * oz;
  w[4] = ox * oy * rz, w[7] = rx * oy * rz, w[5] = ox * ry * rz, w[6] = rx * ry * rz;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (w[c] != 0.0f) {
      atomicAdd(d_table + 2 * (size_t)h.idx[c], w[c] * g0);
      atomicAdd(d_table + 2 * (size_t)h.idx[c] + 1, w[c] * g1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward v2 (default): no global atomics.  MI355X executes global float atomics at the memory side at
// ~20 G requests/s whatever the schedule, so 8 corners x 16 levels x N scattered adds cost ~6 ms at
// N = 262k.  Instead each level's contributions are radix-partitioned by the high bits of their hash
// slot into buckets of 2^13 slots, every (level, bucket) tile is accumulated in LDS by one workgroup
// and added to d_table with plain coalesced stores:
//   count   : per (level, 512-sample run) LDS histogram of bucket ids -> per-workgroup counts
//   scan    : exclusive prefix over the workgroups of a level (hg_wgscan) and over its buckets (hg_scan)
//   scatter : recompute the corners, order the run's records by bucket in LDS, write them out
//   reduce  : one workgroup per (level, bucket): stream its records into an LDS tile, flush (+ Adam)
// Round 4 (in-kernel stamps, profiles/r04/hg_stamps_*.json): neither pass was bound where rounds 1-3 said.  The scatter pass spent
// 40 % of its wave time writing records out and 47 % waiting on its few loads behind those stores (hashing + run merging: 3 %); the
// reduce pass 42-55 % waiting for record loads and 39 % in the Adam stream, 7 % in LDS atomics (the LDS unit alone would do the whole
// pass in 30 us: tools/mb_lds_atomics2.hip).  What cost the time was the SHAPE of the record stream: {uint16 slot, float2 value} in
// two arrays = a 2-byte and an 8-byte access per record (MI355X_MICROARCH.md: short stores cost 12.5x, dwordx2 2.7x the dwordx4 time
// per byte).  Records are now ONE 16-byte word each and there are half as many:
//   * the two x-neighbours of a corner pair hash to slots s and s ^ (xf ^ xc), i.e. into the same bucket (xf ^ xc < 2^13 for every
//     resolution below 8192), and their values are g*wyz*(1-ox) and g*wyz*ox: one PAIR record {g.x*wyz, g.y*wyz, ox, meta} serves
//     both corners (meta = slot_low | k << 13, xf ^ xc = 2^(k+1) - 1) -- 4 records of 16 B per (sample, level) instead of 8 of 10 B,
//     one dwordx4 store / load each, half the LDS placement work; the reduce pass forms the two corner values;
//   * a run of samples that share a cell (coarse / mid levels of a real ray batch: merged by a wave segmented scan as before) emits
//     its 8 corner sums as SINGLE records {v.x, v.y, 0, meta} (k = 15: no partner); so do the (x-integer) and the (partner in another
//     bucket: resolutions >= 8192 only) cases.
// Two MI355X-specific choices kept from round 1 (tools/mb_lds_atomics2.hip): LDS float atomics cost 81 ns per wave-instruction but
// 64-bit INTEGER ones 7.5, so the tile is int64 fixed point (scale from the level's max |value| and the bucket's record count:
// >= 46 bits below the level maximum) -- which also makes every sum exact and order-independent, i.e. bitwise reproducible,
// unlike the reference's index_put_/atomics.
// ---------------------------------------------------------------------------------------------
#ifndef HB_BUCKET_BITS
#define HB_BUCKET_BITS 13
#endif
#define HB_MAX_NB 128  // buckets per level the partition kernels can handle (two per lane of wave 0)
#ifndef HB_SPT
#define HB_SPT 2  // samples per thread -> 512 samples per workgroup (<= 2048 pair records staged in 32 KiB of LDS)
#endif
#define HB_CAP_PER_SAMPLE 5  // record capacity per (sample, level): 4 pair records, or <= 4 per sample from merged runs; the fifth is
                             // slack for pairs split over two buckets (resolutions >= 8192); a level that overflows gets NaN gradients
#ifndef HB_MERGE_MIN
#define HB_MERGE_MIN 16  // lanes of a wave that must continue a run of equal cells for the wave to merge runs
#endif
#define HB_POISON 0xffffffffu
// One word per scatter workgroup and level for the level's max |value|, reduced by the readers.  Rounds 1-3 (and the first round-4
// builds) did `atomicMax(&lmax[level])` once per workgroup run: 8192 device-scope atomics on 16 words of ONE cache line serialise at
// the memory side at ~12 ns each = 100 us -- the whole scatter pass, whatever else it did (ablations of tools/alt_kernels.py:
// 98 us with every store, LDS placement and the write-out removed, 16 us once the maximum stayed zero).
#define HB_LMAX_PARTS 128

// One Adam update (torch.optim.Adam, no weight decay / amsgrad); one expression for the stand-alone kernels and for the
// epilogue of hg_reduce_kernel, so that the fused and the separate update give the same bits.
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, const float gk, const float lr_bc1, const float b1,
                                            const float b2, const float eps, const float sqrt_bc2) {
#pragma clang fp contract(off)  // the same rounding in every kernel this is inlined into (and torch's mul_/add_ sequence)
  m = m * b1 + gk * (1.0f - b1);
  v = v * b2 + gk * gk * (1.0f - b2);
  const float denom = sqrtf(v) / sqrt_bc2 + eps;
  p = p - lr_bc1 * (m / denom);
}

struct HbAdam {  // optional optimizer step in the epilogue of the bucket reduce (one GPU: the gradient is final there)
  float *p, *m, *v;  // hash table parameters / exp_avg / exp_avg_sq, [L*T, 2] like d_table; p == nullptr: off
  float lr_bc1, b1, b2, eps, sqrt_bc2;
  int level_begin;   // absolute level from which on the update is applied (the sparse coarse levels keep their row-wise kernel)
};

struct HbArgs {
  HbAdam adam;
  const float* pos01;
  const float* d_enc;
  int64_t sn, sl;
  const float* scalings;
  int64_t n;
  int log2_T, bucket_bits, nb, level0, nlev;  // level0: first level of the WORKSPACE range; nlev: its size
  int lev_off;                                 // this launch covers workspace levels [lev_off, lev_off + gridDim.y)
  int grad_mask;  // 1: samples whose gradient is exactly zero emit no records (both passes then need d_enc); 0: every sample does
  uint32_t *counts, *offsets;       // [nlev * nb]: records per (level, bucket) and their exclusive prefix INSIDE the level
  uint32_t *wg_counts, *wg_prefix;  // [nlev][nwg][nb]: per-workgroup bucket histogram, and (hg_wgscan) its exclusive prefix over
  int nwg;                          // the workgroups of the level = each workgroup's private, atomics-free place in every bucket
  uint32_t* lmax;                   // [nlev][HB_LMAX_PARTS] bits of the max |record value| seen by each scatter workgroup of the level
                                    // ([..][0] = HB_POISON, set by hg_scan: the level's records do not fit its region)
  uint4* recs;                      // [nlev][cap] 16-byte records
  uint32_t cap;                     // record capacity per level
  int overwrite;                    // reduce: d_table slab = tile (zeros where untouched) instead of +=
};

// DPP row_shr:D -- lane l receives the value of lane l-D of its 16-lane row (0 when l%16 < D).  Pure VALU: unlike
// __shfl_up (ds_bpermute) it does not go through the LDS unit, which this kernel already loads with its atomics.
template <int D>
__device__ __forceinline__ float row_shr(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xF, 0xF, true));
}
template <int D>
__device__ __forceinline__ uint32_t row_shr_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + D, 0xF, 0xF, true);
}
__device__ __forceinline__ int row_shl1(int v) {  // lane l <- lane l+1 of its row (0 at the row end)
  return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true);
}
// 64-lane inclusive prefix sum on the VALU alone (gfx9 DPP: four shifts inside the 16-lane rows, then row_bcast:15 into rows 1 and 3
// and row_bcast:31 into the upper half): the scatter pass is bound by VALU + LDS issue, and a __shfl_up scan is six ds_bpermute
// round trips through the LDS unit per 64 values.
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t v) {
  v += row_shr_u<1>(v);
  v += row_shr_u<2>(v);
  v += row_shr_u<4>(v);
  v += row_shr_u<8>(v);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}
// maximum over the 64 lanes, uniform result (row-wise DPP, then the four row results through readlane)
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
  v = max(v, row_shr_u<1>(v)), v = max(v, row_shr_u<2>(v)), v = max(v, row_shr_u<4>(v)), v = max(v, row_shr_u<8>(v));  // lane 15 of a row: its max (values >= 0)
  return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 15), (uint32_t)__builtin_amdgcn_readlane((int)v, 31)),
             max((uint32_t)__builtin_amdgcn_readlane((int)v, 47), (uint32_t)__builtin_amdgcn_readlane((int)v, 63)));
}
template <int D>
__device__ __forceinline__ void seg_scan_step(float2 (&val)[8], bool& f, int l16) {
  const bool take = l16 >= D && !f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float px = row_shr<D>(val[c].x), py = row_shr<D>(val[c].y);
    if (take) val[c].x += px, val[c].y += py;
  }
  const int pf = (int)row_shr_u<D>((uint32_t)f);
  if (l16 >= D) f = f || pf;
}

// In-kernel phase stamps of the partitioned backward (tools/stamp_hg.py builds this file with -DUMHS_HG_STAMP into its own library;
// no stamp executes in the product).  Every wave sums the cycles between consecutive stamps per phase and adds them to
// g_hg_stamp[kernel][level][phase] once, at its end; [..][15] counts the waves.  HG_STAMP_DRAIN also waits for the wave's outstanding
// vector-memory operations first, so that a phase that issues loads or stores is charged with their completion.
#ifdef UMHS_HG_STAMP
__device__ unsigned long long g_hg_stamp[2][16][16];
#define HG_STAMP_DECL unsigned long long hgs_acc_[15] = {}, hgs_t_ = hg_now_(false)
__device__ __forceinline__ unsigned long long hg_now_(bool drain) {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  if (drain) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  } else {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  }
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define HG_STAMP(k_)                              \
  do {                                            \
    const unsigned long long n_ = hg_now_(false); \
    hgs_acc_[k_] += n_ - hgs_t_, hgs_t_ = n_;     \
  } while (0)
#define HG_STAMP_DRAIN(k_)                       \
  do {                                           \
    const unsigned long long n_ = hg_now_(true); \
    hgs_acc_[k_] += n_ - hgs_t_, hgs_t_ = n_;    \
  } while (0)
#define HG_STAMP_FLUSH(kern_, lev_)                                                                                     \
  do {                                                                                                                  \
    if ((threadIdx.x & 63) == 0) {                                                                                      \
      _Pragma("unroll") for (int q_ = 0; q_ < 15; ++q_) if (hgs_acc_[q_]) atomicAdd(&g_hg_stamp[kern_][(lev_) & 15][q_], hgs_acc_[q_]); \
      atomicAdd(&g_hg_stamp[kern_][(lev_) & 15][15], 1ull);                                                             \
    }                                                                                                                   \
  } while (0)
extern "C" int umhs_debug_hg_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hg_stamp), sizeof(unsigned long long) * 2 * 16 * 16);
}
extern "C" int umhs_debug_hg_stamps_clear() {
  static unsigned long long z[2 * 16 * 16] = {};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_hg_stamp), z, sizeof(z));
}
#else
#define HG_STAMP_DECL \
  do {                \
  } while (0)
#define HG_STAMP(k_) \
  do {               \
  } while (0)
#define HG_STAMP_DRAIN(k_) \
  do {                     \
  } while (0)
#define HG_STAMP_FLUSH(kern_, lev_) \
  do {                              \
  } while (0)
#endif

// x-pair p of a (sample, level): floor-x corner FI[p], ceil-x corner CI[p] of HashCorners' corner order; their common (y, z) weight
__device__ constexpr int HB_FI[4] = {3, 2, 7, 6}, HB_CI[4] = {0, 1, 4, 5};

// record meta word: [23:0] slot index inside the level (its bits above bucket_bits = the bucket), [27:24] k: the partner slot is
// slot ^ (2^(k+1) - 1); 15: no partner

// Everything a workgroup reads from memory for one run of 256 * HB_SPT samples.  Requested in ONE batch (hb_load), a whole run ahead
// of its use by the scatter pass's persistent workgroups: the stamps of round 4 showed the pass as a chain of dependent memory
// latencies per workgroup (level flag -> bucket counts / prefix -> barrier -> gradient -> position: 47 % of a wave's time at 12-16
// waves per CU) behind the CU's own queue of record stores, not as bandwidth.  Lanes past the end load the last sample
// (unconditional loads stay batched; a load under a per-lane condition compiles to a branch + s_waitcnt vmcnt(0)).
struct HbIn {
  float g[HB_SPT][2], p[HB_SPT][3];
  uint32_t c[2], mb[2];  // wave 0: this workgroup's record count in buckets lane / lane + 64 and where its slice of them starts
};

template <bool SCATTER>
__device__ __forceinline__ void hb_load(const HbArgs& a, const int wg, const int lev, const int l, HbIn& in) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < HB_SPT; ++k) {
    const int64_t i = (int64_t)wg * (256 * HB_SPT) + k * 256 + tid, ii = i < a.n ? i : a.n - 1;
    in.g[k][0] = in.g[k][1] = 0.0f;
    if (SCATTER || a.grad_mask) {  // the histogram pass of a prepare/apply pair runs before any gradient exists
      const float* g = a.d_enc + ii * a.sn + (int64_t)l * a.sl;
      if (((a.sn | a.sl) & 1) == 0 && (((uintptr_t)a.d_enc) & 7) == 0) {
        const float2 g2 = *reinterpret_cast<const float2*>(g);
        in.g[k][0] = g2.x, in.g[k][1] = g2.y;
      } else {
        in.g[k][0] = g[0], in.g[k][1] = g[1];
      }
    }
    in.p[k][0] = a.pos01[3 * ii], in.p[k][1] = a.pos01[3 * ii + 1], in.p[k][2] = a.pos01[3 * ii + 2];
  }
  in.c[0] = in.c[1] = in.mb[0] = in.mb[1] = 0u;
  if (SCATTER && tid < 64) {  // the histogram pass left this workgroup's bucket counts and hg_wgscan its place in every bucket
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int bk = tid + 64 * h;
      if (bk < a.nb) {
        const size_t o = ((size_t)lev * a.nwg + wg) * a.nb + bk;
        in.c[h] = a.wg_counts[o];
        in.mb[h] = a.offsets[lev * a.nb + bk] + a.wg_prefix[o];
      }
    }
  }
}

template <bool SCATTER>
__device__ __forceinline__ void hg_partition_body(const HbArgs& a, const int wg, const int lev, const int l, const float s, const HbIn& in,
                                                  uint32_t& wgmax) {
  // cursor[b]: histogram pass: records of bucket b; scatter pass: where the next record of bucket b goes in the LDS staging array
  // (starts at the run's first place, so a returning atomic add IS the place).  delta[b]: (global place) - (staged place) of bucket b.
  __shared__ uint32_t cursor[HB_MAX_NB];
  __shared__ uint32_t delta[HB_MAX_NB];
  __shared__ uint32_t ltotal;
  // scatter pass: records are first ordered by bucket in LDS, then written out with consecutive lanes on consecutive
  // records (tools/mb_scatter_store.hip: 5.7 TB/s in this shape, 3.2 TB/s with every lane storing its own record where it belongs)
  constexpr int MAXREC = SCATTER ? 256 * HB_SPT * 4 : 1;
  __shared__ uint4 stage[MAXREC];
  const int tid = threadIdx.x, lane = tid & 63;
  HG_STAMP_DECL;
  if (!SCATTER && tid < HB_MAX_NB) cursor[tid] = 0;
  // scatter pass: nothing is counted again and no global cursor is touched -- wave 0 turns the workgroup's bucket counts into LDS offsets
  if (SCATTER && tid < 64) {
    uint32_t carry = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int bk = tid + 64 * h;
      const uint32_t c = in.c[h], mb = in.mb[h];
      const uint32_t incl = wave_scan_incl_dpp(c);
      const uint32_t first = carry + incl - c;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      // this workgroup's slice of the bucket: bucket start + the records of the workgroups before it -- no cursor atomics
      cursor[bk] = first, delta[bk] = mb - first;
    }
    if (tid == 0) ltotal = carry;
  }
  __syncthreads();
  const uint32_t mask = (1u << a.log2_T) - 1u;
  const int bb = a.bucket_bits;
  uint4* const __restrict__ out = a.recs + (size_t)lev * a.cap;
  const bool staged = SCATTER && ltotal <= (uint32_t)MAXREC;  // (pairs split over two buckets can exceed 4 per sample: then straight to memory)
  if (SCATTER) HG_STAMP_DRAIN(0);
  // max |record value| of the level as raw bits: for non-negative floats the integer order is the float order, and an Inf / NaN
  // pattern (>= 0x7f800000) beats every finite one -- hg_reduce turns a level that saw one into NaN gradients instead of
  // an arbitrary fixed-point conversion (fmaxf would silently drop a NaN; the reference's index_add propagates it)
  uint32_t vmax = 0u;
  auto emit = [&](const uint32_t meta, const float vx, const float vy, const float ox) {  // meta = slot index | k << 24
    const uint32_t b = __builtin_amdgcn_ubfe(meta, (uint32_t)bb, (uint32_t)(24 - bb));
#ifdef HB_ABL_NOPLACE  // (ablation builds of tools/alt_kernels.py only: never defined in the product)
    asm volatile("" ::"v"(vx), "v"(vy), "v"(ox), "v"(b));
    return;
#endif
    const uint32_t pos = atomicAdd(&cursor[b], 1u);
    if (!SCATTER) return;
    const uint4 r = make_uint4(__float_as_uint(vx), __float_as_uint(vy), __float_as_uint(ox), meta);
    if (staged)
      stage[pos] = r;
    else
      out[delta[b] + pos] = r;
  };
#pragma unroll
  for (int k = 0; k < HB_SPT; ++k) {
    const int64_t i = (int64_t)wg * (256 * HB_SPT) + k * 256 + tid;
    bool act = false;
    float g0 = 0.0f, g1 = 0.0f, ox = 0.0f, oy = 0.0f, oz = 0.0f;
    uint32_t kx = 0xffffffffu, ky = 0, kz = 0, kf = 0x80000000u | (uint32_t)lane;  // unique per lane when inactive
    uint32_t slot[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) slot[c] = 0;
    if (SCATTER) HG_STAMP_DRAIN(1);
#ifdef HB_ABL_NOHASH
    asm volatile("" ::"v"(in.g[k][0]), "v"(in.g[k][1]), "v"(in.p[k][0]), "v"(in.p[k][1]), "v"(in.p[k][2]));
    if (false) {
#else
    if (i < a.n) {
#endif
      g0 = in.g[k][0], g1 = in.g[k][1];
      if (!a.grad_mask || g0 != 0.0f || g1 != 0.0f) {
        act = true;
        HashCorners h = hash_corners(in.p[k][0], in.p[k][1], in.p[k][2], s, mask, 0u);
        ox = h.ox, oy = h.oy, oz = h.oz;
#pragma unroll
        for (int c = 0; c < 8; ++c) slot[c] = h.idx[c];
        // cell identity: floor coordinates + "coordinate is an exact integer" flags (ceil == floor)
        kx = h.fx, ky = h.fy, kz = h.fz, kf = h.eqx | (h.eqy << 1) | (h.eqz << 2);
      }
    }
    // runs of equal cells are merged inside each 16-lane DPP row (all cross-lane ops executed by every lane)
    if (SCATTER) HG_STAMP(2);
    const int l16 = lane & 15;
    const uint32_t px_ = row_shr_u<1>(kx), py_ = row_shr_u<1>(ky), pz_ = row_shr_u<1>(kz), pf_ = row_shr_u<1>(kf);
    bool head = (l16 == 0) | (px_ != kx) | (py_ != ky) | (pz_ != kz) | (pf_ != kf);
    // Merging is a wave-wide decision (the scan below is ~450 VALU instructions per sample for all 64 lanes): it is taken where it
    // removes records in earnest -- at least a quarter of the wave's samples continue a run -- and otherwise every sample stays a
    // run of its own.  (Both passes see the same positions, hence take the same decision.)
#ifdef HB_ABL_NODPP
    head = true;
    const bool merging = false;
#else
    const bool merging = __builtin_popcountll(__builtin_amdgcn_ballot_w64(!head)) >= HB_MERGE_MIN;
#endif
    if (!merging) head = true;
    const int nhead = merging ? row_shl1((int)head) : 1;
    const bool tail = act && (l16 == 15 || nhead);  // tail lane of a run of equal cells emits for the run
    const bool solo = head && tail;                 // a run of one sample: pair records
    const float rx = 1.0f - ox, ry = 1.0f - oy, rz = 1.0f - oz;
    float2 val[8];
    // the 8 corner sums of a merged run: segmented inclusive scan over the row, (f, v) (+) (pf, pv) = (f | pf, f ? v : v + pv)
    if (SCATTER && merging) {
      float w[8];
      w[0] = ox * oy * oz, w[3] = rx * oy * oz, w[1] = ox * ry * oz, w[2] = rx * ry * oz;
      w[4] = ox * oy * rz, w[7] = rx * oy * rz, w[5] = ox * ry * rz, w[6] = rx * ry * rz;
#pragma unroll
      for (int c = 0; c < 8; ++c) val[c] = act ? make_float2(w[c] * g0, w[c] * g1) : make_float2(0.0f, 0.0f);
      bool f = head;
      seg_scan_step<1>(val, f, l16), seg_scan_step<2>(val, f, l16);
      seg_scan_step<4>(val, f, l16), seg_scan_step<8>(val, f, l16);
    }
    if (SCATTER) HG_STAMP(3);
#ifdef HB_ABL_NOEMIT
    asm volatile("" ::"v"(slot[0]), "v"(slot[1]), "v"(slot[2]), "v"(slot[3]), "v"(slot[4]), "v"(slot[5]), "v"(slot[6]), "v"(slot[7]), "v"(ox), "v"(oy), "v"(oz), "v"(g0), "v"(g1));
    if (false) {
#else
    if (tail) {
#endif
      const uint32_t single = 15u << 24;
      if (solo) {
        // the x-neighbours of all four pairs differ by the same pattern (xf ^ xc) & mask = 2^(k+1) - 1; |g wyz| <= |g|: one maximum per sample
        const uint32_t pm = slot[HB_FI[0]] ^ slot[HB_CI[0]];
        const float wyz[4] = {oy * oz, ry * oz, oy * rz, ry * rz};  // (y, z) weight of x-pair p: (c,c) (f,c) (c,f) (f,f)
        if (SCATTER) vmax = max(vmax, max(__float_as_uint(fabsf(g0)), __float_as_uint(fabsf(g1))));
        if ((pm >> bb) == 0) {  // both corners in one bucket (always below resolution 8192): one record, the reduce pass splits it
          const uint32_t km = (pm ? (uint32_t)(31 - __clz((int)pm)) : 15u) << 24;  // (pm == 0: x is an integer, ox == 0, all weight on the floor slot)
          // (the four places first, then the four records: four returning LDS atomics in flight instead of one round trip per record)
          uint32_t meta[4], pos[4], bk[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            meta[p] = slot[HB_FI[p]] | km;
            bk[p] = __builtin_amdgcn_ubfe(meta[p], (uint32_t)bb, (uint32_t)(24 - bb));
#ifdef HB_ABL_NOPLACE
            pos[p] = bk[p];
#else
            pos[p] = atomicAdd(&cursor[bk[p]], 1u);
#endif
          }
          if (SCATTER) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const uint4 r = make_uint4(__float_as_uint(g0 * wyz[p]), __float_as_uint(g1 * wyz[p]), __float_as_uint(ox), meta[p]);
#ifdef HB_ABL_NOPLACE
              asm volatile("" ::"v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w), "v"(pos[p]));
#else
              if (staged)
                stage[pos[p]] = r;
              else
                out[delta[bk[p]] + pos[p]] = r;
#endif
            }
          }
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float gx = g0 * wyz[p], gy = g1 * wyz[p];
            emit(slot[HB_FI[p]] | single, gx * rx, gy * rx, 0.0f);
            emit(slot[HB_CI[p]] | single, gx * ox, gy * ox, 0.0f);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (SCATTER) vmax = max(vmax, max(__float_as_uint(fabsf(val[c].x)), __float_as_uint(fabsf(val[c].y))));
          emit(slot[c] | single, SCATTER ? val[c].x : 0.0f, SCATTER ? val[c].y : 0.0f, 0.0f);
        }
      }
    }
    if (SCATTER) HG_STAMP(4);
  }
  if (!SCATTER) {
    __syncthreads();
    if (tid < a.nb) a.wg_counts[((size_t)lev * a.nwg + wg) * a.nb + tid] = cursor[tid];
    return;
  }
  vmax = wave_max_u32_dpp(vmax);
  if (lane == 0 && vmax > 0u) atomicMax(&wgmax, vmax);  // (LDS; the kernel writes the workgroup's maximum out once, after its last run)
  HG_STAMP(5);
  __syncthreads();
  HG_STAMP(6);
  if (staged) {
    // byte offsets inside the level's record region fit 32 bits (umhs_hashgrid_bwd_workspace_bytes): scalar base + 32-bit offset stores
    char* const ob = reinterpret_cast<char*>(out);
#ifdef HB_ABL_NOWRITEOUT
    const uint32_t total = 0;
#else
    const uint32_t total = ltotal;
#endif
    for (uint32_t i = tid; i < total; i += 256) {
      const uint4 r = stage[i];
      const uint32_t b = __builtin_amdgcn_ubfe(r.w, (uint32_t)bb, (uint32_t)(24 - bb));
#ifdef HB_ABL_NOSTORE
      asm volatile("" ::"v"(r.x), "v"(r.y), "v"(r.z), "v"(delta[b] + i));
#else
      *reinterpret_cast<uint4*>(ob + (size_t)((delta[b] + i) << 4)) = r;
#endif
    }
  }
  HG_STAMP_DRAIN(7);
  HG_STAMP_FLUSH(0, l);
}

// The histogram half of hg_partition_body for ONE sample per lane, as a function of its own: which records the scatter pass will emit
// for the sample (run detection over the 16-lane DPP row, the wave-wide merge decision, pair / single / split records), counted into
// the workgroup's bucket counters.  hashgrid_fwd_count_kernel below calls it: the forward gather has hashed every (sample, level)
// anyway and is bound by the vector-memory path with the VALU idle, so the histogram pass of the backward -- 45 us of hashing, DPP and
// LDS atomics, hidden on the side stream but 16 us of the step all the same (measured: the step without it) -- rides along for free.
// MUST stay the exact mirror of hg_partition_body<false> (tests/test_hip_parity.py compares the counts bit for bit).
__device__ __forceinline__ void hb_count_sample(const uint32_t (&slot)[8], const uint32_t kx, const uint32_t ky, const uint32_t kz, const uint32_t kf,
                                                const bool act, const int bb, const int lane, uint32_t* __restrict__ cursor) {
  const int l16 = lane & 15;
  const uint32_t px_ = row_shr_u<1>(kx), py_ = row_shr_u<1>(ky), pz_ = row_shr_u<1>(kz), pf_ = row_shr_u<1>(kf);
  bool head = (l16 == 0) | (px_ != kx) | (py_ != ky) | (pz_ != kz) | (pf_ != kf);
  const bool merging = __builtin_popcountll(__builtin_amdgcn_ballot_w64(!head)) >= HB_MERGE_MIN;
  if (!merging) head = true;
  const int nhead = merging ? row_shl1((int)head) : 1;
  const bool tail = act && (l16 == 15 || nhead);
  const bool solo = head && tail;
  if (tail) {
    if (solo) {
      // (written out: left as loops the compiler kept them rolled and moved slot[] into LDS for the dynamic index)
      const uint32_t pm = slot[3] ^ slot[0];  // HB_FI[0], HB_CI[0]
      static_assert(HB_FI[0] == 3 && HB_FI[1] == 2 && HB_FI[2] == 7 && HB_FI[3] == 6 && HB_CI[0] == 0, "corner order");
      atomicAdd(&cursor[slot[3] >> bb], 1u), atomicAdd(&cursor[slot[2] >> bb], 1u);
      atomicAdd(&cursor[slot[7] >> bb], 1u), atomicAdd(&cursor[slot[6] >> bb], 1u);
      if ((pm >> bb) != 0) {  // the x-neighbours live in two buckets: every pair becomes two singles
        atomicAdd(&cursor[slot[0] >> bb], 1u), atomicAdd(&cursor[slot[1] >> bb], 1u);
        atomicAdd(&cursor[slot[4] >> bb], 1u), atomicAdd(&cursor[slot[5] >> bb], 1u);
      }
    } else {
      atomicAdd(&cursor[slot[0] >> bb], 1u), atomicAdd(&cursor[slot[1] >> bb], 1u);
      atomicAdd(&cursor[slot[2] >> bb], 1u), atomicAdd(&cursor[slot[3] >> bb], 1u);
      atomicAdd(&cursor[slot[4] >> bb], 1u), atomicAdd(&cursor[slot[5] >> bb], 1u);
      atomicAdd(&cursor[slot[6] >> bb], 1u), atomicAdd(&cursor[slot[7] >> bb], 1u);
    }
  }
}

// hashgrid_fwd_kernel + the backward's bucket histogram: one workgroup = one run of 512 samples of one level (the scatter pass's unit)
static_assert(256 * HB_SPT == 512, "hashgrid_fwd_count_kernel's workgroup is one run of the partition");
__global__ __launch_bounds__(512) void hashgrid_fwd_count_kernel(const float2* __restrict__ table, float* __restrict__ enc, int64_t stride_n,
                                                                 int64_t stride_l, HbArgs a) {
  __shared__ uint32_t cursor[HB_MAX_NB];
  const int tid = threadIdx.x, lane = tid & 63, wg = blockIdx.x, l = blockIdx.y;  // (workspace range starts at level 0: lev == l)
  if (tid < HB_MAX_NB) cursor[tid] = 0;
  __syncthreads();
  const int64_t i = (int64_t)wg * 512 + tid;
  const bool act = i < a.n;
  const int64_t ii = act ? i : a.n - 1;
  const uint32_t mask = (1u << a.log2_T) - 1u, base = (uint32_t)l << a.log2_T;
  const HashCorners h = hash_corners(a.pos01[3 * ii], a.pos01[3 * ii + 1], a.pos01[3 * ii + 2], a.scalings[l], mask, base);
  HashGather hg;
  hash_gather8_issue(table, h, hg);  // the loads are in flight while the wave counts
  __builtin_amdgcn_sched_barrier(0);
  uint32_t slot[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) slot[c] = h.idx[c] - base;
  const uint32_t kx = act ? h.fx : 0xffffffffu, ky = act ? h.fy : 0u, kz = act ? h.fz : 0u;
  const uint32_t kf = act ? (h.eqx | (h.eqy << 1) | (h.eqz << 2)) : (0x80000000u | (uint32_t)lane);  // unique per lane when inactive
  hb_count_sample(slot, kx, ky, kz, kf, act, a.bucket_bits, lane, cursor);
  __syncthreads();
  if (tid < a.nb) a.wg_counts[((size_t)l * a.nwg + wg) * a.nb + tid] = cursor[tid];
  __builtin_amdgcn_sched_barrier(0);
  float2 f[8];
  hash_gather8_select(h, hg, f);
  const float2 r = hash_trilerp(f, h.ox, h.oy, h.oz);
  if (act) {
    float* o = enc + i * stride_n + (int64_t)l * stride_l;
    if (((stride_n | stride_l) & 1) == 0) {
      *reinterpret_cast<float2*>(o) = r;
    } else {
      o[0] = r.x, o[1] = r.y;
    }
  }
}

// Scatter pass: PERSISTENT workgroups, gridDim.x (a multiple of 8) per level; each walks its runs of samples with the next run's
// inputs in flight (hb_load above).  Which runs: workgroups go to the 8 XCDs round-robin by their linear index, and the runs wg,
// wg + 1 write ADJACENT record runs in every bucket (short ones on the coarse levels: most 128-byte lines of the record stream are
// shared by neighbouring runs) -- XCD x takes the CONTIGUOUS runs [x * per, (x + 1) * per) and its workgroups interleave inside
// that range, so that neighbouring runs are written through the same L2 at about the same time and their partial lines combine
// there (tools/mb_scatter_store.hip: 16-byte pieces 3.3 vs 1.3 TB/s, 32-byte 5.7 vs 2.7).
// Histogram pass: gridDim.x workgroups per level walk the runs -- a caller that hides the pass under other kernels
// (umhs_hashgrid_bwd_prepare on a side stream) launches few, so that it takes a small, steady share of the CUs instead of flooding
// the dispatcher in front of the kernels it overlaps with.
template <bool SCATTER>
__global__ __launch_bounds__(256) void hg_partition_kernel(HbArgs a) {
  const int lev = a.lev_off + blockIdx.y, l = a.level0 + lev;
  const float s = a.scalings[l];
  __shared__ uint32_t wgmax;
  HbIn cur;
  if (threadIdx.x == 0) wgmax = 0;  // (ordered before its first use by the barrier inside the body)
  if (SCATTER) {
    const int per = (a.nwg + 7) >> 3, q = (int)(gridDim.x >> 3);  // runs per XCD, workgroups per XCD (and level)
    const int x = (int)(blockIdx.x & 7u), end = min(a.nwg, (x + 1) * per);
    int wg = x * per + (int)(blockIdx.x >> 3);
    const uint32_t lmax0 = a.lmax[(size_t)lev * HB_LMAX_PARTS];
    if (wg >= end) return;
    hb_load<true>(a, wg, lev, l, cur);
    if (lmax0 == HB_POISON) return;  // (uniform) the level's records do not fit its region: hg_reduce writes NaN
    while (true) {
      const int nxt = wg + q;
      HbIn nx;
      if (nxt < end) hb_load<true>(a, nxt, lev, l, nx);  // (uniform branch)
      hg_partition_body<true>(a, wg, lev, l, s, cur, wgmax);
      if (nxt >= end) break;
      cur = nx, wg = nxt;
      __syncthreads();  // the write-out of this run has read the staged records before the next run's placement overwrites them
    }
    __syncthreads();
    // a plain store into the workgroup's own word (hg_scan zeroed them): no atomic, nothing shared
    if (threadIdx.x == 0 && wgmax) a.lmax[(size_t)lev * HB_LMAX_PARTS + (blockIdx.x % HB_LMAX_PARTS)] = wgmax;
  } else {
    for (int wg = blockIdx.x; wg < a.nwg; wg += gridDim.x) {
      hb_load<false>(a, wg, lev, l, cur);
      hg_partition_body<false>(a, wg, lev, l, s, cur, wgmax);
      __syncthreads();  // (the histogram is zeroed again at the top of the next run)
    }
  }
}

// Per (level, bucket): exclusive prefix of the per-workgroup bucket counts over the level's workgroups, and the bucket total.
// One 256-thread workgroup per (bucket, level): a thread sums its run of consecutive workgroups, the runs are scanned across the
// block, and the thread writes its run's prefixes.  Also clears the level's max-|value| word for the scatter pass.
__global__ __launch_bounds__(256) void hg_wgscan_kernel(HbArgs a) {
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x, lev = blockIdx.y;
  const int per = (a.nwg + 255) / 256, w0 = min(a.nwg, tid * per), w1 = min(a.nwg, w0 + per);
  const uint32_t* __restrict__ col = a.wg_counts + (size_t)lev * a.nwg * a.nb + b;
  uint32_t* __restrict__ pre = a.wg_prefix + (size_t)lev * a.nwg * a.nb + b;
  uint32_t sum = 0;
  for (int w = w0; w < w1; ++w) sum += col[(size_t)w * a.nb];
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint32_t run = incl - sum, total = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < wv) run += wsum[k];
    total += wsum[k];
  }
  for (int w = w0; w < w1; ++w) {
    pre[(size_t)w * a.nb] = run;
    run += col[(size_t)w * a.nb];
  }
  if (tid == 0) a.counts[lev * a.nb + b] = total;
}

// One wave per level: exclusive scan of the level's bucket counts (nb <= 128: two per lane) -> offsets inside the level's record
// region; sets the level's max-|value| word to 0, or to HB_POISON when its records exceed the region.
__global__ void hg_scan_kernel(HbArgs a) {
  const int lane = threadIdx.x, lev = blockIdx.x;
  uint32_t carry = 0;
  for (int b0 = 0; b0 < a.nb; b0 += 64) {
    const int b = b0 + lane;
    const uint32_t c = b < a.nb ? a.counts[lev * a.nb + b] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (b < a.nb) a.offsets[lev * a.nb + b] = carry + incl - c;
    carry += __shfl(incl, 63, 64);
  }
  for (int j = lane; j < HB_LMAX_PARTS; j += 64) a.lmax[(size_t)lev * HB_LMAX_PARTS + j] = (j == 0 && carry > a.cap) ? HB_POISON : 0u;
}

// x = value * 2^(k - 32) -> floor(value * 2^k) as a 64-bit two's complement integer: hi = floor(x) (signed), lo = (x - floor(x)) * 2^32
// (exact: the remainder of a float has at most 24 bits).  Six VALU instructions; __float2ll_rn(ldexpf(v, k)) compiles to fourteen, and
// the record phase of the reduce pass -- four conversions per record -- was bound by exactly that plus the LDS atomics (round 4:
// records stream at 3.2 TB/s, the Adam epilogue at the HBM rate).  Rounds down instead of to nearest: <= 2^-46 of the level maximum
// per addend, the same for every order of the addends.
__device__ __forceinline__ unsigned long long hb_fixed(const float x) {
  const float fl = floorf(x);
  const int hi = (int)fl;
  const uint32_t lo = (uint32_t)((x - fl) * 4294967296.0f);
  return ((unsigned long long)(uint32_t)hi << 32) | lo;
}

__global__ __launch_bounds__(1024) void hg_reduce_kernel(HbArgs a, float* __restrict__ d_table) {
  extern __shared__ __attribute__((aligned(16))) long long tile[];  // [2 << bucket_bits] int64 fixed point
  // (levels in dispatch order.  Last level first -- the records the scatter pass wrote last are the likeliest to sit in the 256 MiB
  // Infinity Cache -- was measured in round 4: 211-220 vs 186-204 us at C2, 361-382 vs 325-347 us at C5.  Dropped.)
  const int tid = threadIdx.x, b = blockIdx.x, lev = a.lev_off + blockIdx.y, l = a.level0 + lev;
  HG_STAMP_DECL;
  const uint32_t start = a.offsets[lev * a.nb + b], cnt = a.counts[lev * a.nb + b];
  const int nsl = 2 << a.bucket_bits;
  const size_t slab = 2 * (((size_t)l << a.log2_T) + ((size_t)b << a.bucket_bits));  // element offset of this (level, bucket)
  const bool adam = a.adam.p != nullptr && l >= a.adam.level_begin;
  auto step4 = [&](const float4& g, float4& pp, float4& mm, float4& vv) {  // Adam on 4 consecutive table entries whose final gradient is g
    adam_update(pp.x, mm.x, vv.x, g.x, a.adam.lr_bc1, a.adam.b1, a.adam.b2, a.adam.eps, a.adam.sqrt_bc2);
    adam_update(pp.y, mm.y, vv.y, g.y, a.adam.lr_bc1, a.adam.b1, a.adam.b2, a.adam.eps, a.adam.sqrt_bc2);
    adam_update(pp.z, mm.z, vv.z, g.z, a.adam.lr_bc1, a.adam.b1, a.adam.b2, a.adam.eps, a.adam.sqrt_bc2);
    adam_update(pp.w, mm.w, vv.w, g.w, a.adam.lr_bc1, a.adam.b1, a.adam.b2, a.adam.eps, a.adam.sqrt_bc2);
  };
  // epilogue of every path: the slab's gradient (+ its Adam step) as float4 lanes; the optimizer operands of a thread's (up to)
  // four chunks are requested before anything is computed -- three loads in flight per chunk, not three per thread
  // the level's max |value| = max over its scatter workgroups' words (one word per lane, L2 hits; every wave computes it for itself)
  const uint32_t lmax_bits =
      wave_max_u32_dpp(max(a.lmax[(size_t)lev * HB_LMAX_PARTS + (tid & 63)], a.lmax[(size_t)lev * HB_LMAX_PARTS + 64 + (tid & 63)]));
  // a non-finite gradient reached this level: the slabs its records land in are NaN, as after the reference's index_add; a level
  // whose records did not fit its region (HB_POISON) is NaN everywhere
  const bool nan_level = lmax_bits >= 0x7f800000u && (cnt != 0 || lmax_bits == HB_POISON);
  const bool from_tile = cnt != 0 && !nan_level;
  if (!from_tile && !a.overwrite && !nan_level) return;  // nothing lands in this slab and it is not ours to zero
  int kfix = 0;
  if (from_tile) {
    for (int i = tid; i < nsl; i += 1024) tile[i] = 0;
    // fixed-point scale 2^k:  |v| <= vmax < 2^e, at most cnt < 2^hb addends  =>  |sum| * 2^k < 2^62
    int e;
    (void)frexpf(__uint_as_float(lmax_bits), &e);
    const int hb = 33 - __clz(cnt);  // cnt < 2^(32-clz) ; one spare bit
    kfix = min(62 - hb - e, 150);  // (2^(kfix - 32) must be a finite float: levels whose largest |value| is below 2^-60)
    __syncthreads();
    const uint4* __restrict__ rp = a.recs + (size_t)lev * a.cap + start;
    typedef unsigned long long u64;
    u64* ut = reinterpret_cast<u64*>(tile);
    const uint32_t lowmask = (1u << a.bucket_bits) - 1u;
    const float fscale = ldexpf(1.0f, kfix - 32);
    auto add = [&](const uint4& r) {
#ifdef HR_ABL_NOADD  // (ablation builds of tools/alt_kernels.py only)
      asm volatile("" ::"v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w));
      return;
#endif
      const float vx = __uint_as_float(r.x), vy = __uint_as_float(r.y), ox = __uint_as_float(r.z);
      const uint32_t s = r.w & lowmask, kk = (r.w >> 24) & 15u;
      const float rx = (1.0f - ox) * fscale, oxs = ox * fscale;  // the weights carry the fixed-point scale 2^(kfix - 32)
#ifdef HR_ABL_NOATOMIC
      asm volatile("" ::"v"(hb_fixed(vx * rx)), "v"(hb_fixed(vy * rx)), "v"(hb_fixed(vx * oxs)), "v"(hb_fixed(vy * oxs)), "v"(s), "v"(kk));
      return;
#endif
      atomicAdd(&ut[2 * s], hb_fixed(vx * rx)), atomicAdd(&ut[2 * s + 1], hb_fixed(vy * rx));
      if (kk != 15u) {
        const uint32_t cs = s ^ (((2u << kk) - 1u) & lowmask);
        atomicAdd(&ut[2 * cs], hb_fixed(vx * oxs)), atomicAdd(&ut[2 * cs + 1], hb_fixed(vy * oxs));
      }
    };
    uint32_t i = tid;
    HG_STAMP(0);
    // 4 records per thread and batch, the NEXT batch requested before this one is accumulated (every record slot past the end
    // re-reads the bucket's last record and is dropped: unconditional loads stay batched).  Two register sets in turn, no copies:
    // with `r = n` moves at the loop's end hipcc waits for vmcnt(0) at its top -- in front of the next requests -- and nothing overlaps.
    const uint32_t last = cnt - 1;
    auto load4 = [&](uint4 (&r)[4], const uint32_t at) {
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = rp[min(at + 1024u * u, last)];
    };
    auto add4 = [&](const uint4 (&r)[4], const uint32_t at) {
      add(r[0]);
#pragma unroll
      for (int u = 1; u < 4; ++u)
        if (at + 1024u * u < cnt) add(r[u]);
    };
    uint4 ra[4], rb[4];
    if (i < cnt) {
      load4(ra, i);
      while (true) {
        HG_STAMP_DRAIN(1);
        load4(rb, i + 4096u);
        add4(ra, i);
        HG_STAMP(3);
        i += 4096u;
        if (i >= cnt) break;
        HG_STAMP_DRAIN(1);
        load4(ra, i + 4096u);
        add4(rb, i);
        HG_STAMP(3);
        i += 4096u;
        if (i >= cnt) break;
      }
    }
    HG_STAMP_DRAIN(3);
    __syncthreads();
    HG_STAMP(4);
  }
  float* const dst = d_table + slab;
  const float qnan = __uint_as_float(0x7fc00000u);
  for (int j0 = tid * 4; j0 < nsl; j0 += 4 * 4096) {
    float4 pp[4], mm[4], vv[4], dd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = min(j0 + c * 4096, nsl - 4);  // (chunks past the slab re-read its last float4 and are not stored)
      if (adam) {
        pp[c] = *reinterpret_cast<const float4*>(a.adam.p + slab + j), mm[c] = *reinterpret_cast<const float4*>(a.adam.m + slab + j);
        vv[c] = *reinterpret_cast<const float4*>(a.adam.v + slab + j);
      }
      if (!a.overwrite) dd[c] = *reinterpret_cast<const float4*>(dst + j);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + c * 4096;
      if (j < nsl) {
        float4 d = a.overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : dd[c];
        if (nan_level) {
          d = make_float4(qnan, qnan, qnan, qnan);
        } else if (from_tile) {
          d.x += (float)ldexp((double)tile[j], -kfix), d.y += (float)ldexp((double)tile[j + 1], -kfix);
          d.z += (float)ldexp((double)tile[j + 2], -kfix), d.w += (float)ldexp((double)tile[j + 3], -kfix);
        }
        *reinterpret_cast<float4*>(dst + j) = d;
        // The gradient of these entries is final here (one GPU, overwrite mode): update them in place of a separate pass -- the
        // 67 MB gradient is not read back and the stand-alone Adam launch shrinks to the MLP tail + the sparse rows.  (A slab no
        // record lands in still steps: with a zero gradient the moments decay and move the entry.)
        if (adam) {
          step4(d, pp[c], mm[c], vv[c]);
          *reinterpret_cast<float4*>(a.adam.p + slab + j) = pp[c];
          *reinterpret_cast<float4*>(a.adam.m + slab + j) = mm[c];
          *reinterpret_cast<float4*>(a.adam.v + slab + j) = vv[c];
        }
      }
    }
  }
  HG_STAMP_DRAIN(5);
  HG_STAMP_FLUSH(1, l);
}

static inline int hb_scatter_wgs(int n_levels) {  // workgroups per level of the scatter pass (a multiple of 8); UMHS_HB_WGS: measurement knob
  static const int forced = [] {
    const char* e = getenv("UMHS_HB_WGS");
    return e ? atoi(e) : 0;
  }();
  int w = forced > 0 ? forced : 1024 / (n_levels > 0 ? n_levels : 1);
  w = (w + 7) / 8 * 8;
  return w < 8 ? 8 : w;
}

static inline int hb_bucket_bits(int log2_T) { return log2_T < HB_BUCKET_BITS ? log2_T : HB_BUCKET_BITS; }

static inline size_t hb_align(size_t x) { return (x + 255) & ~(size_t)255; }
// records per level: HB_CAP_PER_SAMPLE per sample, a multiple of 8 (the level regions start on 128-byte lines)
static inline size_t hb_cap(int64_t n, size_t nwg, int nb) {
  (void)nwg, (void)nb;
  return ((size_t)n * HB_CAP_PER_SAMPLE + 7) & ~(size_t)7;
}

extern "C" size_t umhs_hashgrid_bwd_workspace_bytes(int64_t n, int n_levels, int log2_T) {
  if (n <= 0 || n_levels < 1 || log2_T < 2) return 0;
  const int nb = 1 << (log2_T - hb_bucket_bits(log2_T));
  if (nb > HB_MAX_NB) return 0;  // larger tables: only the atomic path is available
  const size_t nwg = (size_t)((n + 256 * HB_SPT - 1) / (256 * HB_SPT));
  const size_t m = (size_t)n_levels * nb, cap = hb_cap(n, nwg, nb);
  if (cap >= ((size_t)1 << 28)) return 0;  // byte offsets inside a level's record region are 32-bit (53 M samples per call)
  return 256 + hb_align((2 * m + (size_t)n_levels * HB_LMAX_PARTS) * 4) + 2 * hb_align((size_t)n_levels * nwg * nb * 4) + hb_align((size_t)n_levels * cap * 16);
}

static int hb_args(HbArgs* a, const float* pos01, const float* scalings, int64_t n, int ws_begin, int ws_levels, int log2_T,
                   void* workspace, size_t workspace_bytes);
static int hb_run_prepare(const HbArgs& a, int n_levels, int count_wgs, umhs_stream_t stream);
static int hb_run_apply(const HbArgs& a, int n_levels, float* d_table, umhs_stream_t stream);

extern "C" int umhs_hashgrid_bwd(const float* pos01, const float* d_enc, int64_t stride_n, int64_t stride_l,
                                 const float* scalings, int64_t n, int level_begin, int n_levels, int log2_T,
                                 float* d_table, int overwrite, void* workspace, size_t workspace_bytes,
                                 umhs_stream_t stream) {
  if (n < 0 || !scalings || !d_table || level_begin < 0) return UMHS_ERR_ARG;
  if (n > 0 && (!pos01 || !d_enc)) return UMHS_ERR_ARG;  // (an empty batch has no per-sample arrays)
  if (overwrite && (!workspace || n == 0)) {  // only the partitioned path writes every slot itself
    if (hipMemsetAsync(d_table + (((size_t)level_begin << log2_T) * 2), 0, ((size_t)n_levels << log2_T) * 8, umhs_s(stream)) !=
        hipSuccess)
      return UMHS_ERR_LAUNCH;
  }
  if (n_levels < 1 || level_begin + n_levels > 32 || log2_T < 2 || log2_T > 24) return UMHS_ERR_UNSUPPORTED;
  if (n == 0) return UMHS_OK;
  if (!workspace) {  // v1: memory-side float atomics (no workspace needed; fine for small N)
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)n_levels);
    hipLaunchKernelGGL(hashgrid_bwd_kernel, grid, dim3(256), 0, umhs_s(stream), pos01, d_enc, stride_n, stride_l,
                       scalings, n, log2_T, d_table, level_begin);
    UMHS_CHECK_LAUNCH();
    return UMHS_OK;
  }
  // one-call form: the histogram pass may look at the gradient too, so zero-gradient samples are skipped in both passes
  HbArgs a;
  int rc = hb_args(&a, pos01, scalings, n, level_begin, n_levels, log2_T, workspace, workspace_bytes);
  if (rc) return rc;
  if ((uintptr_t)d_table & 15) return UMHS_ERR_WORKSPACE;
  a.d_enc = d_enc, a.sn = stride_n, a.sl = stride_l, a.overwrite = overwrite, a.grad_mask = 1;
  rc = hb_run_prepare(a, n_levels, 0, stream);
  if (rc) return rc;
  return hb_run_apply(a, n_levels, d_table, stream);
}

static int hb_args(HbArgs* a, const float* pos01, const float* scalings, int64_t n, int ws_begin, int ws_levels, int log2_T,
                   void* workspace, size_t workspace_bytes) {
  const size_t need = umhs_hashgrid_bwd_workspace_bytes(n, ws_levels, log2_T);
  if (need == 0) return UMHS_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < need) return UMHS_ERR_WORKSPACE;
  a->pos01 = pos01, a->d_enc = nullptr, a->sn = 0, a->sl = 0, a->scalings = scalings, a->n = n;
  a->log2_T = log2_T, a->bucket_bits = hb_bucket_bits(log2_T), a->nb = 1 << (log2_T - a->bucket_bits), a->level0 = ws_begin;
  a->nlev = ws_levels, a->lev_off = 0, a->overwrite = 0, a->grad_mask = 0;
  a->adam = HbAdam{};
  const size_t m = (size_t)ws_levels * a->nb;
  a->nwg = (int)((n + 256 * HB_SPT - 1) / (256 * HB_SPT));
  a->cap = (uint32_t)hb_cap(n, (size_t)a->nwg, a->nb);
  uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  a->counts = reinterpret_cast<uint32_t*>(p), a->offsets = a->counts + m, a->lmax = a->offsets + m;
  p += hb_align((2 * m + (size_t)ws_levels * HB_LMAX_PARTS) * 4);
  a->wg_counts = reinterpret_cast<uint32_t*>(p);
  p += hb_align((size_t)ws_levels * a->nwg * a->nb * 4);
  a->wg_prefix = reinterpret_cast<uint32_t*>(p);
  p += hb_align((size_t)ws_levels * a->nwg * a->nb * 4);
  a->recs = reinterpret_cast<uint4*>(p);
  return UMHS_OK;
}

// histogram (count_wgs workgroups per level; 0 = one per run of samples), per-workgroup prefix, bucket scan
static int hb_run_prepare(const HbArgs& a, int n_levels, int count_wgs, umhs_stream_t stream) {
  // The histogram pass reads the gradient only in the one-call form (grad_mask): a prepare half built from hb_args() has
  // d_enc == nullptr, and a pass that dereferenced it anyway is the nil-address GPU fault recorded in DESIGN.md section 9.
  if (a.grad_mask && !a.d_enc) return UMHS_ERR_ARG;
  if (n_levels != a.nlev) return UMHS_ERR_ARG;  // (the scans walk the whole workspace range)
  dim3 pgrid((unsigned)(count_wgs > 0 && count_wgs < a.nwg ? count_wgs : a.nwg), (unsigned)n_levels);
  hipLaunchKernelGGL(hg_partition_kernel<false>, pgrid, dim3(256), 0, umhs_s(stream), a);
  hipLaunchKernelGGL(hg_wgscan_kernel, dim3((unsigned)a.nb, (unsigned)n_levels), dim3(256), 0, umhs_s(stream), a);
  hipLaunchKernelGGL(hg_scan_kernel, dim3((unsigned)n_levels), dim3(64), 0, umhs_s(stream), a);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

static int hb_run_apply(const HbArgs& a, int n_levels, float* d_table, umhs_stream_t stream) {  // scatter + bucket reduce
  if (!a.d_enc || !d_table || !a.pos01 || !a.scalings) return UMHS_ERR_ARG;  // every pointer the two kernels dereference
  // persistent workgroups: the 1024 that 256 CUs hold at four each, spread over the levels of this launch (16 levels: 64 per level =
  // 8 per XCD; a level group of 8 -- the multi-GPU exchange applies the levels in groups -- 128 per level)
  int per_level = hb_scatter_wgs(n_levels);
  if (per_level > HB_LMAX_PARTS) per_level = HB_LMAX_PARTS;  // (one max-|value| word per workgroup and level)
  if (per_level > ((a.nwg + 7) / 8) * 8) per_level = ((a.nwg + 7) / 8) * 8;
  dim3 pgrid((unsigned)per_level, (unsigned)n_levels);
  hipLaunchKernelGGL(hg_partition_kernel<true>, pgrid, dim3(256), 0, umhs_s(stream), a);
  const size_t lds = (size_t)(2 << a.bucket_bits) * 8;
  {  // raise the dynamic-LDS limit once per device, not per call (the driver call is a bubble in front of the launch)
    static std::atomic<size_t> granted[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    if (dev < 0 || granted[dev].load(std::memory_order_relaxed) < lds) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(hg_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return UMHS_ERR_LAUNCH;
      if (dev >= 0) granted[dev].store(lds, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL(hg_reduce_kernel, dim3((unsigned)a.nb, (unsigned)n_levels), dim3(1024), lds, umhs_s(stream), a, d_table);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// Gradient-independent half of the partitioned backward (bucket histogram + exclusive scan): needs only the positions, so
// a caller may run it on a side stream while the forward pass is still in flight.  (Every sample emits records in this
// form; the one-call umhs_hashgrid_bwd skips samples whose gradient is exactly zero.)
extern "C" int umhs_hashgrid_bwd_prepare(const float* pos01, const float* scalings, int64_t n, int level_begin, int n_levels,
                                         int log2_T, void* workspace, size_t workspace_bytes, umhs_stream_t stream) {
  if (n < 0 || !scalings || level_begin < 0) return UMHS_ERR_ARG;
  if (n_levels < 1 || level_begin + n_levels > 32 || log2_T < 2 || log2_T > 24) return UMHS_ERR_UNSUPPORTED;
  if (n == 0) return UMHS_OK;
  if (!pos01) return UMHS_ERR_ARG;
  HbArgs a;
  int rc = hb_args(&a, pos01, scalings, n, level_begin, n_levels, log2_T, workspace, workspace_bytes);
  if (rc) return rc;
  const int throttle = 16;  // workgroups per level of the hidden histogram pass (unthrottled it delays the forward's workgroups)
  return hb_run_prepare(a, n_levels, throttle, stream);
}

// umhs_hashgrid_fwd for ALL levels of the workspace range [0, n_levels) + the histogram pass of umhs_hashgrid_bwd_prepare for the same
// positions in one launch; umhs_hashgrid_bwd_prepare_counted then only runs the two small scans (a caller may put it on a side stream).
extern "C" int umhs_hashgrid_fwd_count(const float* pos01, const float* table, const float* scalings, int64_t n, int n_levels, int log2_T,
                                       float* enc, int64_t stride_n, int64_t stride_l, void* workspace, size_t workspace_bytes,
                                       umhs_stream_t stream) {
  if (n < 0 || !table || !scalings) return UMHS_ERR_ARG;
  if (n_levels < 1 || n_levels > 32 || log2_T < 2 || log2_T > 24) return UMHS_ERR_UNSUPPORTED;
  if (n == 0) return UMHS_OK;
  if (!pos01 || !enc) return UMHS_ERR_ARG;
  if (((uintptr_t)table & 15) || ((uintptr_t)enc & 7)) return UMHS_ERR_ARG;
  HbArgs a;
  int rc = hb_args(&a, pos01, scalings, n, 0, n_levels, log2_T, workspace, workspace_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(hashgrid_fwd_count_kernel, dim3((unsigned)a.nwg, (unsigned)n_levels), dim3(512), 0, umhs_s(stream),
                     reinterpret_cast<const float2*>(table), enc, stride_n, stride_l, a);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_hashgrid_bwd_prepare_counted(const float* pos01, const float* scalings, int64_t n, int n_levels, int log2_T,
                                                 void* workspace, size_t workspace_bytes, umhs_stream_t stream) {
  if (n < 0 || !pos01 || !scalings) return UMHS_ERR_ARG;
  if (n_levels < 1 || n_levels > 32 || log2_T < 2 || log2_T > 24) return UMHS_ERR_UNSUPPORTED;
  if (n == 0) return UMHS_OK;
  HbArgs a;
  int rc = hb_args(&a, pos01, scalings, n, 0, n_levels, log2_T, workspace, workspace_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(hg_wgscan_kernel, dim3((unsigned)a.nb, (unsigned)n_levels), dim3(256), 0, umhs_s(stream), a);
  hipLaunchKernelGGL(hg_scan_kernel, dim3((unsigned)n_levels), dim3(64), 0, umhs_s(stream), a);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// Gradient-dependent half: scatter the records of levels [level_begin, +n_levels) into their buckets and reduce every bucket
// into its d_table slab.  The workspace must hold a umhs_hashgrid_bwd_prepare of [ws_level_begin, +ws_n_levels) for the SAME
// positions, and that range must contain the levels applied; each level may be applied once per prepare.
static int hb_apply(const float* pos01, const float* d_enc, int64_t stride_n, int64_t stride_l, const float* scalings, int64_t n,
                    int level_begin, int n_levels, int ws_level_begin, int ws_n_levels, int log2_T, float* d_table, int overwrite,
                    const HbAdam* adam, void* workspace, size_t workspace_bytes, umhs_stream_t stream) {
  if (n < 0 || !scalings || !d_table || level_begin < 0) return UMHS_ERR_ARG;
  if (n > 0 && (!pos01 || !d_enc)) return UMHS_ERR_ARG;
  if (n_levels < 1 || level_begin < ws_level_begin || level_begin + n_levels > ws_level_begin + ws_n_levels ||
      ws_level_begin + ws_n_levels > 32 || log2_T < 2 || log2_T > 24)
    return UMHS_ERR_UNSUPPORTED;
  if ((uintptr_t)d_table & 15) return UMHS_ERR_WORKSPACE;
  if (n == 0) {
    if (overwrite && hipMemsetAsync(d_table + (((size_t)level_begin << log2_T) * 2), 0, ((size_t)n_levels << log2_T) * 8,
                                    umhs_s(stream)) != hipSuccess)
      return UMHS_ERR_LAUNCH;
    return UMHS_OK;
  }
  HbArgs a;
  int rc = hb_args(&a, pos01, scalings, n, ws_level_begin, ws_n_levels, log2_T, workspace, workspace_bytes);
  if (rc) return rc;
  a.d_enc = d_enc, a.sn = stride_n, a.sl = stride_l, a.lev_off = level_begin - ws_level_begin, a.overwrite = overwrite;
  if (adam) a.adam = *adam;
  return hb_run_apply(a, n_levels, d_table, stream);
}

extern "C" int umhs_hashgrid_bwd_apply(const float* pos01, const float* d_enc, int64_t stride_n, int64_t stride_l,
                                       const float* scalings, int64_t n, int level_begin, int n_levels, int ws_level_begin,
                                       int ws_n_levels, int log2_T, float* d_table, int overwrite, void* workspace,
                                       size_t workspace_bytes, umhs_stream_t stream) {
  return hb_apply(pos01, d_enc, stride_n, stride_l, scalings, n, level_begin, n_levels, ws_level_begin, ws_n_levels, log2_T, d_table,
                  overwrite, nullptr, workspace, workspace_bytes, stream);
}

// umhs_hashgrid_bwd_apply (overwrite mode) + the Adam step of the table entries of levels >= adam_level_begin in the epilogue of
// the bucket reduce, where their gradient is final: for a single-GPU trainer whose optimizer step follows the backward anyway.
// table_params / exp_avg / exp_avg_sq: [L*T,2] like d_table; hyper-parameters as umhs_adam_step (grad_scale 1).  The gradient is
// still written to d_table.  n must be > 0 (with no samples there is no reduce pass to ride on).
extern "C" int umhs_hashgrid_bwd_apply_adam(const float* pos01, const float* d_enc, int64_t stride_n, int64_t stride_l,
                                            const float* scalings, int64_t n, int level_begin, int n_levels, int ws_level_begin,
                                            int ws_n_levels, int log2_T, float* d_table, void* workspace, size_t workspace_bytes,
                                            float* table_params, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                                            float beta2, float eps, int64_t step, int adam_level_begin, umhs_stream_t stream) {
  if (!table_params || !exp_avg || !exp_avg_sq || step < 1 || adam_level_begin < 0) return UMHS_ERR_ARG;
  if (((uintptr_t)table_params | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return UMHS_ERR_ARG;
  if (n <= 0) return UMHS_ERR_UNSUPPORTED;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  HbAdam ad;
  ad.p = table_params, ad.m = exp_avg, ad.v = exp_avg_sq, ad.lr_bc1 = (float)(lr / bc1), ad.b1 = beta1, ad.b2 = beta2, ad.eps = eps;
  ad.sqrt_bc2 = (float)sqrt(bc2), ad.level_begin = adam_level_begin;
  return hb_apply(pos01, d_enc, stride_n, stride_l, scalings, n, level_begin, n_levels, ws_level_begin, ws_n_levels, log2_T, d_table,
                  1, &ad, workspace, workspace_bytes, stream);
}

// Compaction of level-major hash features: out[l][i] = in[l][idx[i]].  The sampler already encoded every candidate sample for
// its density query; the survivors' features are gathered (128 B per sample, near-sequential: idx ascends) instead of hashed
// and gathered again from the table (1 KiB per sample, random).
__global__ __launch_bounds__(256) void enc_gather_kernel(const float2* __restrict__ in, const int64_t* __restrict__ idx, int64_t m,
                                                         int64_t n, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int64_t j = idx[i];
  j = j < 0 ? 0 : (j >= m ? m - 1 : j);
  const int l = blockIdx.y;
  out[(int64_t)l * n + i] = in[(int64_t)l * m + j];
}

extern "C" int umhs_enc_gather(const float* enc_in, const int64_t* index, int64_t m, int64_t n, int n_levels, float* enc_out,
                               umhs_stream_t stream) {
  if (m < 0 || n < 0 || n_levels < 1 || n_levels > 64) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  if (m < 1 || !enc_in || !index || !enc_out || (((uintptr_t)enc_in | (uintptr_t)enc_out) & 7)) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(enc_gather_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n_levels), dim3(256), 0, umhs_s(stream),
                     reinterpret_cast<const float2*>(enc_in), index, m, n, reinterpret_cast<float2*>(enc_out));
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// =============================================================================================
// R11: pack_info  (ray_indices sorted ascending -> (start, count) per ray, by binary search)
// =============================================================================================
__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void pack_info_kernel(const int64_t* __restrict__ ray_indices, int64_t n,
                                                        int64_t n_rays, int64_t* __restrict__ packed) {
  int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  int64_t s = lower_bound_i64(ray_indices, n, r);
  int64_t e = lower_bound_i64(ray_indices, n, r + 1);
  packed[2 * r] = s;
  packed[2 * r + 1] = e - s;
}

extern "C" int umhs_pack_info(const int64_t* ray_indices, int64_t n, int64_t n_rays, int64_t* packed_info,
                              umhs_stream_t stream) {
  if (n < 0 || n_rays < 0 || !packed_info || (n > 0 && !ray_indices)) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  hipLaunchKernelGGL(pack_info_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, umhs_s(stream),
                     ray_indices, n, n_rays, packed_info);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// =============================================================================================
// R11-R13: compositing.  One wavefront per ray; lane = sample for the transmittance scan (64-lane
// shuffle prefix sum with a carry across 64-sample chunks), then lane = band for the accumulation so
// every [N,K] row is read as one coalesced run.  Accumulation order is the sample order: results are
// bitwise reproducible (the reference's index_add_ is not).
// =============================================================================================
struct CompStreams {
  int n;
  int k[UMHS_MAX_STREAMS];
  const float* v[UMHS_MAX_STREAMS];
  float* out[UMHS_MAX_STREAMS];
};

__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ sigma,
                                                            const float* __restrict__ t0,
                                                            const float* __restrict__ t1,
                                                            const int64_t* __restrict__ pinfo, int64_t n_rays,
                                                            CompStreams st, float* __restrict__ weights,
                                                            float* __restrict__ acc_out,
                                                            float* __restrict__ depth_out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (r >= n_rays) return;
  const int64_t start = pinfo[2 * r];
  const int cnt = (int)pinfo[2 * r + 1];
  float carry = 0.0f, acc = 0.0f, dnum = 0.0f;
  for (int base = 0; base < cnt || base == 0; base += 64) {
    const int i = base + lane;
    const bool valid = i < cnt;
    const int64_t nidx = start + i;
    float a = 0.0f, b = 0.0f, x = 0.0f;
    if (valid) {
      a = t0[nidx], b = t1[nidx];
      x = sigma[nidx] * (b - a);
    }
    float incl = wave_inclusive_scan(x, lane);
    float T = expf(-(carry + (incl - x)));
    float alpha = 1.0f - expf(-x);
    float w = valid ? alpha * T : 0.0f;
    if (valid) weights[nidx] = w;
    acc += w;
    dnum += w * ((a + b) / 2.0f);
    carry += __shfl(incl, 63, 64);
    const int nvalid = min(64, cnt - base);
    for (int s = 0; s < st.n;) {
      // two streams of <= 32 values share the wave (lanes 0-31 / 32-63): at the reference's 21-31 bands a single stream would leave
      // half the lanes idle, and this loop is latency-bound -- the number of row-load rounds is what it costs.  (Round 4 tried rows as
      // float4 pieces, 64 / ceil(K / 4) rows per load instruction -- 8 instead of 64 load instructions per chunk at 31 bands, partial
      // sums joined by xor-shuffles: 25.6 vs 20.5 us at C2, 189 vs 159 us with 128-band streams.  4-byte-aligned dwordx4 rows and the
      // 12 extra shuffles cost more than the load rounds they save.  Reverted.)
      const bool pair = s + 1 < st.n && st.k[s] <= 32 && st.k[s + 1] <= 32;
      const int half = pair ? (lane >> 5) : 0;
      const int K = half ? st.k[s + 1] : st.k[s];
      const float* __restrict__ v = (half ? st.v[s + 1] : st.v[s]) + (start + base) * (int64_t)K;
      float* __restrict__ outp = half ? st.out[s + 1] : st.out[s];
      const int kspan = pair ? 32 : 64, klane = pair ? (lane & 31) : lane;
      const int kmax = pair ? 32 : st.k[s];
      for (int kc = 0; kc < kmax; kc += kspan) {
        const int k = kc + klane;
        const bool kv = k < K;
        // the per-ray sum runs in sample order: 64 dependent-free row loads, 16 of them in flight per lane
        float p0 = 0.0f, p1 = 0.0f;
        const float* __restrict__ vk = v + (kv ? k : 0);
        int j = 0;
        for (; j + 15 < nvalid; j += 16) {
          float x[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) x[u] = vk[(int64_t)(j + u) * K];
#pragma unroll
          for (int u = 0; u < 16; u += 2) p0 += __shfl(w, j + u, 64) * x[u], p1 += __shfl(w, j + u + 1, 64) * x[u + 1];
        }
        for (; j + 7 < nvalid; j += 8) {  // (same association of the partial sums as ever: blocks of 8 alternate p0 / p1, the tail is p0)
          float x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = vk[(int64_t)(j + u) * K];
#pragma unroll
          for (int u = 0; u < 8; u += 2) p0 += __shfl(w, j + u, 64) * x[u], p1 += __shfl(w, j + u + 1, 64) * x[u + 1];
        }
        for (; j < nvalid; ++j) p0 += __shfl(w, j, 64) * vk[(int64_t)j * K];
        if (!kv) p0 = p1 = 0.0f;
        if (kv) {
          float* o = outp + r * K + k;
          *o = (base == 0) ? (p0 + p1) : (*o + (p0 + p1));
        }
      }
      s += pair ? 2 : 1;
    }
    if (cnt == 0) break;
  }
  acc = wave_reduce_sum(acc);
  dnum = wave_reduce_sum(dnum);
  if (lane == 0) {
    if (acc_out) acc_out[r] = acc;
    if (depth_out) depth_out[r] = dnum / (acc + 1e-10f);
  }
}

extern "C" int umhs_composite_fwd(const float* sigma, const float* t_starts, const float* t_ends,
                                  const int64_t* packed_info, int64_t n_rays, int64_t n,
                                  const umhs_value_streams* streams, float* weights, float* accumulation,
                                  float* depth, umhs_stream_t stream) {
  if (n_rays < 0 || n < 0 || !packed_info || !weights) return UMHS_ERR_ARG;
  if (n > 0 && (!sigma || !t_starts || !t_ends)) return UMHS_ERR_ARG;
  CompStreams st;
  st.n = streams ? streams->n_streams : 0;
  if (st.n < 0 || st.n > UMHS_MAX_STREAMS) return UMHS_ERR_ARG;
  for (int s = 0; s < UMHS_MAX_STREAMS; ++s) {
    st.k[s] = 0, st.v[s] = nullptr, st.out[s] = nullptr;
    if (s < st.n) {
      st.k[s] = streams->k[s], st.v[s] = streams->values[s], st.out[s] = streams->out[s];
      if (st.k[s] < 1 || !st.out[s] || (n > 0 && !st.v[s])) return UMHS_ERR_ARG;
    }
  }
  if (n_rays == 0) return UMHS_OK;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), sigma,
                     t_starts, t_ends, packed_info, n_rays, st, weights, accumulation, depth);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

struct CompGrads {
  int n;
  int k[UMHS_MAX_STREAMS];
  const float* v[UMHS_MAX_STREAMS];
  const float* dout[UMHS_MAX_STREAMS];
  float* dv[UMHS_MAX_STREAMS];
};

// w_n = alpha_n T_n,  alpha = 1-exp(-x_n),  T_n = exp(-X_n),  X_n = sum_{m<n} x_m,  x = sigma*delta
//   dL/dx_n = dw_n * T_n * exp(-x_n)  -  sum_{m>n} dw_m w_m
// pass 1 walks the ray forward and parks exp(-(X_n + x_n)) in d_sigma[n]; pass 2 walks it backward with a
// suffix scan of dw*w.  The lane<->sample mapping is identical in both passes (same-thread RAW only).
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ sigma,
                                                            const float* __restrict__ t0,
                                                            const float* __restrict__ t1,
                                                            const int64_t* __restrict__ pinfo, int64_t n_rays,
                                                            const float* __restrict__ weights, CompGrads gr,
                                                            const float* __restrict__ d_acc, int grad_scaling,
                                                            float* __restrict__ d_sigma, const float* __restrict__ dots) {
  const int lane = threadIdx.x & 63;
  __shared__ float lds_tile[4][64 * 33 + 64 + 32];  // per wave: [64][33] value tile, 64 weights, 32 upstream gradients
  float* const tile = lds_tile[threadIdx.x >> 6];
  float* const wsl = tile + 64 * 33;
  float* const dl = wsl + 64;
  const int64_t r = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (r >= n_rays) return;
  const int64_t start = pinfo[2 * r];
  const int cnt = (int)pinfo[2 * r + 1];
  if (cnt == 0) return;
  const int nchunks = (cnt + 63) >> 6;
  float carry = 0.0f;
  for (int c = 0; c < nchunks; ++c) {
    const int i = c * 64 + lane;
    const bool valid = i < cnt;
    const int64_t nidx = start + i;
    float x = valid ? sigma[nidx] * (t1[nidx] - t0[nidx]) : 0.0f;
    float incl = wave_inclusive_scan(x, lane);
    if (valid) d_sigma[nidx] = expf(-(carry + incl));
    carry += __shfl(incl, 63, 64);
  }
  const float dacc = d_acc ? d_acc[r] : 0.0f;
  float carry_after = 0.0f;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int i = c * 64 + lane;
    const bool valid = i < cnt;
    const int64_t nidx = start + i;
    float dw = 0.0f, w = 0.0f, delta = 0.0f, scale = 1.0f, tnext = 0.0f;
    if (valid) {
      float a = t0[nidx], b = t1[nidx];
      delta = b - a;
      if (grad_scaling) {
        float m = (a + b) / 2.0f;
        scale = fminf(fmaxf(m * m, 0.0f), 1.0f);
      }
      w = weights[nidx];
      tnext = d_sigma[nidx];
      dw = dacc;
      if (dots) dw += dots[nidx];  // sum_k d_out[r][k] v[n][k], formed by the caller (umhs_composite_bwd_dots)
    }
    const int nvalid = min(64, cnt - c * 64);
    // The chunk's [nvalid x K] rows are one contiguous block.  K <= 32: read it with full 256-byte wave loads into an LDS tile
    // (row stride K|1: odd, so the per-lane row walk below is bank-conflict free) instead of 64 rows x K strided dwords.  Wider
    // streams (128 / 141 bands) walk 32-band slices of the block the same way, two 128-byte row segments per wave load (the
    // row-per-lane loop they used to take ran at a third of the narrow streams' rate: 289 us at C3, 463 us at C5).
    for (int s = 0; s < gr.n; ++s) {
      const int K = gr.k[s];
      const int KS = K <= 32 ? (K | 1) : 33;
      const float* __restrict__ vb = gr.v[s] + (start + c * 64) * (int64_t)K;
      const float* __restrict__ drow = gr.dout[s] + r * (int64_t)K;
      for (int k0 = 0; k0 < K; k0 += 32) {
        const int kw = min(32, K - k0);
        if (K <= 32) {
          const int tot = nvalid * K;
          for (int e = lane; e < tot; e += 64) {
            const int jj = e / K;
            tile[jj * KS + (e - jj * K)] = vb[e];
          }
        } else {
          const int col = lane & 31;
          for (int jj = lane >> 5; jj < nvalid; jj += 2)
            if (col < kw) tile[jj * 33 + col] = vb[(int64_t)jj * K + k0 + col];
        }
        if (lane < kw) dl[lane] = drow[k0 + lane];
        if (valid) {
          float d0 = 0.0f, d1 = 0.0f;
          int k = 0;
          for (; k + 1 < kw; k += 2) {
            d0 += dl[k] * tile[lane * KS + k];
            d1 += dl[k + 1] * tile[lane * KS + k + 1];
          }
          if (k < kw) d0 += dl[k] * tile[lane * KS + k];
          dw += d0 + d1;
        }
      }
    }
    float p = dw * w;
    float suf = wave_inclusive_scan_rev(p, lane);
    float S = carry_after + (suf - p);
    if (valid) d_sigma[nidx] = (dw * tnext - S) * delta * scale;
    carry_after += __shfl(suf, 0, 64);
    // d_values[n][k] = scale_n * w_n * d_out[r][k]   (lane = band: coalesced row stores)
    const float ws = w * scale;
    wsl[lane] = ws;
    for (int s = 0; s < gr.n; ++s) {
      if (!gr.dv[s]) continue;
      const int K = gr.k[s];
      float* __restrict__ dv = gr.dv[s] + (start + c * 64) * (int64_t)K;
      const float* __restrict__ drow = gr.dout[s] + r * (int64_t)K;
      if (K <= 32) {  // the [nvalid x K] block of d_values is contiguous too: full-wave stores
        if (lane < K) dl[lane] = drow[lane];
        const int tot = nvalid * K;
        for (int e = lane; e < tot; e += 64) {
          const int jj = e / K;
          dv[e] = wsl[jj] * dl[e - jj * K];
        }
        continue;
      }
      for (int kc = 0; kc < K; kc += 64) {
        const int k = kc + lane;
        const bool kv = k < K;
        const float d = kv ? drow[k] : 0.0f;
        for (int j = 0; j < nvalid; ++j)
          if (kv) dv[(int64_t)j * K + k] = wsl[j] * d;  // 256-byte row segments; the row's weight comes from LDS
      }
    }
  }
}

extern "C" int umhs_composite_bwd(const float* sigma, const float* t_starts, const float* t_ends,
                                  const int64_t* packed_info, int64_t n_rays, int64_t n, const float* weights,
                                  const umhs_value_grads* grads, const float* d_accumulation, int grad_scaling,
                                  float* d_sigma, umhs_stream_t stream) {
  if (n_rays < 0 || n < 0 || !packed_info || !d_sigma) return UMHS_ERR_ARG;
  if (n > 0 && (!sigma || !t_starts || !t_ends || !weights)) return UMHS_ERR_ARG;
  CompGrads gr;
  gr.n = grads ? grads->n_streams : 0;
  if (gr.n < 0 || gr.n > UMHS_MAX_STREAMS) return UMHS_ERR_ARG;
  for (int s = 0; s < UMHS_MAX_STREAMS; ++s) {
    gr.k[s] = 0, gr.v[s] = nullptr, gr.dout[s] = nullptr, gr.dv[s] = nullptr;
    if (s < gr.n) {
      gr.k[s] = grads->k[s], gr.v[s] = grads->values[s], gr.dout[s] = grads->d_out[s], gr.dv[s] = grads->d_values[s];
      if (gr.k[s] < 1 || !gr.dout[s] || (n > 0 && !gr.v[s])) return UMHS_ERR_ARG;
    }
  }
  if (n_rays == 0 || n == 0) return UMHS_OK;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), sigma,
                     t_starts, t_ends, packed_info, n_rays, weights, gr, d_accumulation, grad_scaling, d_sigma, (const float*)nullptr);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// The density half of umhs_composite_bwd when the caller has already formed dots[n] = sum over streams and bands of
// d_out[ray(n)][k] * value[n][k] (the field backward does, from values it recomputes): d_sigma only, no [N,k] array read or written.
extern "C" int umhs_composite_bwd_dots(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                                       int64_t n_rays, int64_t n, const float* weights, const float* dots,
                                       const float* d_accumulation, int grad_scaling, float* d_sigma, umhs_stream_t stream) {
  if (n_rays < 0 || n < 0 || !packed_info || !d_sigma) return UMHS_ERR_ARG;
  if (n > 0 && (!sigma || !t_starts || !t_ends || !weights || !dots)) return UMHS_ERR_ARG;
  if (n_rays == 0 || n == 0) return UMHS_OK;
  CompGrads gr = {};
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), sigma, t_starts, t_ends,
                     packed_info, n_rays, weights, gr, d_accumulation, grad_scaling, d_sigma, dots);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// ---- accumulate with caller-provided weights (SpectralRenderer.forward called stand-alone) -----------------
__global__ __launch_bounds__(256) void accumulate_fwd_kernel(const float* __restrict__ weights,
                                                             const int64_t* __restrict__ pinfo, int64_t n_rays,
                                                             CompStreams st) {
  const int lane = threadIdx.x & 63;
  const int64_t r = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (r >= n_rays) return;
  const int64_t start = pinfo[2 * r];
  const int cnt = (int)pinfo[2 * r + 1];
  for (int s = 0; s < st.n; ++s) {
    const int K = st.k[s];
    for (int kc = 0; kc < K; kc += 64) {
      const int k = kc + lane;
      if (k >= K) continue;
      const float* __restrict__ v = st.v[s] + start * (int64_t)K + k;
      float p0 = 0.0f, p1 = 0.0f;
      int j = 0;
      for (; j + 1 < cnt; j += 2) {
        p0 += weights[start + j] * v[(int64_t)j * K];
        p1 += weights[start + j + 1] * v[(int64_t)(j + 1) * K];
      }
      if (j < cnt) p0 += weights[start + j] * v[(int64_t)j * K];
      st.out[s][r * K + k] = p0 + p1;
    }
  }
}

__global__ __launch_bounds__(256) void accumulate_bwd_kernel(const float* __restrict__ weights,
                                                             const int64_t* __restrict__ pinfo, int64_t n_rays,
                                                             CompGrads gr, float* __restrict__ d_weights) {
  const int lane = threadIdx.x & 63;
  const int64_t r = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (r >= n_rays) return;
  const int64_t start = pinfo[2 * r];
  const int cnt = (int)pinfo[2 * r + 1];
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    const bool valid = i < cnt;
    const int64_t nidx = start + i;
    float w = valid ? weights[nidx] : 0.0f, dw = 0.0f;
    if (valid) {
      for (int s = 0; s < gr.n; ++s) {
        const int K = gr.k[s];
        const float* __restrict__ vrow = gr.v[s] + nidx * (int64_t)K;
        const float* __restrict__ drow = gr.dout[s] + r * (int64_t)K;
        for (int k = 0; k < K; ++k) dw += drow[k] * vrow[k];
      }
      if (d_weights) d_weights[nidx] = dw;
    }
    const int nvalid = min(64, cnt - base);
    for (int s = 0; s < gr.n; ++s) {
      if (!gr.dv[s]) continue;
      const int K = gr.k[s];
      float* __restrict__ dv = gr.dv[s] + (start + base) * (int64_t)K;
      const float* __restrict__ drow = gr.dout[s] + r * (int64_t)K;
      for (int kc = 0; kc < K; kc += 64) {
        const int k = kc + lane;
        const bool kv = k < K;
        const float d = kv ? drow[k] : 0.0f;
        for (int j = 0; j < nvalid; ++j) {
          const float wj = __shfl(w, j, 64);
          if (kv) dv[(int64_t)j * K + k] = wj * d;
        }
      }
    }
  }
}

static int fill_streams(const umhs_value_streams* streams, int64_t n, CompStreams* st) {
  st->n = streams ? streams->n_streams : 0;
  if (st->n < 0 || st->n > UMHS_MAX_STREAMS) return UMHS_ERR_ARG;
  for (int s = 0; s < UMHS_MAX_STREAMS; ++s) {
    st->k[s] = 0, st->v[s] = nullptr, st->out[s] = nullptr;
    if (s < st->n) {
      st->k[s] = streams->k[s], st->v[s] = streams->values[s], st->out[s] = streams->out[s];
      if (st->k[s] < 1 || !st->out[s] || (n > 0 && !st->v[s])) return UMHS_ERR_ARG;
    }
  }
  return UMHS_OK;
}

static int fill_grads(const umhs_value_grads* grads, int64_t n, CompGrads* gr) {
  gr->n = grads ? grads->n_streams : 0;
  if (gr->n < 0 || gr->n > UMHS_MAX_STREAMS) return UMHS_ERR_ARG;
  for (int s = 0; s < UMHS_MAX_STREAMS; ++s) {
    gr->k[s] = 0, gr->v[s] = nullptr, gr->dout[s] = nullptr, gr->dv[s] = nullptr;
    if (s < gr->n) {
      gr->k[s] = grads->k[s], gr->v[s] = grads->values[s], gr->dout[s] = grads->d_out[s], gr->dv[s] = grads->d_values[s];
      if (gr->k[s] < 1 || !gr->dout[s] || (n > 0 && !gr->v[s])) return UMHS_ERR_ARG;
    }
  }
  return UMHS_OK;
}

extern "C" int umhs_accumulate_fwd(const float* weights, const int64_t* packed_info, int64_t n_rays, int64_t n,
                                   const umhs_value_streams* streams, umhs_stream_t stream) {
  if (n_rays < 0 || n < 0 || !packed_info || (n > 0 && !weights)) return UMHS_ERR_ARG;
  CompStreams st;
  int rc = fill_streams(streams, n, &st);
  if (rc) return rc;
  if (n_rays == 0 || st.n == 0) return UMHS_OK;
  hipLaunchKernelGGL(accumulate_fwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), weights,
                     packed_info, n_rays, st);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_accumulate_bwd(const float* weights, const int64_t* packed_info, int64_t n_rays, int64_t n,
                                   const umhs_value_grads* grads, float* d_weights, umhs_stream_t stream) {
  if (n_rays < 0 || n < 0 || !packed_info || (n > 0 && !weights)) return UMHS_ERR_ARG;
  CompGrads gr;
  int rc = fill_grads(grads, n, &gr);
  if (rc) return rc;
  if (n_rays == 0 || n == 0) return UMHS_OK;
  hipLaunchKernelGGL(accumulate_bwd_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), weights,
                     packed_info, n_rays, gr, d_weights);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// =============================================================================================
// R14: spectrum -> sRGB (one thread per ray; M [B,3] is tiny and stays in L1/scalar cache)
// =============================================================================================
#define GAMMA_KNEE 0.0031308f

__device__ __forceinline__ float srgb_gamma(float x) {
  return x < GAMMA_KNEE ? 12.92f * x : 1.055f * powf(fmaxf(x, 1e-6f), 1.0f / 2.4f) - 0.055f;
}

__global__ __launch_bounds__(256) void spec2rgb_fwd_kernel(const float* __restrict__ spec,
                                                           const float* __restrict__ M, int64_t n_rays, int B,
                                                           float* __restrict__ rgb) {
  int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  const float* row = spec + r * B;
  float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f;
  for (int b = 0; b < B; ++b) {
    float s = row[b];
    x0 += s * M[3 * b], x1 += s * M[3 * b + 1], x2 += s * M[3 * b + 2];
  }
  rgb[3 * r] = fminf(fmaxf(srgb_gamma(x0), 0.0f), 1.0f);
  rgb[3 * r + 1] = fminf(fmaxf(srgb_gamma(x1), 0.0f), 1.0f);
  rgb[3 * r + 2] = fminf(fmaxf(srgb_gamma(x2), 0.0f), 1.0f);
}

__device__ __forceinline__ float srgb_gamma_grad(float x) {
  // d/dx of clamp(gamma(x), 0, 1): torch passes the clamp gradient where 0 <= y <= 1
  float y = srgb_gamma(x);
  if (!(y >= 0.0f && y <= 1.0f)) return 0.0f;
  if (x < GAMMA_KNEE) return 12.92f;
  return 1.055f * (1.0f / 2.4f) * powf(fmaxf(x, 1e-6f), 1.0f / 2.4f - 1.0f);
}

__global__ __launch_bounds__(256) void spec2rgb_bwd_kernel(const float* __restrict__ spec,
                                                           const float* __restrict__ M,
                                                           const float* __restrict__ d_rgb, int64_t n_rays, int B,
                                                           float* __restrict__ d_spec, int accumulate) {
  int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  const float* row = spec + r * B;
  float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f;
  for (int b = 0; b < B; ++b) {
    float s = row[b];
    x0 += s * M[3 * b], x1 += s * M[3 * b + 1], x2 += s * M[3 * b + 2];
  }
  float g0 = d_rgb[3 * r] * srgb_gamma_grad(x0);
  float g1 = d_rgb[3 * r + 1] * srgb_gamma_grad(x1);
  float g2 = d_rgb[3 * r + 2] * srgb_gamma_grad(x2);
  float* drow = d_spec + r * B;
  for (int b = 0; b < B; ++b) {
    float v = g0 * M[3 * b] + g1 * M[3 * b + 1] + g2 * M[3 * b + 2];
    drow[b] = accumulate ? drow[b] + v : v;
  }
}

extern "C" int umhs_spec2rgb_fwd(const float* spec, const float* M, int64_t n_rays, int B, float* rgb,
                                 umhs_stream_t stream) {
  if (n_rays < 0 || B < 1 || !spec || !M || !rgb) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  hipLaunchKernelGGL(spec2rgb_fwd_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, umhs_s(stream), spec,
                     M, n_rays, B, rgb);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_spec2rgb_bwd(const float* spec, const float* M, const float* d_rgb, int64_t n_rays, int B,
                                 float* d_spec, int accumulate, umhs_stream_t stream) {
  if (n_rays < 0 || B < 1 || !spec || !M || !d_rgb || !d_spec) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  hipLaunchKernelGGL(spec2rgb_bwd_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, umhs_s(stream), spec,
                     M, d_rgb, n_rays, B, d_spec, accumulate);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// =============================================================================================
// R14-R16 fused per-ray epilogue and loss.  The reference runs ~85 tiny torch kernels per step for these
// (depth clip with a global min/max, ColourSystem, ClusterLookup + argmax + label colours, random-background
// blend, two MSE losses and their autograd); here they are one forward kernel each and the loss kernel also
// writes the gradients of both losses (the upstream gradient of a loss is a scalar, applied by the caller).
// =============================================================================================
__device__ __forceinline__ uint32_t f2ord(float f) {  // order-preserving float -> uint (for atomicMin/Max)
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// min/max of the sample mid-points t_mid = (t0+t1)/2 over the whole batch (DepthRenderer's clip bounds)
__global__ __launch_bounds__(256) void tmid_minmax_kernel(const float* __restrict__ t0, const float* __restrict__ t1, int64_t n,
                                                          uint32_t* __restrict__ mm) {
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float m = (t0[i] + t1[i]) / 2.0f;
    lo = fminf(lo, m), hi = fmaxf(hi, m);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) lo = fminf(lo, __shfl_xor(lo, d, 64)), hi = fmaxf(hi, __shfl_xor(hi, d, 64));
  __shared__ float plo[4], phi[4];
  if ((threadIdx.x & 63) == 0) plo[threadIdx.x >> 6] = lo, phi[threadIdx.x >> 6] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {  // one atomic pair per workgroup (memory-side atomics on two words serialise)
    atomicMin(&mm[0], f2ord(fminf(fminf(plo[0], plo[1]), fminf(plo[2], plo[3]))));
    atomicMax(&mm[1], f2ord(fmaxf(fmaxf(phi[0], phi[1]), fmaxf(phi[2], phi[3]))));
  }
}

__global__ void tmid_init_kernel(uint32_t* mm) { mm[0] = 0xffffffffu, mm[1] = 0u; }

extern "C" int umhs_tmid_minmax(const float* t_starts, const float* t_ends, int64_t n, float* minmax2, umhs_stream_t stream) {
  if (n < 0 || !minmax2 || (n > 0 && (!t_starts || !t_ends))) return UMHS_ERR_ARG;
  // identity of (min, max) in the ordered encoding; a 1-thread kernel, not a host-to-device copy of a stack variable (that blit
  // queued behind whatever else the device was running: 70 us on a side stream)
  hipLaunchKernelGGL(tmid_init_kernel, dim3(1), dim3(1), 0, umhs_s(stream), reinterpret_cast<uint32_t*>(minmax2));
  if (n == 0) {
    UMHS_CHECK_LAUNCH();
    return UMHS_OK;
  }
  // (one atomic pair per workgroup: 512 same-line device atomics serialise at ~12 ns each = ~6 us behind the kernel's 2 MB read, on the
  // side stream.  Fewer, longer workgroups were tried in round 4 -- 32 x 8192 elements: 63 us at C2 and 1 ms on an eval image's 18 M
  // candidates, a serial chain of loads per thread -- and reverted.)
  int64_t blocks = (n + 2047) / 2048;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(tmid_minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, umhs_s(stream), t_starts, t_ends, n,
                     reinterpret_cast<uint32_t*>(minmax2));
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// one thread per ray: rgb = ColourSystem(spectral); depth clip; ClusterLookup(alpha) against the endmembers;
// seg_raw = argmax * [acc > 0.5]; seg_pred = class colour * [acc > 0.5]
__global__ __launch_bounds__(256) void ray_epilogue_kernel(const float* __restrict__ spec, const float* __restrict__ M,
                                                           const float* __restrict__ E, const float* __restrict__ acc,
                                                           const float* __restrict__ depth_in, const uint32_t* __restrict__ mm,
                                                           const float* __restrict__ colors, int64_t n_rays, int B, int C,
                                                           float alpha, float* __restrict__ rgb, float* __restrict__ depth_out,
                                                           float* __restrict__ seg_probs, float* __restrict__ seg_raw,
                                                           float* __restrict__ seg_pred) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  const float* row = spec + r * B;
  float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f, ss = 0.0f;
  float ip[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) ip[c] = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float s = row[b];
    x0 += s * M[3 * b], x1 += s * M[3 * b + 1], x2 += s * M[3 * b + 2];
    ss += s * s;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < C) ip[c] += s * E[c * B + b];
  }
  if (rgb) {
    rgb[3 * r] = fminf(fmaxf(srgb_gamma(x0), 0.0f), 1.0f);
    rgb[3 * r + 1] = fminf(fmaxf(srgb_gamma(x1), 0.0f), 1.0f);
    rgb[3 * r + 2] = fminf(fmaxf(srgb_gamma(x2), 0.0f), 1.0f);
  }
  if (depth_out) depth_out[r] = fminf(fmaxf(depth_in[r], ord2f(mm[0])), ord2f(mm[1]));
  if (seg_probs) {
    // F.normalize: x / max(||x||, 1e-12) for the ray spectrum and for every endmember row (utils/clusterprobe.py:20-25)
    const float inv_x = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    float mx = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < C) {
        float ee = 0.0f;
        for (int b = 0; b < B; ++b) ee += E[c * B + b] * E[c * B + b];
        ip[c] = ip[c] * inv_x / fmaxf(sqrtf(ee), 1e-12f);
        if (ip[c] > mx) mx = ip[c], arg = c;
      }
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < C) ip[c] = expf(alpha * (ip[c] - mx)), sum += ip[c];
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < C) seg_probs[r * C + c] = ip[c] / sum;
    const float on = acc[r] > 0.5f ? 1.0f : 0.0f;
    if (seg_raw) seg_raw[r] = (float)arg * on;
    if (seg_pred) {
      seg_pred[3 * r] = colors[3 * arg] * on, seg_pred[3 * r + 1] = colors[3 * arg + 1] * on;
      seg_pred[3 * r + 2] = colors[3 * arg + 2] * on;
    }
  }
}

extern "C" int umhs_ray_epilogue_fwd(const float* spectral, const float* M, const float* endmembers, const float* accumulation,
                                     const float* depth, const float* tmid_minmax2, const float* class_colors, int64_t n_rays,
                                     int n_bands, int n_classes, float alpha, float* rgb, float* depth_clipped, float* seg_probs,
                                     float* seg_raw, float* seg_pred, umhs_stream_t stream) {
  if (n_rays < 0 || n_bands < 1 || !spectral || !M) return UMHS_ERR_ARG;
  if (seg_probs && (!endmembers || !accumulation || n_classes < 1)) return UMHS_ERR_ARG;
  if (seg_pred && !class_colors) return UMHS_ERR_ARG;
  if (depth_clipped && (!depth || !tmid_minmax2)) return UMHS_ERR_ARG;
  if (n_classes > 16) return UMHS_ERR_UNSUPPORTED;
  if (n_rays == 0) return UMHS_OK;
  hipLaunchKernelGGL(ray_epilogue_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, umhs_s(stream), spectral, M,
                     endmembers, accumulation, depth, reinterpret_cast<const uint32_t*>(tmid_minmax2), class_colors, n_rays,
                     n_bands, n_classes, alpha, rgb, depth_clipped, seg_probs, seg_raw, seg_pred);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// losses[0] = w_spec * mean((spec - gt_spec)^2)             (umhs_model.py:366-369)
// losses[1] = w_rgb  * mean((rgb + bg*(1-acc) - gt_rgb)^2)  (umhs_model.py:358-370, random background blend)
// Forward call: losses != NULL, d_* == NULL.  Backward call: d_* != NULL, g_up = upstream gradients of the two
// losses (device [2], NULL = 1): writes d_spec [R,B], d_rgb [R,3], d_acc [R].  One wave per ray.
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ spec, const float* __restrict__ gt_spec,
                                                   const float* __restrict__ rgb, const float* __restrict__ acc,
                                                   const float* __restrict__ bg, const float* __restrict__ gt_rgb,
                                                   int64_t n_rays, int B, float w_spec, float w_rgb,
                                                   const float* __restrict__ g_up, float* __restrict__ losses,
                                                   float* __restrict__ d_spec, float* __restrict__ d_rgb,
                                                   float* __restrict__ d_acc) {
  const int lane = threadIdx.x & 63;
  float ls = 0.0f, lr = 0.0f;
  const float gs = g_up ? g_up[0] : 1.0f, gr = g_up ? g_up[1] : 1.0f;
  const float cs = gs * w_spec * 2.0f / ((float)n_rays * (float)B);
  for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < n_rays; r0 += (int64_t)gridDim.x * 4) {  // wave-uniform trip count
    const int64_t r = r0 + (threadIdx.x >> 6);
    const bool live = r < n_rays;
    float ga = 0.0f;
    if (live) {
      for (int b = lane; b < B; b += 64) {
        const float d = spec[r * B + b] - gt_spec[r * B + b];
        ls += d * d;
        if (d_spec) d_spec[r * B + b] = cs * d;
      }
      if (rgb && lane < 3) {
        const float beta = bg ? bg[3 * r + lane] : 0.0f;
        const float d = rgb[3 * r + lane] + beta * (1.0f - acc[r]) - gt_rgb[3 * r + lane];
        lr += d * d;
        const float g = gr * w_rgb * 2.0f / ((float)n_rays * 3.0f) * d;
        if (d_rgb) d_rgb[3 * r + lane] = g;
        ga = -g * beta;  // d/d acc of beta*(1-acc)
      }
    }
    ga = wave_reduce_sum(ga);
    if (live && rgb && d_acc && lane == 0) d_acc[r] = ga;
  }
  if (!losses) return;
  ls = wave_reduce_sum(ls), lr = wave_reduce_sum(lr);
  __shared__ float part[2][4];
  if (lane == 0) part[0][threadIdx.x >> 6] = ls, part[1][threadIdx.x >> 6] = lr;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&losses[0], (part[0][0] + part[0][1] + part[0][2] + part[0][3]) * (w_spec / ((float)n_rays * (float)B)));
    if (rgb) atomicAdd(&losses[1], (part[1][0] + part[1][1] + part[1][2] + part[1][3]) * (w_rgb / ((float)n_rays * 3.0f)));
  }
}

extern "C" int umhs_loss_fwd(const float* spectral, const float* gt_spectral, const float* rgb, const float* accumulation,
                             const float* background, const float* gt_rgb, int64_t n_rays, int n_bands, float w_spectral,
                             float w_rgb, float* losses2, umhs_stream_t stream) {
  if (n_rays < 1 || n_bands < 1 || !spectral || !gt_spectral || !losses2) return UMHS_ERR_ARG;
  if (rgb && (!accumulation || !gt_rgb)) return UMHS_ERR_ARG;
  if (hipMemsetAsync(losses2, 0, 8, umhs_s(stream)) != hipSuccess) return UMHS_ERR_LAUNCH;
  hipLaunchKernelGGL(loss_kernel, dim3((unsigned)((n_rays + 3) / 4 < 256 ? (n_rays + 3) / 4 : 256)), dim3(256), 0,
                     umhs_s(stream), spectral, gt_spectral, rgb, accumulation, background, gt_rgb, n_rays, n_bands,
                     w_spectral, w_rgb, (const float*)nullptr,
                     losses2, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_loss_bwd(const float* spectral, const float* gt_spectral, const float* rgb, const float* accumulation,
                             const float* background, const float* gt_rgb, int64_t n_rays, int n_bands, float w_spectral,
                             float w_rgb, const float* grad_losses2, float* d_spectral, float* d_rgb, float* d_accumulation,
                             umhs_stream_t stream) {
  if (n_rays < 1 || n_bands < 1 || !spectral || !gt_spectral || !d_spectral) return UMHS_ERR_ARG;
  if (rgb && (!accumulation || !gt_rgb || !d_rgb || !d_accumulation)) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(loss_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), spectral, gt_spectral,
                     rgb, accumulation, background, gt_rgb, n_rays, n_bands, w_spectral, w_rgb, grad_losses2,
                     (float*)nullptr, d_spectral, d_rgb, d_accumulation);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// =============================================================================================
// Training tail of one step, fused: ray epilogue (rgb, depth clip, cluster probe) + both losses + their backward down to
// d_spectral / d_accumulation (loss_bwd with unit upstream gradients + spec2rgb_bwd).  Four launches of per-ray work on
// R = 4096 rays (16 workgroups each, ~60 us together, almost all of it latency) become one with 16 lanes per ray.
// Loss sums: per-block partials, added in block order by the last block to finish (reproducible); the arrival counter
// in `scratch` is left at zero again.
// =============================================================================================
__device__ __forceinline__ float red16(float v) {  // sum over the 16 lanes of a ray group (= a DPP row), result in every lane
  // four rotate-and-add steps on the VALU (v_add_f32_dpp row_ror:8/4/2/1); __shfl_xor compiles to ds_bpermute here, an LDS round
  // trip per step in kernels that are nothing but chains of such reductions
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
  return v;
}

struct TailArgs {
  const float *spec, *M, *E, *acc, *depth, *colors, *gt_spec, *gt_rgb, *bg;
  const uint32_t* mm;
  int64_t n_rays;
  int B, C, rgb_loss;
  float alpha, w_spec, w_rgb;
  float *rgb, *depth_out, *seg_probs, *seg_raw, *seg_pred, *losses, *d_spec, *d_acc;
  float* partial;     // [gridDim.x][2]
  uint32_t* counter;  // zero on entry, zero on exit
};

__global__ __launch_bounds__(256) void ray_train_tail_kernel(TailArgs a) {
  __shared__ float ee_inv[16];
  __shared__ float part[2][16];
  __shared__ bool last;
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4;
  const int B = a.B, C = a.C;
  {  // 1 / max(||E_c||, 1e-12): F.normalize of the endmember rows (clusterprobe.py:20-25), once per block, 16 lanes per class
    float ee = 0.0f;
    if (grp < C)
      for (int b = l; b < B; b += 16) ee += a.E[grp * B + b] * a.E[grp * B + b];
    ee = red16(ee);
    if (l == 0) ee_inv[grp] = 1.0f / fmaxf(sqrtf(ee), 1e-12f);
  }
  __syncthreads();
  const float cs = a.w_spec * 2.0f / ((float)a.n_rays * (float)B);
  const float cr = a.w_rgb * 2.0f / ((float)a.n_rays * 3.0f);
  const float tlo = ord2f(a.mm[0]), thi = ord2f(a.mm[1]);
  float ls = 0.0f, lr = 0.0f;
  for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < a.n_rays; r0 += (int64_t)gridDim.x * 16) {  // block-uniform trip count
    const int64_t r = r0 + grp;
    const bool live = r < a.n_rays;
    const int64_t rc = live ? r : a.n_rays - 1;
    const float* row = a.spec + rc * B;
    const float* grow = a.gt_spec + rc * B;
    float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f, ss = 0.0f, dl = 0.0f;
    float ip[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) ip[c] = 0.0f;
    for (int b = l; b < B; b += 16) {
      const float s = row[b], d = s - grow[b];
      x0 += s * a.M[3 * b], x1 += s * a.M[3 * b + 1], x2 += s * a.M[3 * b + 2];
      ss += s * s, dl += d * d;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < C) ip[c] += s * a.E[c * B + b];
    }
    x0 = red16(x0), x1 = red16(x1), x2 = red16(x2), ss = red16(ss);
    if (live) ls += dl;  // per-lane partial; reduced once at the end
    const float x[3] = {x0, x1, x2};
    float rgbv[3], g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) rgbv[k] = fminf(fmaxf(srgb_gamma(x[k]), 0.0f), 1.0f);
    const float accv = a.acc[rc];
    if (a.rgb_loss) {
      float ga = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float beta = a.bg ? a.bg[3 * rc + k] : 0.0f;
        const float d = rgbv[k] + beta * (1.0f - accv) - a.gt_rgb[3 * rc + k];
        if (live && l == 0) lr += d * d;
        const float gk = cr * d;
        ga -= gk * beta;
        g[k] = gk * srgb_gamma_grad(x[k]);
      }
      if (live && l == 0 && a.d_acc) a.d_acc[r] = ga;
    }
    if (live) {
      float* drow = a.d_spec + r * B;
      for (int b = l; b < B; b += 16)
        drow[b] = cs * (row[b] - grow[b]) + (g[0] * a.M[3 * b] + g[1] * a.M[3 * b + 1] + g[2] * a.M[3 * b + 2]);
      if (l < 3 && a.rgb) a.rgb[3 * r + l] = rgbv[l];
      if (l == 3 && a.depth_out) a.depth_out[r] = fminf(fmaxf(a.depth[r], tlo), thi);
    }
    if (a.seg_probs) {
      const float inv_x = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
      float mx = -INFINITY, mine = 0.0f;
      int arg = 0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c < C) {
          const float v = red16(ip[c]) * inv_x * ee_inv[c];
          if (v > mx) mx = v, arg = c;
          if (c == l) mine = v;
        }
      }
      const float e = l < C ? expf(a.alpha * (mine - mx)) : 0.0f;
      const float sum = red16(e);
      if (live) {
        if (l < C) a.seg_probs[r * C + l] = e / sum;
        const float on = accv > 0.5f ? 1.0f : 0.0f;
        if (l == 0 && a.seg_raw) a.seg_raw[r] = (float)arg * on;
        if (l < 3 && a.seg_pred) a.seg_pred[3 * r + l] = a.colors[3 * arg + l] * on;
      }
    }
  }
  ls = red16(ls), lr = red16(lr);
  if (l == 0) part[0][grp] = ls, part[1][grp] = lr;
  __syncthreads();
  if (tid == 0) {
    float s0 = 0.0f, s1 = 0.0f;
    for (int i = 0; i < 16; ++i) s0 += part[0][i], s1 += part[1][i];
    a.partial[2 * blockIdx.x] = s0, a.partial[2 * blockIdx.x + 1] = s1;
    __threadfence();
    last = atomicAdd(a.counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {  // block-wide tree over the (<= 256) partials: fixed order, no serial chain of L2 round trips
    __threadfence();
    float s0 = 0.0f, s1 = 0.0f;
    for (unsigned i = tid; i < gridDim.x; i += 256) s0 += a.partial[2 * i], s1 += a.partial[2 * i + 1];  // (<= 1024 partials)
    s0 = red16(s0), s1 = red16(s1);
    __syncthreads();
    if (l == 0) part[0][grp] = s0, part[1][grp] = s1;
    __syncthreads();
    if (tid == 0) {
      s0 = 0.0f, s1 = 0.0f;
      for (int i = 0; i < 16; ++i) s0 += part[0][i], s1 += part[1][i];
      a.losses[0] = s0 * (a.w_spec / ((float)a.n_rays * (float)B));
      a.losses[1] = a.rgb_loss ? s1 * (a.w_rgb / ((float)a.n_rays * 3.0f)) : 0.0f;
      *a.counter = 0u;
    }
  }
}

constexpr int TAIL_MAX_BLOCKS = 1024;  // 16 rays per workgroup iteration: up to 16 k rays get a workgroup each (4 resident per CU)
extern "C" size_t umhs_ray_train_tail_scratch_bytes(void) { return TAIL_MAX_BLOCKS * 2 * sizeof(float) + 64; }

extern "C" int umhs_ray_train_tail(const float* spectral, const float* M, const float* endmembers, const float* accumulation,
                                   const float* depth, const float* tmid_minmax2, const float* class_colors,
                                   const float* gt_spectral, const float* gt_rgb, const float* background, int64_t n_rays,
                                   int n_bands, int n_classes, float alpha, float w_spectral, float w_rgb, int rgb_loss,
                                   float* rgb, float* depth_clipped, float* seg_probs, float* seg_raw, float* seg_pred,
                                   float* losses2, float* d_spectral, float* d_accumulation, void* scratch,
                                   size_t scratch_bytes, umhs_stream_t stream) {
  if (n_rays < 1 || n_bands < 1 || !spectral || !M || !gt_spectral || !accumulation || !tmid_minmax2 || !losses2 || !d_spectral ||
      !scratch)
    return UMHS_ERR_ARG;
  if (rgb_loss && (!gt_rgb || !d_accumulation)) return UMHS_ERR_ARG;
  if (depth_clipped && !depth) return UMHS_ERR_ARG;
  if (seg_probs && (!endmembers || n_classes < 1)) return UMHS_ERR_ARG;
  if (seg_pred && (!class_colors || !seg_probs)) return UMHS_ERR_ARG;
  if (n_classes > 16) return UMHS_ERR_UNSUPPORTED;
  if (scratch_bytes < umhs_ray_train_tail_scratch_bytes() || ((uintptr_t)scratch & 3)) return UMHS_ERR_WORKSPACE;
  TailArgs a;
  a.spec = spectral, a.M = M, a.E = endmembers, a.acc = accumulation, a.depth = depth, a.colors = class_colors;
  a.gt_spec = gt_spectral, a.gt_rgb = gt_rgb, a.bg = background, a.mm = reinterpret_cast<const uint32_t*>(tmid_minmax2);
  a.n_rays = n_rays, a.B = n_bands, a.C = seg_probs ? n_classes : 0, a.rgb_loss = rgb_loss, a.alpha = alpha;
  a.w_spec = w_spectral, a.w_rgb = w_rgb, a.rgb = rgb, a.depth_out = depth_clipped, a.seg_probs = seg_probs;
  a.seg_raw = seg_raw, a.seg_pred = seg_pred, a.losses = losses2, a.d_spec = d_spectral, a.d_acc = d_accumulation;
  a.counter = reinterpret_cast<uint32_t*>(scratch), a.partial = reinterpret_cast<float*>(scratch) + 16;
  const int64_t blocks = (n_rays + 15) / 16;
  hipLaunchKernelGGL(ray_train_tail_kernel, dim3((unsigned)(blocks < TAIL_MAX_BLOCKS ? blocks : TAIL_MAX_BLOCKS)), dim3(256), 0,
                     umhs_s(stream), a);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// (The per-ray part of a training step as ONE launch -- compositing forward + training tail + compositing backward, umhs_ray_train_fused --
// was built in round 2, equal to the three kernels bit for bit, and lost its A/B twice: 0.966 vs 0.934 ms per step at C2, each of the
// three being one latency chain per ray already.  Removed in round 3.)

// =============================================================================================
// Fused Adam over the flat "fields" parameter buffer (28 B/param of pure HBM streaming, float4 lanes)
// =============================================================================================
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_bc1,
                                                   float b1, float b2, float eps, float sqrt_bc2, float gscale,
                                                   int64_t cb, int64_t ce) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      float4 pp = *reinterpret_cast<float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
      float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      float* pa = reinterpret_cast<float*>(&pp);
      float* ga = reinterpret_cast<float*>(&gg);
      float* ma = reinterpret_cast<float*>(&mm);
      float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        adam_update(pa[k], ma[k], va[k], ga[k] * gscale, lr_bc1, b1, b2, eps, sqrt_bc2);
        if (i + k >= cb && i + k < ce) pa[k] = fminf(fmaxf(pa[k], 0.0f), 1.0f);
      }
      *reinterpret_cast<float4*>(p + i) = pp;
      *reinterpret_cast<float4*>(m + i) = mm;
      *reinterpret_cast<float4*>(v + i) = vv;
    } else {
      for (int64_t j = i; j < n; ++j) {
        float mk = m[j], vk = v[j], pk = p[j];
        adam_update(pk, mk, vk, g[j] * gscale, lr_bc1, b1, b2, eps, sqrt_bc2);
        if (j >= cb && j < ce) pk = fminf(fmaxf(pk, 0.0f), 1.0f);
        p[j] = pk, m[j] = mk, v[j] = vk;
      }
    }
  }
}

// Adam on selected 2-float rows only (the live rows of the sparse coarse hash levels: every other row of those levels has
// g = m = v = 0 for ever, so its update is exactly zero and it is not touched).  Same arithmetic as adam_kernel.
__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const int64_t* __restrict__ rows, int64_t n_rows,
                                                        float lr_bc1, float b1, float b2, float eps, float sqrt_bc2, float gscale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  const int64_t o = rows[i] * 2;
  float2 pp = *reinterpret_cast<float2*>(p + o), mm = *reinterpret_cast<float2*>(m + o), vv = *reinterpret_cast<float2*>(v + o);
  const float2 gg = *reinterpret_cast<const float2*>(g + o);
  float* pa = reinterpret_cast<float*>(&pp);
  float* ma = reinterpret_cast<float*>(&mm);
  float* va = reinterpret_cast<float*>(&vv);
  const float ga[2] = {gg.x, gg.y};
#pragma unroll
  for (int k = 0; k < 2; ++k) adam_update(pa[k], ma[k], va[k], ga[k] * gscale, lr_bc1, b1, b2, eps, sqrt_bc2);
  *reinterpret_cast<float2*>(p + o) = pp;
  *reinterpret_cast<float2*>(m + o) = mm;
  *reinterpret_cast<float2*>(v + o) = vv;
}

extern "C" int umhs_adam_step_rows(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* rows,
                                   int64_t n_rows, float lr, float beta1, float beta2, float eps, int64_t step,
                                   float grad_scale, umhs_stream_t stream) {
  if (n_rows < 0 || step < 1 || !params || !grads || !exp_avg || !exp_avg_sq || (n_rows > 0 && !rows)) return UMHS_ERR_ARG;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 7) return UMHS_ERR_ARG;
  if (n_rows == 0) return UMHS_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, umhs_s(stream), params, grads,
                     exp_avg, exp_avg_sq, rows, n_rows, (float)(lr / bc1), beta1, beta2, eps, (float)sqrt(bc2), grad_scale);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// adam_rows_kernel and adam_kernel in one launch: the first row_blocks workgroups take the rows, the others the dense range
// [t0, t0 + tn) (what is left for the optimizer when the dense hash levels were updated inside the backward: the live rows of the
// coarse levels and the MLP / endmember tail -- two ~6 us launches at the very end of the step).
__global__ __launch_bounds__(256) void adam_rows_range_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, const int64_t* __restrict__ rows, int64_t n_rows,
                                                              int row_blocks, int64_t t0, int64_t tn, float lr_bc1, float b1, float b2,
                                                              float eps, float sqrt_bc2, float gscale, int64_t cb, int64_t ce) {
  if ((int)blockIdx.x < row_blocks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const int64_t o = rows[i] * 2;
    float2 pp = *reinterpret_cast<float2*>(p + o), mm = *reinterpret_cast<float2*>(m + o), vv = *reinterpret_cast<float2*>(v + o);
    const float2 gg = *reinterpret_cast<const float2*>(g + o);
    adam_update(pp.x, mm.x, vv.x, gg.x * gscale, lr_bc1, b1, b2, eps, sqrt_bc2);
    adam_update(pp.y, mm.y, vv.y, gg.y * gscale, lr_bc1, b1, b2, eps, sqrt_bc2);
    *reinterpret_cast<float2*>(p + o) = pp;
    *reinterpret_cast<float2*>(m + o) = mm;
    *reinterpret_cast<float2*>(v + o) = vv;
    return;
  }
  const int64_t nblk = (int64_t)gridDim.x - row_blocks;
  for (int64_t j = (((int64_t)blockIdx.x - row_blocks) * 256 + threadIdx.x); j < tn; j += nblk * 256) {
    const int64_t e = t0 + j;
    float pk = p[e], mk = m[e], vk = v[e];
    adam_update(pk, mk, vk, g[e] * gscale, lr_bc1, b1, b2, eps, sqrt_bc2);
    if (e >= cb && e < ce) pk = fminf(fmaxf(pk, 0.0f), 1.0f);
    p[e] = pk, m[e] = mk, v[e] = vk;
  }
}

extern "C" int umhs_adam_step_rows_range(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* rows,
                                         int64_t n_rows, int64_t range_begin, int64_t range_count, float lr, float beta1, float beta2,
                                         float eps, int64_t step, float grad_scale, int64_t clamp_begin, int64_t clamp_end,
                                         umhs_stream_t stream) {
  if (n_rows < 0 || range_begin < 0 || range_count < 0 || step < 1 || !params || !grads || !exp_avg || !exp_avg_sq ||
      (n_rows > 0 && !rows))
    return UMHS_ERR_ARG;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 7) return UMHS_ERR_ARG;
  if (n_rows == 0 && range_count == 0) return UMHS_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const int64_t row_blocks = (n_rows + 255) / 256;
  int64_t range_blocks = (range_count + 255) / 256;
  if (range_blocks > 1024) range_blocks = 1024;
  if (row_blocks + range_blocks > 0x7fffffff) return UMHS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(adam_rows_range_kernel, dim3((unsigned)(row_blocks + range_blocks)), dim3(256), 0, umhs_s(stream), params, grads,
                     exp_avg, exp_avg_sq, rows, n_rows, (int)row_blocks, range_begin, range_count, (float)(lr / bc1), beta1, beta2, eps,
                     (float)sqrt(bc2), grad_scale, clamp_begin, clamp_end);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                              float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale,
                              int64_t clamp_begin, int64_t clamp_end, umhs_stream_t stream) {
  if (n < 0 || step < 1 || !params || !grads || !exp_avg || !exp_avg_sq) return UMHS_ERR_ARG;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, umhs_s(stream), params, grads, exp_avg,
                     exp_avg_sq, n, (float)(lr / bc1), beta1, beta2, eps, (float)sqrt(bc2), grad_scale,
                     clamp_begin, clamp_end);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// =============================================================================================
extern "C" const char* umhs_strerror(int code) {
  switch (code) {
    case UMHS_OK: return "ok";
    case UMHS_ERR_ARG: return "invalid argument";
    case UMHS_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case UMHS_ERR_WORKSPACE: return "workspace missing or too small";
    case UMHS_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
  }
}
extern "C" int umhs_abi_version(void) { return UMHS_ABI_VERSION; }
