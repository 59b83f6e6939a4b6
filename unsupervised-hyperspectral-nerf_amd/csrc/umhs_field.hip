// Fused per-sample UMHS field for gfx950 (R3-R9, R18): mlp_base MLP, NeRF/SH encodings, mlp_head,
// feature_mlp, mlp_directional, sigmoid / temperature-softmax and endmember mixing, forward and
// backward, on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact f32 fmaf chain, needed for the 1e-4
// radiance parity).  Reference: umhs_field.py:151-261,300-329.
//
// Data flow ("samples on lanes"): every GEMM is computed transposed, Y^T[out][sample] = W[out][in] X^T,
// with the WEIGHTS as the MFMA A operand and the ACTIVATIONS as the B operand.  A 16x16 result tile
// then has its 16 samples on lane&15 and its 16 output features on (lane>>4, reg) -- which is exactly
// the B-operand shape of the next layer (k-slot <-> lane>>4), so an accumulator register feeds the
// next MFMA directly: no LDS transpose, no cross-lane traffic between layers.  The price is a permuted
// k order, paid once by packing each weight matrix in the matching order (fwd image in LDS; the
// transposed images for dX come from global/L2).  Bias rides in as the initial accumulator.
//
// Backward recomputes the forward per 16-sample tile (saved: hash features, sigma_raw, emb, feature logits), runs the dX chain the
// same way with transposed packs, and forms dW = dZ X^T (contraction over samples, i.e. over lanes) from tiles transposed by an
// identity MFMA, as three bf16 products into accumulators every wave keeps in AGPRs for the whole launch ("transpose-free": no LDS
// staging); per-workgroup slabs are folded and summed by two small kernels.  Two main kernels: part 0 = head MLP + directional MLP +
// mixing, part 1 = feature MLP + mlp_base; their instruction schedule is in umhs_field_zip.h / umhs_zip_plan.h.
//
// Translation units: this file compiles four times side by side (umhsnerf/build.py): -DUMHS_FIELD_TU=0 the forward, the small
// kernels and the host side; =1 the zipped part-0 backward kernels; =2 the part-1 backward kernels; =3 the fp32-chain part-0 kernels
// (the long poles of the build).  Without the define (tools/stamp_fbwd.py, tools/build_alt.sh) everything is one unit.
#include <cstdlib>

#include "umhs_common.h"
#include <atomic>

#ifndef UMHS_FIELD_TU
#define UMHS_TU_MAIN 1
#define UMHS_TU_P0Z 1
#define UMHS_TU_P0F 1
#define UMHS_TU_P1 1
#else
#define UMHS_TU_MAIN (UMHS_FIELD_TU == 0)
#define UMHS_TU_P0Z (UMHS_FIELD_TU == 1)
#define UMHS_TU_P0F (UMHS_FIELD_TU == 3)
#define UMHS_TU_P1 (UMHS_FIELD_TU == 2)
#if defined(UMHS_TF_STAMP)
#error "the stamped build is a single translation unit (the stamp buffer is a __device__ variable)"
#endif
#endif

typedef float v4f __attribute__((ext_vector_type(4)));

// timing-only ablation builds (tools/ablate_fwd.sh) define UMHS_ABL_*; never defined in the shipped library
#ifdef UMHS_ABL_NO_MFMA
__device__ __forceinline__ v4f mfma_stub(float a, float b, v4f c) { c[0] += a * b; return c; }
#define MFMA(a, b, c) mfma_stub((a), (b), (c))
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif
#ifdef UMHS_ABL_NO_SYNC  // timing-only ablation build (tools/ablate_field.sh); never defined in the shipped library
#define BSYNC()
#else
#define BSYNC() __syncthreads()
#endif

enum InKind { IN_ENC = 0, IN_HID64 = 1, IN_HID16 = 2, IN_27 = 3, IN_DIR28 = 4, IN_MIX = 5 };
enum LayerId { L_B0 = 0, L_B1, L_H0, L_H1, L_H2, L_F0, L_F1, L_F2, L_D0, L_D1, L_MX, NLAYERS };

struct LayerDesc {
  const float* W;  // [OUT][IN] row-major (L_MX: endmembers [C][B], addressed transposed)
  const float* b;  // [OUT] or null
  int kind, KS, OT, OUT, IN;
  int off_w, off_b;  // float offsets in the forward pack image
};
struct PackDesc {
  LayerDesc L[NLAYERS];
  int total_w, total;  // floats
};

// k-slot (step s, lane quarter q) -> column of the reference weight matrix, or -1
__device__ __forceinline__ int kmap_in(int kind, int s, int q) {
  switch (kind) {
    case IN_ENC: return 8 * q + s;
    case IN_HID64: return 16 * (s >> 2) + 4 * q + (s & 3);
    case IN_HID16: return 4 * q + s;
    case IN_27: {
      if (s < 3) return 3 * q + s;            // positional encoding p = 3q+s
      int e = 4 * q + (s - 3) - 1;            // base-MLP output slot 4q+r, slot 0 is sigma_raw
      return e >= 0 ? 12 + e : -1;
    }
    case IN_DIR28: return s < 4 ? 4 * q + s : 16 + 3 * q + (s - 4);
    default: return 4 * q + s;  // IN_MIX: class index
  }
}

// forward pack image: for each layer, A-operand values in the exact order the waves consume them:
//   w[off_w + ((t*KS4 + s4)*64 + lane)*4 + ss] = W[16t + (lane&15)][kmap(4*s4+ss, lane>>4)]
__device__ __forceinline__ float fwd_pack_value(const PackDesc& pd, int idx) {
  int li = 0;
  while (li + 1 < NLAYERS && idx >= pd.L[li + 1].off_w) ++li;
  const LayerDesc& L = pd.L[li];
  const int rel = idx - L.off_w;
  const int ss = rel & 3, ln = (rel >> 2) & 63, blk = rel >> 8;
  const int KS4 = (L.KS + 3) >> 2;
  const int t = blk / KS4, s = (blk % KS4) * 4 + ss;
  const int out = 16 * t + (ln & 15), q = ln >> 4;
  if (s >= L.KS || out >= L.OUT) return 0.0f;
  const int in = kmap_in(L.kind, s, q);
  if (li == L_MX) return (in >= 0 && in < L.IN) ? L.W[(size_t)in * L.OUT + out] : 0.0f;  // E[c][b]
  return (in >= 0 && in < L.IN) ? L.W[(size_t)out * L.IN + in] : 0.0f;
}

__device__ __forceinline__ void build_fwd_image(float* lds, const PackDesc& pd) {
  for (int idx = threadIdx.x; idx < pd.total_w; idx += blockDim.x) lds[idx] = fwd_pack_value(pd, idx);
  for (int li = 0; li < NLAYERS; ++li) {
    const LayerDesc& L = pd.L[li];
    if (li == L_MX) continue;
    for (int o = threadIdx.x; o < 16 * L.OT; o += blockDim.x) lds[L.off_b + o] = (L.b && o < L.OUT) ? L.b[o] : 0.0f;
  }
}

// LDS image = image[first .. total): copy when a prebuilt image is given, else gather-build in place
__device__ __forceinline__ void load_fwd_image(float* lds, const PackDesc& pd, const float* __restrict__ image, int first) {
  if (image) {
    const int n4 = (pd.total - first + 3) >> 2;  // first and the image buffer are 16-byte aligned
    for (int i = threadIdx.x; i < n4; i += blockDim.x)
      reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(image + first)[i];
  } else {
    build_fwd_image(lds - first, pd);
  }
}

// acc[ct][t] (+)= W-pack(t, :) x B-operand regs b[ct][:]   (A from LDS or global, 16 B per lane per 4 k-steps)
// INIT: 0 = accumulate into acc, 1 = start from zero, 2 = start from the bias (compile-time: a runtime `if (bias)` is a
// real branch -- LDS address 0 is valid -- and every branch ends a scheduling region, pinning the operand loads to
// their gemm instead of letting them be hoisted over the previous one)
// Software pipeline: the A fragments (one ds_read_b128 = 4 k-steps of one output tile) are consumed in bundles of G; the
// next bundle's reads are issued BEFORE the current bundle's MFMAs (the compiler on its own emits read -> s_waitcnt
// lgkmcnt(0) -> MFMAs, exposing the LDS latency once per fragment), and inside a bundle the MFMAs alternate between
// >= 2 accumulators so that none waits on its predecessor (32-cycle issue vs 40-cycle dependent issue).
// SWAP: operands exchanged -> the TRANSPOSED tile D[sample 4q+r][feature lane&15] (the same pack image serves: lane l holds
// W[out = l&15][in = l>>4] either way); used for the last layers so that output rows are written 16 consecutive floats
// per quarter-wave instead of one float per row.
template <int OT, int KS, int NT, int INIT, bool SWAP = false>
__device__ __forceinline__ void gemm_pack(v4f (&acc)[NT][OT], const float (&b)[NT][KS], const float* __restrict__ w,
                                          const float* __restrict__ bias, int lane) {
  constexpr int KS4 = (KS + 3) / 4;
  constexpr int NF = KS4 * OT;                 // fragment f = s4 * OT + t
  constexpr bool SPLIT = (OT == 1 && NT == 1 && KS4 >= 2);  // one tile, one column block: split K over two accumulators
  constexpr int G = (NT >= 2) ? 1 : 2;
  constexpr int NBUN = (NF + G - 1) / G;
  if (INIT != 0) {
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      v4f bv = {0.0f, 0.0f, 0.0f, 0.0f};
      if (INIT == 2 && !SWAP) bv = *reinterpret_cast<const v4f*>(bias + 16 * t + 4 * (lane >> 4));
      if (INIT == 2 && SWAP) {
        const float bj = bias[16 * t + (lane & 15)];
        bv = v4f{bj, bj, bj, bj};
      }
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct][t] = bv;
    }
  }
  v4f acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
  v4f a[2][G];
#pragma unroll
  for (int g = 0; g < G; ++g)
    if (g < NF) a[0][g] = *reinterpret_cast<const v4f*>(w + (((g % OT) * KS4 + g / OT) * 64 + lane) * 4);
#pragma unroll
  for (int bun = 0; bun < NBUN; ++bun) {
    if (bun + 1 < NBUN) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int f = (bun + 1) * G + g;
        if (f < NF) a[(bun + 1) & 1][g] = *reinterpret_cast<const v4f*>(w + (((f % OT) * KS4 + f / OT) * 64 + lane) * 4);
      }
    }
    __builtin_amdgcn_sched_barrier(0x7ff & ~0x180);  // everything but LDS reads may move across: the prefetch stays ahead
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int f = bun * G + g, s4 = f / OT, t = f % OT;
        if (f < NF && s4 * 4 + ss < KS) {
          if (SPLIT && (f & 1)) {
            acc2 = SWAP ? MFMA(b[0][s4 * 4 + ss], a[bun & 1][g][ss], acc2) : MFMA(a[bun & 1][g][ss], b[0][s4 * 4 + ss], acc2);
          } else {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
              acc[ct][t] = SWAP ? MFMA(b[ct][s4 * 4 + ss], a[bun & 1][g][ss], acc[ct][t])
                                : MFMA(a[bun & 1][g][ss], b[ct][s4 * 4 + ss], acc[ct][t]);
          }
        }
      }
    }
  }
  if (SPLIT) acc[0][0] += acc2;
}

// relu as ONE integer max on the bit pattern (negative floats are negative ints; +NaN stays NaN like torch.relu);
// fmaxf() on an MFMA result costs two instructions because hipcc first canonicalises a possible sNaN
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

template <int OT, int NT>
__device__ __forceinline__ void relu_to(float (&x)[NT][OT * 4], const v4f (&acc)[NT][OT]) {
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[ct][4 * t + r] = relu1(acc[ct][t][r]);
}

// exp via v_exp_f32 (2^x) and reciprocal via v_rcp_f32: ~1e-7..1e-6 relative error for the |x| <~ 30 seen here, an
// order of magnitude inside the parity budget, and ~10x fewer instructions than the IEEE sequences between MFMAs
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return frcp(1.0f + fexp(-x)); }
// Reductions over the 4 lane quarters (lanes l, l^16, l^32, l^48).  gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd
// 16-lane rows (resp. the upper 32 lanes) of one operand with the even rows (lower half) of the other: called with v for both,
// the two results are v's even-row and odd-row (lower / upper half) copies, i.e. {v, v from the partner quarter} in every lane --
// pure VALU, where __shfl_xor compiles to ds_bpermute_b32 and pays an LDS round trip (9 of them per tile of the backward, with
// nothing else on the SIMD to cover them).  Same values, same association as the shuffles they replace.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xq_max(float v) {
  v2u a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xq_sum(float v) {
  v2u a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
// sum over the 16 lanes of a DPP row (the lanes that share a quarter q), result in every lane: four rotate-and-add steps on the
// VALU (row_ror 8 / 4 / 2 / 1) -- __shfl_xor compiles to ds_bpermute here, ~100 cycles of LDS latency per step that a kernel at one
// wave per SIMD cannot hide (part 0 of the backward: +14 us at C2 with four of those per value)
__device__ __forceinline__ float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}
__device__ __forceinline__ float sel4(const v4f& v, int r) { return r == 0 ? v[0] : (r == 1 ? v[1] : (r == 2 ? v[2] : v[3])); }

struct FieldIO {
  const float* enc;
  int64_t sn, sl;
  const float *wpos, *dirs, *sel;
  int64_t n;
  int B, C, TB;
  float temperature;
  // forward outputs
  float *sigma, *sigma_raw, *emb, *spectral, *spectral2, *specular, *abund;
  // backward
  const float *d_sigma, *d_spectral, *d_emb;
  float* d_enc;
  const float *emb_in, *sigma_raw_in;  // saved forward outputs (heads / base backward)
  float* d_bo;                         // [N,16] gradient w.r.t. the base MLP's outputs (heads -> base)
  float* feat_logits;                  // forward: optional [N,16] feature_mlp logits (rows 0..C), saved for the split backward
  const float* feat_logits_in;         // split backward, part 0
  float* d_fl;                         // [N,16] gradient w.r.t. the feature logits (part 0 -> part 1)
  float* d_bo2;                        // [N,16] part 1's share of d_bo (base kernel adds the two)
  // heads-only forward with the per-ray band sums taken inside the kernel (field_fwd_kernel<.., HEADS = true>)
  const float* weights;                // [N] rendering weights of the samples
  const int64_t* ray_of;               // [N] ray of each sample (non-decreasing)
  float* part;                         // per-16-sample-tile partial sums, see HeadsComp
  float* comp[3];                      // [R,B] band sums of spectral / spectral2 / specular (rays inside one tile are written here directly)
  int n_streams;                       // 1 without the specular head, else 3
  float* part_m;                       // w m (mixing input): per-tile partials [(g*2 + az)*16 + c]
  float* mix16;                        // [R,16] per-ray sums of w m (written here for rays strictly inside a tile, else by the finish pass)
  float* part_ab;                      // abundances: per-tile partials [(g*2 + az)*16 + c]
  float* comp_ab;                      // [R,C] per-ray abundance sums (or null)
  float* bo16;                         // density half: the base MLP's 16 outputs as aligned rows [N,16] (slot 0 = sigma_raw), or null
  const float* bo16_in;                // heads forward / backward part 0: read emb from such rows instead of [N,15]
  // transpose-free backward, part 0 with the compositing backward's value half folded in (FUSED): d_spectral[n][b] =
  // scale_n * weights[n] * d_comp[ray(n)][b] is formed on the fly, and dots[n] = sum_b d_comp[ray(n)][b] * spectral[n][b] goes out for
  // umhs_composite_bwd_dots (mixing half: sum_c m[c] (d_comp E^T)[c], which the kernel's d m accumulator already is; specular half
  // from the sigmoids it computes anyway)
  const float* d_comp;                 // [R,B] gradient w.r.t. the per-ray band sums of spectral
  const float* mix_g;                  // [R,16] G[r][c] = sum_b d_comp[r][b] E[c][b] (field_mix_grad_kernel): d m_n = ws_n G[ray(n)]
  float* part_ms;                      // per-tile partials of ws_n m_n [(g*2 + az)*16 + c] (-> dE = (sum_n ws_n m_n)^T d_comp per ray)
  float* mws16;                        // [R,16] the same sums for rays strictly inside one tile
  const float *t0, *t1;                // [N] sample intervals (gradient scaling by distance), or null
  float* dots;                         // [N]
};

// NeRF positional encoding slots of quarter q (3 per lane) and SH slots (4 per lane)
__device__ __forceinline__ void pe_slots(float (&pe)[3], float x, float y, float z, int q) {
#ifdef UMHS_ABL_NO_TRIG
  pe[0] = x * 0.5f, pe[1] = y * 0.25f, pe[2] = z + q;
  return;
#endif
  const bool odd = q & 1;
  const float c0 = odd ? y : x, c1 = odd ? z : x, c2 = odd ? z : y;
  const float f0 = odd ? 2.0f : 1.0f, f1 = odd ? 1.0f : 2.0f, f2 = odd ? 2.0f : 1.0f;
  // sin(2 pi x f [+ pi/2]) as v_sin_f32 of the phase in revolutions (x f [+ 1/4], reduced by v_fract): ~1e-6 absolute, an
  // order of magnitude inside the parity budget; the libm sinf it replaces was ~6 % of the forward kernel (range reduction)
  float r0 = c0 * f0, r1 = c1 * f1, r2 = c2 * f2;
  if (q >= 2) r0 += 0.25f, r1 += 0.25f, r2 += 0.25f;
  pe[0] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r0));
  pe[1] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r1));
  pe[2] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r2));
}

__device__ __forceinline__ void sh_slots(float (&sh)[4], float dx, float dy, float dz, int q) {
  const float x = (dx + 1.0f) / 2.0f, y = (dy + 1.0f) / 2.0f, z = (dz + 1.0f) / 2.0f;
  const float xx = x * x, yy = y * y, zz = z * z;
  if (q == 0) {
    sh[0] = 0.28209479177387814f, sh[1] = 0.4886025119029199f * y, sh[2] = 0.4886025119029199f * z;
    sh[3] = 0.4886025119029199f * x;
  } else if (q == 1) {
    sh[0] = 1.0925484305920792f * x * y, sh[1] = 1.0925484305920792f * y * z;
    sh[2] = 0.9461746957575601f * zz - 0.31539156525251999f, sh[3] = 1.0925484305920792f * x * z;
  } else if (q == 2) {
    sh[0] = 0.5462742152960396f * (xx - yy), sh[1] = 0.5900435899266435f * y * (3.0f * xx - yy);
    sh[2] = 2.890611442640554f * x * y * z, sh[3] = 0.4570457994644658f * y * (5.0f * zz - 1.0f);
  } else {
    sh[0] = 0.3731763325901154f * z * (5.0f * zz - 3.0f), sh[1] = 0.4570457994644658f * x * (5.0f * zz - 1.0f);
    sh[2] = 1.445305721320277f * z * (xx - yy), sh[3] = 0.5900435899266435f * x * (xx - 3.0f * yy);
  }
}

// Everything the heads need, recomputed per 16-sample column tile (NT tiles per wave).
template <int NT>
struct HeadState {
  float m[NT][4];   // sigmoid(head) * softmax(feat/T)   (rows c = 4q+r, zero for c >= C)
  float sg[NT][4];  // sigmoid(head)
  float ab[NT][4];  // abundances
  float s1[NT];     // sigmoid of the extra feature logit (specular gate)
};

template <int NT, bool SPEC>
__device__ __forceinline__ void head_epilogue(HeadState<NT>& hs, const v4f (&hd4)[NT][1], const v4f (&fl4)[NT][1], int C,
                                              float temperature, int lane) {
  const int q = lane >> 4;
  const float inv_t = 1.0f / temperature;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    float z[4], zmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      z[r] = fl4[ct][0][r] * inv_t;
      if (4 * q + r < C) zmax = fmaxf(zmax, z[r]);
    }
    zmax = xq_max(zmax);
    float e[4], sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e[r] = (4 * q + r < C) ? fexp(z[r] - zmax) : 0.0f;
      sum += e[r];
    }
    sum = frcp(xq_sum(sum));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool v = 4 * q + r < C;
      hs.ab[ct][r] = e[r] * sum;
      hs.sg[ct][r] = v ? sigmoidf_(hd4[ct][0][r]) : 0.0f;
      hs.m[ct][r] = hs.sg[ct][r] * hs.ab[ct][r];
    }
    if (SPEC) {
      const float mine = sel4(fl4[ct][0], C & 3);
      hs.s1[ct] = sigmoidf_(__shfl(mine, ((C >> 2) << 4) | (lane & 15), 64));
    } else {
      hs.s1[ct] = 0.0f;
    }
  }
}

template <int NT>
__device__ __forceinline__ void store_density(const FieldIO& io, const v4f (&bo4)[NT][1], const int64_t (&nn)[NT],
                                              const bool (&ok)[NT], int q) {
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    if (ok[ct]) {
      if (q == 0) {
        const float raw = bo4[ct][0][0];
        io.sigma[nn[ct]] = expf(raw) * io.sel[nn[ct]];
        if (io.sigma_raw) io.sigma_raw[nn[ct]] = raw;
      }
      if (io.emb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 4 * q + r - 1;
          if (e >= 0) io.emb[nn[ct] * 15 + e] = bo4[ct][0][r];
        }
      }
      if (io.bo16) *reinterpret_cast<v4f*>(io.bo16 + nn[ct] * 16 + 4 * q) = bo4[ct][0];  // one 64-byte row per sample
    }
  }
}

// =============================================================================================
// Shared by the forward and the transpose-free backward: LDS image segments, the three-piece bf16 form of the fp32 GEMM chain
// =============================================================================================
struct ImgSegs {
  int n, src[6], dst[6], len[6];  // float offsets / lengths, multiples of 4
};
__device__ __forceinline__ void copy_segs(float* dst, const float* __restrict__ src, const ImgSegs& sg) {
  // Every workgroup of a launch copies the SAME image at the same moment: walking it in the same order queues all CUs of an XCD on
  // one L2 channel at a time (stamps: 21 k cycles from kernel start to the barrier behind the copy of ~100 KB, 6 % of the backward
  // kernels; 15-19 k with each workgroup starting at its own rotation of the chunk sequence; 8 unconditional loads in flight per
  // thread: 21 k again, 4-8 conditional ones 26-31 k).
  for (int k = 0; k < sg.n; ++k) {
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src + sg.src[k]);
    float4* d4 = reinterpret_cast<float4*>(dst + sg.dst[k]);
    const int n4 = sg.len[k] >> 2, bd = blockDim.x;
    const int nfull = n4 / bd;  // whole chunks of blockDim float4s: copied without a condition, 4 loads in flight, rotated start
    const int c0 = nfull ? (int)((blockIdx.x * 7u) % (unsigned)nfull) : 0;
    int c = 0;
    for (; c + 4 <= nfull; c += 4) {
      float4 v[4];
      int idx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int cc = c0 + c + u;
        cc = cc >= nfull ? cc - nfull : cc;
        idx[u] = cc * bd + (int)threadIdx.x;
        v[u] = s4[idx[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) d4[idx[u]] = v[u];
    }
    for (; c < nfull; ++c) {
      int cc = c0 + c;
      cc = cc >= nfull ? cc - nfull : cc;
      d4[cc * bd + threadIdx.x] = s4[cc * bd + threadIdx.x];
    }
    const int i = nfull * bd + threadIdx.x;  // the partial last chunk
    if (i < n4) d4[i] = s4[i];
  }
}

enum TLayerId { T_B1 = 0, T_B0, T_H2, T_H1, T_H0, T_F2, T_F1, T_F0, T_D1, T_MX, NTLAYERS };

typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
// two floats -> their bf16 roundings (nearest even) in one dword (v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ uint32_t cvt_pk_bf(float a, float b) {
  const v2f v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, v2bf));
}
// (x0, x1) -> packed bf16 pieces: h = rne(x), m = rne(x - h) [, l = rne(x - h - m)]; the subtractions are exact.  The transposes of
// the dW operands and the chain's three-piece products call this on the same registers: the compiler keeps one computation.
__device__ __forceinline__ void bf_split_pair(float x0, float x1, uint32_t& h, uint32_t& m, float& r0, float& r1) {
  h = cvt_pk_bf(x0, x1);
  r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf(r0, r1);
}
__device__ __forceinline__ void bf_split_pair3(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  float r0, r1;
  bf_split_pair(x0, x1, h, m, r0, r1);
  l = cvt_pk_bf(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}

// ---- the exact fp32 chain on the bf16 MFMA: x = hi + mid + lo (three bf16 pieces, |x - hi - mid - lo| <= 2^-25 |x|), a product
// a*b = hh + hm + mh + hl + lh + mm (the three terms left out are <= 2^-24 |ab|: fp32's own rounding), exact bf16 products,
// fp32 accumulation.  v_mfma_f32_16x16x32_bf16 issues in half the cycles of v_mfma_f32_16x16x4_f32 for 8x its K, so a 64-wide
// layer costs 12 bf16 MFMAs per output tile instead of 16 fp32 ones at half the cycles each: 0.375x the matrix time.
// Pack images: the fp32 images re-laid for K = 32 (lane (out, q) holds the weights of k-slots 8S+u, u < 8: the SAME k-slots as
// steps 8S .. 8S+7 of the fp32 form, so the activation registers are used in the order they are) and split into the three pieces:
//   wbf[off + ((((t*K8 + S)*3 + piece)*64 + lane)*4 + u/2] = pack(piece(w(t, 8S+u, lane)), piece(w(t, 8S+u+1, lane)))      (dwords)
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct BfConv {  // one converted layer: where its fp32 pack sits in its source image, where the bf16x3 pack goes
  int src_img, src_off, KS4, OT, dst_off;  // src_img 0: forward pack image, 1: transposed pack image
};
struct BfPlan {
  int n, total;  // dwords
  BfConv c[16];
};
__device__ __forceinline__ void bf_split3_scalar(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  const __bf16 bh = (__bf16)x;
  const float r1 = x - (float)bh;
  const __bf16 bm = (__bf16)r1;
  const float r2 = r1 - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = (uint16_t)__builtin_bit_cast(short, bh), m = (uint16_t)__builtin_bit_cast(short, bm), l = (uint16_t)__builtin_bit_cast(short, bl);
}
// acc[ct][t] (+)= W-pack(t, :) x b[ct][:] with the three-piece bf16 products (same INIT meaning and result tile layout as gemm_pack
// -- the C/D map of the MFMA does not depend on the input type; the NT sample tiles share every weight fragment)
#ifndef BF_PF
#define BF_PF 2
#endif
#ifndef BF_TMAJOR  // fragment order of gemm_bf: 0 = k-step major (every output tile finishes at the end), 1 = output-tile major
#define BF_TMAJOR 0
#endif
#ifndef BF_PIN  // 1: every step of gemm_bf pinned with full scheduling barriers, the first fragments requested in front of the B splits
#define BF_PIN 1
#endif
template <int OT, int KS, int NT, int INIT>
__device__ __forceinline__ void gemm_bf(v4f (&acc)[NT][OT], const float (&b)[NT][KS], const uint32_t* __restrict__ w,
                                        const float* __restrict__ bias, int lane) {
  constexpr int K8 = (KS + 7) / 8;
  constexpr int NF = OT * K8;  // fragment f = S * OT + t: three 16-byte pieces each, requested BF_PF fragments ahead
  constexpr int PF = BF_PF < NF ? BF_PF : NF;
  v4u A[PF + 1][3];
  auto load = [&](int f, int slot) __attribute__((always_inline)) {
    const int t = BF_TMAJOR ? f / K8 : f % OT, S = BF_TMAJOR ? f % K8 : f / OT;
    const uint32_t* pw = w + (((t * K8 + S) * 3) * 64 + lane) * 4;
#pragma unroll
    for (int p = 0; p < 3; ++p) A[slot][p] = *reinterpret_cast<const v4u*>(pw + p * 256);
  };
  // Stamps of the backward (tools/stamp_fbwd.py) gave 33-42 cycles per v_mfma_f32_16x16x32 in every gemm of a one-wave-per-SIMD kernel,
  // against the pipe's 16: the ISA had each fragment's ds_read_b128s right in front of its MFMAs with s_waitcnt lgkmcnt(0) in between --
  // under register pressure the scheduler sinks the reads this loop requests ahead down to their uses (the masked barrier below only
  // kept their order).  With BF_PIN every step is a scheduling region of its own, [reads of fragment f + PF] [products of fragment f],
  // and the first PF fragments (and the bias tile) are requested BEFORE the bf16 splits of the B operand, ~70 VALU instructions that
  // cover their latency: 19-20 cycles per MFMA.
  v4f bv[OT];
  if (BF_PIN) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < PF; ++f) load(f, f);
    if (INIT == 2) {
#pragma unroll
      for (int t = 0; t < OT; ++t) bv[t] = *reinterpret_cast<const v4f*>(bias + 16 * t + 4 * (lane >> 4));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  v4u B[NT][3][K8];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int S = 0; S < K8; ++S)
#pragma unroll
      for (int up = 0; up < 4; ++up) {
        const int s0 = 8 * S + 2 * up;  // KS is even or the last slot is a zero pad
        uint32_t h, m, l;
        bf_split_pair3(s0 < KS ? b[ct][s0 < KS ? s0 : 0] : 0.0f, s0 + 1 < KS ? b[ct][s0 + 1 < KS ? s0 + 1 : 0] : 0.0f, h, m, l);
        B[ct][0][S][up] = h, B[ct][1][S][up] = m, B[ct][2][S][up] = l;
      }
  if (INIT != 0) {
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      v4f bvt = {0.0f, 0.0f, 0.0f, 0.0f};
      if (INIT == 2) bvt = BF_PIN ? bv[t] : *reinterpret_cast<const v4f*>(bias + 16 * t + 4 * (lane >> 4));
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct][t] = bvt;
    }
  }
  auto mf = [](const v4u& a, const v4u& bb, const v4f& c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, bb), c, 0, 0, 0);
  };
  if (!BF_PIN) {
#pragma unroll
    for (int f = 0; f < PF; ++f) load(f, f);
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (BF_PIN) __builtin_amdgcn_sched_barrier(0);
    if (f + PF < NF) load(f + PF, (f + PF) % (PF + 1));
    if (BF_PIN)
      __builtin_amdgcn_sched_barrier(0);
    else
      __builtin_amdgcn_sched_barrier(0x7ff & ~0x180);  // LDS reads stay where they are; everything else may move
    const int t = BF_TMAJOR ? f / K8 : f % OT, S = BF_TMAJOR ? f % K8 : f / OT, k = f % (PF + 1);
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};  // hh, hm, mh, hl, lh, mm
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[ct][t] = mf(A[k][PA[p]], B[ct][PB[p]][S], acc[ct][t]);
  }
  if (BF_PIN) __builtin_amdgcn_sched_barrier(0);
}

struct BfOffs {  // LDS dword offsets (from the bf16 region) of the converted layers' bf16x3 packs, -1: layer keeps its fp32 pack
  int f[NLAYERS], t[NTLAYERS];
};


// =============================================================================================
// Forward
// =============================================================================================
// BF: the layers with >= 7 k-steps (everything but the band tiles and the directional hidden layer) run as three-piece bf16
// products (gemm_bf): 6 v_mfma_f32_16x16x32_bf16 per 8 k-slots instead of 8 v_mfma_f32_16x16x4_f32 at twice the cycles each.  Their
// bf16x3 packs (1.5x the fp32 bytes) make the image ~110 KB: one 8-wave workgroup per CU instead of two 4-wave ones.
struct FwdBfArgs {
  ImgSegs seg_f, seg_b;  // fp32 part (packs of the other layers + every bias) and bf16x3 part of the LDS image
  int bf_off;            // dword offset of the bf16x3 part in LDS
  BfOffs bo;
  const float* bf_image;
};
// HEADS: the kernel starts from the base MLP's saved outputs (emb) instead of the hash features -- the density half ran as its own
// launch, so the rendering weights of every sample are known -- and forms the per-ray band sums of its outputs itself (HeadsComp
// below): the [N,B] streams that carry no loss (spectral2, specular) are never written, the third only if the caller wants it.
//
// Per-ray sums without atomics, same bits every run: a wave's 16-sample tile g = n/16 covers at most a tail of one ray, whole rays,
// and a head of another.  It stores  A[g][stream][b] = sum over the samples of the tile's FIRST sample's ray,
// Z[g][stream][b] = the same for its LAST sample's ray (when that is another ray), and writes rays that lie strictly inside the
// tile straight to comp (no other tile contributes to them).  field_heads_finish_kernel then adds, for every other ray, the A / Z
// entries of its tiles in tile order.  part[((g*2 + az)*n_streams + stream)*16*TB + b].
template <bool SPEC, bool DENSITY_ONLY, int NT, int WAVES, bool BF = false, bool HEADS = false>
__global__ __launch_bounds__(64 * WAVES, BF ? WAVES / 4 : (2 * WAVES) / 4) void field_fwd_kernel(FieldIO io, PackDesc pd,
                                                                                        const float* __restrict__ image, FwdBfArgs fb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (BF) {
    copy_segs(lds, image, fb.seg_f);  // pd carries offsets local to this compact image
    copy_segs(lds + fb.bf_off, fb.bf_image, fb.seg_b);
  } else {
    load_fwd_image(lds, pd, image, 0);
  }
  __syncthreads();
  const uint32_t* const wbf = reinterpret_cast<const uint32_t*>(lds + fb.bf_off);
#define FWD_GEMM(OT_, KS_, ACC_, B_, LID_)                                                                     \
  do {                                                                                                          \
    if constexpr (BF)                                                                                           \
      gemm_bf<OT_, KS_, NT, 2>(ACC_, B_, wbf + fb.bo.f[LID_], lds + pd.L[LID_].off_b, lane);                    \
    else                                                                                                        \
      gemm_pack<OT_, KS_, NT, 2>(ACC_, B_, lds + pd.L[LID_].off_w, lds + pd.L[LID_].off_b, lane);               \
  } while (0)
  constexpr int TILE = 16 * NT * WAVES;  // samples per workgroup iteration
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const int64_t ntiles = (io.n + TILE - 1) / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t nn[NT];
    bool ok[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      int64_t n = tile * TILE + wave * (16 * NT) + ct * 16 + j;
      ok[ct] = n < io.n;
      nn[ct] = ok[ct] ? n : io.n - 1;
    }
    float encf[NT][8];
    if (HEADS) {
      // nothing to load here: the base MLP's outputs are read below
    } else {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int lv = 0; lv < 4; ++lv) {
          const float2 v = *reinterpret_cast<const float2*>(io.enc + nn[ct] * io.sn + (int64_t)(4 * q + lv) * io.sl);
          encf[ct][2 * lv] = v.x, encf[ct][2 * lv + 1] = v.y;
        }
    }
    // ---- encodings first (sinf's slow path and the loads branch; keep them out of the gemm chain) --------
    float pe_[NT][3], sh_[NT][4];
    if (!DENSITY_ONLY) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        pe_slots(pe_[ct], io.wpos[3 * nn[ct]], io.wpos[3 * nn[ct] + 1], io.wpos[3 * nn[ct] + 2], q);
        if (SPEC) sh_slots(sh_[ct], io.dirs[3 * nn[ct]], io.dirs[3 * nn[ct] + 1], io.dirs[3 * nn[ct] + 2], q);
      }
    }
    // ---- mlp_base: 32 -> 64 -> 16 -------------------------------------------------------------
    v4f bo4[NT][1];
    if constexpr (HEADS) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 4 * q + r - 1;
          bo4[ct][0][r] = (e >= 0 && !io.bo16_in) ? io.emb_in[nn[ct] * 15 + e] : 0.0f;  // slot 0 (sigma_raw) meets a zero weight column
        }
      if (io.bo16_in) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          bo4[ct][0] = *reinterpret_cast<const v4f*>(io.bo16_in + nn[ct] * 16 + 4 * q);
          if (q == 0) bo4[ct][0][0] = 0.0f;
        }
      }
    } else {
      v4f h4[NT][4];
      FWD_GEMM(4, 8, h4, encf, L_B0);
      float h[NT][16];
      relu_to<4, NT>(h, h4);
      FWD_GEMM(1, 16, bo4, h, L_B1);
    }
    if (DENSITY_ONLY) {
      store_density<NT>(io, bo4, nn, ok, q);
      continue;
    }
    float in27[NT][7], dir28[NT][7];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
      for (int s = 0; s < 3; ++s) in27[ct][s] = pe_[ct][s];
#pragma unroll
      for (int r = 0; r < 4; ++r) in27[ct][3 + r] = bo4[ct][0][r];
      if (SPEC) {
#pragma unroll
        for (int s = 0; s < 4; ++s) dir28[ct][s] = sh_[ct][s];
#pragma unroll
        for (int s = 0; s < 3; ++s) dir28[ct][4 + s] = pe_[ct][s];
      }
    }
    // ---- mlp_head / feature_mlp: 27 -> 64 -> 64 -> C(+1) ----------------------------------------
    v4f t4[NT][4], hd4[NT][1], fl4[NT][1];
    float a1[NT][16], a2[NT][16];
    FWD_GEMM(4, 7, t4, in27, L_H0);
    relu_to<4, NT>(a1, t4);
    FWD_GEMM(4, 16, t4, a1, L_H1);
    relu_to<4, NT>(a2, t4);
    FWD_GEMM(1, 16, hd4, a2, L_H2);
    FWD_GEMM(4, 7, t4, in27, L_F0);
    relu_to<4, NT>(a1, t4);
    FWD_GEMM(4, 16, t4, a1, L_F1);
    relu_to<4, NT>(a2, t4);
    FWD_GEMM(1, 16, fl4, a2, L_F2);
    if (io.feat_logits) {  // saved for the split backward: [N,16] rows 4q..4q+3 of the logit tile, 64 B per sample
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
        if (ok[ct]) *reinterpret_cast<v4f*>(io.feat_logits + nn[ct] * 16 + 4 * q) = fl4[ct][0];
    }
    HeadState<NT> hs;
    head_epilogue<NT, SPEC>(hs, hd4, fl4, io.C, io.temperature, lane);
    // ---- mlp_directional hidden: 28 -> 16 ---------------------------------------------------------
    float hdir[NT][4];
    if (SPEC) {
      v4f d4[NT][1];
      gemm_pack<1, 7, NT, 2>(d4, dir28, lds + pd.L[L_D0].off_w, lds + pd.L[L_D0].off_b, lane);
      relu_to<1, NT>(hdir, d4);
    }
    // ---- per 16-band tile: mixing (K = classes) and specular (K = 16 hidden) -------------------------
    if constexpr (!HEADS) {
#pragma unroll 1  // a runtime loop: left alone hipcc unrolls the (small) no-specular body 8x and spills 200+ registers
      for (int t = 0; t < io.TB; ++t) {
        // transposed tiles: rows = samples 4q+r of the column tile, lanes&15 = bands 16t..16t+15 -> 64-byte row segments
        v4f sp[NT][1], sc[NT][1];
        gemm_pack<1, 4, NT, 1, true>(sp, hs.m, lds + pd.L[L_MX].off_w + t * 256, nullptr, lane);
        if (SPEC) gemm_pack<1, 4, NT, 2, true>(sc, hdir, lds + pd.L[L_D1].off_w + t * 256, lds + pd.L[L_D1].off_b + 16 * t, lane);
        const int b = 16 * t + j;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const int64_t nb = tile * TILE + wave * (16 * NT) + ct * 16 + 4 * q;  // first sample of this lane's 4 rows
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s1r = SPEC ? __shfl(hs.s1[ct], 4 * q + r, 64) : 0.0f;  // s1 lives on lane&15 = sample
#ifdef UMHS_ABL_NO_STORE
            if (nb + r < io.n && b < io.B && sp[ct][0][r] == 1.2345e30f) {
#else
            if (nb + r < io.n && b < io.B) {
#endif
              const float spec = sp[ct][0][r];
              const float spl = SPEC ? s1r * sigmoidf_(sc[ct][0][r]) : 0.0f;
              const int64_t o = (nb + r) * io.B + b;
              io.spectral[o] = SPEC ? spec + spl : spec;
              if (SPEC && io.spectral2) io.spectral2[o] = spec;
              if (SPEC && io.specular) io.specular[o] = spl;
            }
          }
        }
      }
      // ---- per-sample scalars last (conditional stores = branches) -----------------------------------
      store_density<NT>(io, bo4, nn, ok, q);
    }
    if (io.abund) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ok[ct] && 4 * q + r < io.C) io.abund[nn[ct] * io.C + 4 * q + r] = hs.ab[ct][r];
    }
    if constexpr (HEADS) {
      // Per-ray sums inside the kernel.  The mixing term is linear in m: sum_n w_n (m_n E) = (sum_n w_n m_n) E, so the kernel sums
      // w_n m_n (16 classes) per ray and the finish pass multiplies by E once per RAY -- no mixing product per sample and band tile at
      // all; only the specular term (a sigmoid per sample and band) is formed per band tile.  Abundances are a third 16-wide stream.
      float w4[NT][4], wj[NT];
      int r4[NT][4], rj[NT], rfirst[NT], rlast[NT];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const int64_t nb0 = tile * TILE + wave * (16 * NT) + ct * 16;  // first sample of the tile
        const int64_t last = nb0 + 15 < io.n ? nb0 + 15 : io.n - 1;
        rfirst[ct] = nb0 < io.n ? __builtin_amdgcn_readfirstlane((int)io.ray_of[nb0]) : -1;
        rlast[ct] = nb0 < io.n ? __builtin_amdgcn_readfirstlane((int)io.ray_of[last]) : -1;
        const bool vj = nb0 + j < io.n;
        wj[ct] = vj ? io.weights[nb0 + j] : 0.0f;  // this lane's sample (samples-on-lanes tiles) ...
        rj[ct] = vj ? (int)io.ray_of[nb0 + j] : -2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // ... and the four rows 4q..4q+3 it holds of a transposed band tile
          w4[ct][r] = __shfl(wj[ct], 4 * q + r, 64);
          r4[ct][r] = __shfl(rj[ct], 4 * q + r, 64);
        }
      }
      if (SPEC) {
#pragma unroll 1
        for (int t = 0; t < io.TB; ++t) {
          v4f sc[NT][1];
          gemm_pack<1, 4, NT, 2, true>(sc, hdir, lds + pd.L[L_D1].off_w + t * 256, lds + pd.L[L_D1].off_b + 16 * t, lane);
          const int b = 16 * t + j, BP = 16 * io.TB;
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            if (rfirst[ct] < 0) continue;
            const int64_t g = (tile * TILE + wave * (16 * NT) + ct * 16) >> 4;
            float val[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = w4[ct][r] * (__shfl(hs.s1[ct], 4 * q + r, 64) * sigmoidf_(sc[ct][0][r]));
            auto col_sum = [&](int ray) __attribute__((always_inline)) {
              float a = 0.0f;
#pragma unroll
              for (int r = 0; r < 4; ++r) a += (r4[ct][r] == ray) ? val[r] : 0.0f;
              return xq_sum(a);
            };
            const float a = col_sum(rfirst[ct]);
            if (q == 0) io.part[(g * 2 + 0) * BP + b] = a;
            if (rlast[ct] != rfirst[ct]) {  // wave-uniform
              const float z = col_sum(rlast[ct]);
              if (q == 0) io.part[(g * 2 + 1) * BP + b] = z;
              for (int m = rfirst[ct] + 1; m < rlast[ct]; ++m) {  // rays strictly inside this tile (short rays; rare)
                const float v = col_sum(m);
                if (q == 0 && b < io.B) io.comp[2][(int64_t)m * io.B + b] = v;
              }
            }
          }
        }
      }
      // the two 16-wide streams (w m and w abundances): samples on lanes (lane = (sample j, q), reg r <-> class 4q+r) -- the sum
      // over a tile's samples is a reduction over the 16 lanes of a row, masked by the sample's ray
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        if (rfirst[ct] < 0) continue;
        const int64_t g = (tile * TILE + wave * (16 * NT) + ct * 16) >> 4;
        auto row_sum = [&](const float(&x)[4], int ray, float(&out)[4]) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            out[r] = row_sum16((rj[ct] == ray) ? wj[ct] * x[r] : 0.0f);
          }
        };
        auto stream16 = [&](const float(&x)[4], float* __restrict__ part16, float* __restrict__ direct, int width) __attribute__((always_inline)) {
          float a[4];
          row_sum(x, rfirst[ct], a);
          if (j == 0) *reinterpret_cast<v4f*>(part16 + (g * 2 + 0) * 16 + 4 * q) = v4f{a[0], a[1], a[2], a[3]};
          if (rlast[ct] != rfirst[ct]) {
            row_sum(x, rlast[ct], a);
            if (j == 0) *reinterpret_cast<v4f*>(part16 + (g * 2 + 1) * 16 + 4 * q) = v4f{a[0], a[1], a[2], a[3]};
            for (int m = rfirst[ct] + 1; m < rlast[ct]; ++m) {
              row_sum(x, m, a);
              if (j == 0 && direct) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (4 * q + r < width) direct[(int64_t)m * width + 4 * q + r] = a[r];
              }
            }
          }
        };
        stream16(hs.m[ct], io.part_m, io.mix16, 16);
        if (io.part_ab) stream16(hs.ab[ct], io.part_ab, io.comp_ab, io.C);
      }
    }
  }
}
#undef FWD_GEMM

// =============================================================================================
// Backward.  Two kernels so that each fits 2 waves per SIMD (8-wave workgroups, <= 256 VGPRs):
//   heads: feature_mlp / mlp_head / mlp_directional / mixing.  Reads the forward's saved emb, recomputes the head
//          activations per 16-sample column tile, runs the dX chain and the dW products, emits d_bo [N,16]
//          (gradient w.r.t. the base MLP's 16 outputs).
//   base : mlp_base.  Recomputes its hidden layer from the hash features, consumes d_bo + d_sigma, emits d_enc.
// Transposed pack images (A operands of the dX chain) are built once per call in global memory (L2-resident).
// =============================================================================================
struct TDesc {
  const float* W;
  int OUT, IN, KS, OT, rowmap, off;  // rowmap 0: in = rho, 1: emb slots of the 27-d input, 2: L_MX (E[c=rho][b=k])
};
struct TPackDesc {
  TDesc L[NTLAYERS];
  int total;
};

//   wT[off + ((t*KS4 + s4)*64 + lane)*4 + ss] = W[k(4*s4+ss, lane>>4)][rowmap(16t + (lane&15))]
__device__ __forceinline__ float t_pack_value(const TPackDesc& td, int idx) {
  int li = 0;
  while (li + 1 < NTLAYERS && idx >= td.L[li + 1].off) ++li;
  const TDesc& L = td.L[li];
  const int rel = idx - L.off;
  const int ss = rel & 3, ln = (rel >> 2) & 63, blk = rel >> 8;
  const int KS4 = (L.KS + 3) >> 2;
  const int t = blk / KS4, s = (blk % KS4) * 4 + ss;
  const int rho = 16 * t + (ln & 15), q = ln >> 4;
  const int k = 16 * (s >> 2) + 4 * q + (s & 3);  // dZ row held by (step s, quarter q)
  float v = 0.0f;
  if (L.rowmap == 2) {
    if (rho < L.OUT && k < L.IN) v = L.W[(size_t)rho * L.IN + k];  // E[c][b], OUT=C, IN=B
  } else if (s < L.KS && k < L.OUT) {
    int in = rho;
    if (L.rowmap == 1) in = (rho >= 1 && rho <= 15) ? 12 + rho - 1 : -1;
    if (in >= 0 && in < L.IN) v = L.W[(size_t)k * L.IN + in];
  }
  return v;
}

// Every weight image of one direction in ONE launch, each dword computed straight from the parameters: the fp32 forward image
// (weights + biases), the transposed image, the bf16x3 image of the converted layers.  (As three dependent launches -- fp32 images,
// then the bf16x3 image read back from them -- the packs of a step cost 26 us of small kernels; 2 x 6 us like this.)
struct PackJob {
  PackDesc pd;
  TPackDesc td;
  BfPlan bp;
  float* img;    // [pd.total] or null
  float* wT;     // [td.total] or null
  uint32_t* bf;  // [bp.total] or null
};
#if UMHS_TU_MAIN
__global__ __launch_bounds__(256) void field_pack_all_kernel(PackJob jb) {
  const int n_img = jb.img ? jb.pd.total : 0, n_wT = jb.wT ? jb.td.total : 0, n_bf = jb.bf ? jb.bp.total : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_img + n_wT + n_bf; i += gridDim.x * 256) {
    if (i < n_img) {
      float v = 0.0f;
      if (i < jb.pd.total_w) {
        v = fwd_pack_value(jb.pd, i);
      } else {
        int li = 0;
        while (li + 1 < NLAYERS && i >= jb.pd.L[li + 1].off_b) ++li;  // off_b is non-decreasing (L_MX has no bias tile)
        const LayerDesc& L = jb.pd.L[li];
        const int o = i - L.off_b;
        v = (li != L_MX && L.b && o < L.OUT) ? L.b[o] : 0.0f;
      }
      jb.img[i] = v;
    } else if (i < n_img + n_wT) {
      jb.wT[i - n_img] = t_pack_value(jb.td, i - n_img);
    } else {
      const int idx = i - n_img - n_wT;
      int ci = 0;
      while (ci + 1 < jb.bp.n && idx >= jb.bp.c[ci + 1].dst_off) ++ci;
      const BfConv& c = jb.bp.c[ci];
      const int rel = idx - c.dst_off;
      const int up = rel & 3, lane = (rel >> 2) & 63, blk = rel >> 8;  // blk = (t*K8 + S)*3 + piece
      const int piece = blk % 3, ts = blk / 3;
      const int K8 = (c.KS4 + 1) >> 1, t = ts / K8, S = ts % K8;
      uint32_t out = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int u = 2 * up + e, s4 = 2 * S + (u >> 2);
        float w = 0.0f;
        if (s4 < c.KS4) {
          const int src = c.src_off + ((t * c.KS4 + s4) * 64 + lane) * 4 + (u & 3);
          w = c.src_img ? t_pack_value(jb.td, src) : fwd_pack_value(jb.pd, src);
        }
        uint32_t h, m, l;
        bf_split3_scalar(w, h, m, l);
        out |= (piece == 0 ? h : (piece == 1 ? m : l)) << (16 * e);
      }
      jb.bf[idx] = out;
    }
  }
}
#endif

// =============================================================================================
// Transpose-free backward (the default): two kernels, no LDS staging and no barrier inside their loops.
//   PART 0: mlp_head + mlp_directional + mixing.  Reads the forward's emb and feature logits, emits d_fl [N,16] (gradient of
//           the feature logits) and d_bo [N,16] (its share of the gradient of the base MLP's outputs).
//   PART 1: feature_mlp + mlp_base.  Recomputes the base MLP from the hash features (so it needs no saved emb / sigma_raw),
//           consumes d_fl, d_bo, d_sigma, writes d_enc.
//
// The dX chain and the forward recompute are the exact fp32 MFMA chain of the kernels above.  What changed is dW = dZ^T X, the
// contraction over samples (= over lanes in the "samples on lanes" layout).  The kernels above stage both operands through LDS
// to transpose them ([sample][feature] rows written with ds_write_b128, read back column-wise) -- by the PMC and the
// section ablations ~100 us of staging and ~20 us of barriers for 58 us of MFMA work at C2.  Here a tile is transposed ON THE
// MATRIX PIPE: an fp32 value is split into two bf16 pieces (x = hi + lo + O(2^-17 x)), and one v_mfma_f32_16x16x16_bf16 of a
// piece against an identity B operand (every lane builds its fragment from its own id) yields the "swapped" tile -- lane =
// (feature l&15, quarter q), register r <-> sample 4q+r -- exactly, because the products are x * 1.  Two swapped tiles ARE the
// A and B operands of dW[out][in] += sum_s dZ[s][out] X[s][in] on the bf16 MFMA (k-slot <-> sample), evaluated as
// hi*hi + hi*lo + lo*hi with fp32 accumulation: 2^-16 relative per product, unbiased (round-to-nearest pieces), summed over
// 262 k samples -- well inside the 5e-5 gradient budget (tests/test_hip_parity.py::test_field_bwd, test_hip_trajectory.py).
// The bf16 MFMA issues in half the cycles of the fp32 one for 4x its K, so transposes + dW cost ~1/4 of the fp32 dW they replace.
// Consequence: every wave owns ALL dW tiles of its part for its own samples (148 / 174 accumulator registers at C2, 238 in
// part 0 at 192 bands) -- one wave per SIMD with the accumulators in AGPRs, four independent waves per workgroup, every weight
// pack (forward and transposed) LDS-resident for any band count the forward supports.
// =============================================================================================
typedef short v4s __attribute__((ext_vector_type(4)));
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

struct STile {  // swapped 16-feature x 16-sample tile in bf16 pieces: lane (feature c = l&15, q = l>>4), element r <-> sample 4q+r
  v4s hi, lo;
};

__device__ __forceinline__ v4s ident_frag(int lane) {  // B operand of the transposing MFMA: I[k = 4q+u][col c] = (k == c)
  const int c = lane & 15, q = lane >> 4;
  v4s f;
#pragma unroll
  for (int u = 0; u < 4; ++u) f[u] = ((c >> 2) == q && (c & 3) == u) ? (short)0x3F80 : (short)0;
  return f;
}
__device__ __forceinline__ v4s pack_hi16(const v4f& v) {  // the four values ARE bf16 numbers: keep their upper halves
  const uint32_t a = __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
  const uint32_t b = __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07060302u);
  return __builtin_bit_cast(v4s, make_uint2(a, b));
}

// "samples on lanes" tiles (4 registers each: features 4q+r of a 16-feature tile, this lane's sample) -> swapped bf16 tiles.
// NTILE tiles at once, in three phases -- split every value into its bf16 pieces (VALU), all 2*NTILE transposing MFMAs back to
// back, then pack the results: with one wave per SIMD nothing else hides an MFMA's latency, so a tile-by-tile split -> MFMA ->
// pack chain would stall on every tile (PMC of the first version: SQ_WAIT_INST_ANY 36-44 % of the wave cycles).
// COLSUM: also adds each lane's share of the column sums (sum over its 4 samples; the 4 lane quarters are added by the reduce).
template <int NTILE, bool COLSUM>
__device__ __forceinline__ void to_swapped_n(STile* __restrict__ out, const float* __restrict__ x, const v4s& ident,
                                             float* __restrict__ colsum = nullptr) {
  v4s hi[NTILE], lo[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    uint32_t h[2], m[2];
    float r0, r1;
#pragma unroll
    for (int e = 0; e < 2; ++e) bf_split_pair(x[4 * t + 2 * e], x[4 * t + 2 * e + 1], h[e], m[e], r0, r1);
    hi[t] = __builtin_bit_cast(v4s, make_uint2(h[0], h[1])), lo[t] = __builtin_bit_cast(v4s, make_uint2(m[0], m[1]));
  }
  const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
  v4f dh[NTILE], dl[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) dh[t] = MFMA_BF(hi[t], ident, z);
#pragma unroll
  for (int t = 0; t < NTILE; ++t) dl[t] = MFMA_BF(lo[t], ident, z);
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    if (COLSUM) colsum[t] += ((dh[t][0] + dh[t][1]) + (dh[t][2] + dh[t][3])) + ((dl[t][0] + dl[t][1]) + (dl[t][2] + dl[t][3]));
    out[t].hi = pack_hi16(dh[t]), out[t].lo = pack_hi16(dl[t]);
  }
}
template <bool COLSUM>
__device__ __forceinline__ STile to_swapped(const float* __restrict__ x4, const v4s& ident, float* colsum = nullptr) {
  STile s;
  to_swapped_n<1, COLSUM>(&s, x4, ident, colsum);
  return s;
}

// acc[to * TI + ti] += Z[to]^T X[ti]: three bf16 products per tile pair (hi*hi, hi*lo, lo*hi), the passes run over the ti's of a row so
// that no MFMA waits for the one before it on the same accumulator.
// These MFMAs are written as inline asm with the accumulators constrained to AGPRs ("+a"), and this file is compiled with
// -amdgpu-mfma-vgpr-form: a kernel whose register budget exceeds 256 otherwise gets the AGPR form of EVERY MFMA, and each result the
// VALU touches (every ReLU input, every transposed tile: 516 of the 2,066 instructions of part 1's loop) is first copied out of the
// accumulator file with v_accvgpr_read.  With the flag the builtin MFMAs (fp32 chain, transposes) write VGPRs; only the dW
// accumulators, which nothing but these MFMAs touches until the end of the launch, live in AGPRs.
// Hazards the compiler cannot see inside the string: the A / B operands come out of v_perm_b32 (VALU write -> MFMA read: 2 wait
// states = the leading s_nop 1); an MFMA accumulating onto the previous one's D needs none; D is next read by v_accvgpr_read after
// the loop.
template <int TI>
__device__ __forceinline__ void dw_row(v4f* __restrict__ acc, const STile& z, const STile* __restrict__ X);
template <>
__device__ __forceinline__ void dw_row<1>(v4f* __restrict__ acc, const STile& z, const STile* __restrict__ X) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %1, %3, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %1, %4, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %2, %3, %0"
      : "+a"(acc[0])
      : "v"(z.hi), "v"(z.lo), "v"(X[0].hi), "v"(X[0].lo));
}
template <>
__device__ __forceinline__ void dw_row<2>(v4f* __restrict__ acc, const STile& z, const STile* __restrict__ X) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %2, %4, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %1, %2, %6, %1\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %2, %5, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %1, %2, %7, %1\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %1, %3, %6, %1"
      : "+a"(acc[0]), "+a"(acc[1])
      : "v"(z.hi), "v"(z.lo), "v"(X[0].hi), "v"(X[0].lo), "v"(X[1].hi), "v"(X[1].lo));
}
template <>
__device__ __forceinline__ void dw_row<4>(v4f* __restrict__ acc, const STile& z, const STile* __restrict__ X) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %4, %6, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %1, %4, %8, %1\n\t"
      "v_mfma_f32_16x16x16_bf16 %2, %4, %10, %2\n\t"
      "v_mfma_f32_16x16x16_bf16 %3, %4, %12, %3\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %4, %7, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %1, %4, %9, %1\n\t"
      "v_mfma_f32_16x16x16_bf16 %2, %4, %11, %2\n\t"
      "v_mfma_f32_16x16x16_bf16 %3, %4, %13, %3\n\t"
      "v_mfma_f32_16x16x16_bf16 %0, %5, %6, %0\n\t"
      "v_mfma_f32_16x16x16_bf16 %1, %5, %8, %1\n\t"
      "v_mfma_f32_16x16x16_bf16 %2, %5, %10, %2\n\t"
      "v_mfma_f32_16x16x16_bf16 %3, %5, %12, %3"
      : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3])
      : "v"(z.hi), "v"(z.lo), "v"(X[0].hi), "v"(X[0].lo), "v"(X[1].hi), "v"(X[1].lo), "v"(X[2].hi), "v"(X[2].lo), "v"(X[3].hi),
        "v"(X[3].lo));
}
template <int TO, int TI>
__device__ __forceinline__ void dw_pairs(v4f* __restrict__ acc, const STile (&Z)[TO], const STile (&X)[TI]) {
#pragma unroll
  for (int to = 0; to < TO; ++to) dw_row<TI>(acc + to * TI, Z[to], X);
}

// accumulator / bias-sum slots of a wave (items of 64 lanes x 4 floats; the slab keeps this order).  Part 0 owns the accumulator
// items [0, A1) and the bias tiles [0, D1S); part 1 the rest.
template <int TBMAX>
struct TfSlots {
  static constexpr int A_H0 = 0, A_H1 = 8, A_H2 = 24, A_D0 = 28, A_D1 = 30, A_MX = 30 + TBMAX, A1 = 30 + 2 * TBMAX;
  static constexpr int A_F0 = A1, A_F1 = A1 + 8, A_F2 = A1 + 24, A_B0 = A1 + 28, A_B1 = A1 + 36, NACC = A1 + 40;
  static constexpr int D_H0 = 0, D_H1 = 4, D_H2 = 8, D_D0 = 9, D_D1 = 10, D1S = 4 * ((10 + TBMAX + 3) / 4);
  static constexpr int D_F0 = D1S, D_F1 = D1S + 4, D_F2 = D1S + 8, D_B0 = D1S + 9, D_B1 = D1S + 13, NDB = 4 * ((D1S + 14 + 3) / 4);
  static constexpr int NITEMS = NACC + NDB / 4;
  // part p's accumulator items [acc0(p), acc1(p)) and bias v4f items [dbv0(p), dbv1(p)) (absolute item = NACC + dbv)
  static constexpr int acc0(int p) { return p == 0 ? 0 : A1; }
  static constexpr int acc1(int p) { return p == 0 ? A1 : NACC; }
  static constexpr int dbv0(int p) { return p == 0 ? 0 : D1S / 4; }
  static constexpr int dbv1(int p) { return p == 0 ? D1S / 4 : NDB / 4; }
};
constexpr int TF_CHUNK = 32;  // items per round of the end-of-launch reduction over the 4 waves (4 x 32 x 1 KiB = 128 KiB of LDS)

// A lane's four bands of every band tile out of one row of d_comp / d_spectral: ``rowp`` = the row + 4q.  One 16-byte request per band
// tile with an immediate offset (rows are only 4-byte aligned when B is not a multiple of 4: dword-aligned dwordx4 is a legal global
// access) instead of four 4-byte loads with a 64-bit address each: at 128 bands the 32 scalar loads and their ~300 address
// instructions were 3.1 k cycles at the top of every 20 k-cycle tile (stamps, round 3).  Quads that straddle the end of the row (the
// last band tile when B % 4 != 0) and lanes without a sample keep the element-wise, predicated form.
struct __attribute__((packed, aligned(4))) F4u {
  float v[4];
};
template <int TBMAX>
__device__ __forceinline__ void band_row_load(float (&dall)[TBMAX][4], const float* __restrict__ rowp, bool live, int q, int TB, int B) {
#pragma unroll
  for (int t = 0; t < TBMAX; ++t) {
    const int b0 = 16 * t + 4 * q;
    if (live && t < TB && b0 + 3 < B) {
      const F4u v = *reinterpret_cast<const F4u*>(rowp + 16 * t);
#pragma unroll
      for (int r = 0; r < 4; ++r) dall[t][r] = v.v[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) dall[t][r] = (live && t < TB && b0 + r < B) ? rowp[16 * t + r] : 0.0f;
    }
  }
}

// In-kernel phase stamps of the transpose-free backward (tools/stamp_fbwd.py builds this file with -DUMHS_TF_STAMP into its own
// library): s_memtime at the phase boundaries of every tile, pinned by scheduling barriers, summed per phase by wave 0 of workgroup 0.
#ifdef UMHS_TF_STAMP
__device__ unsigned long long g_tf_stamp[2][24];
#define TF_STAMP(k_)                                  \
  do {                                                \
    __builtin_amdgcn_sched_barrier(0);                \
    stamp_[k_] = __builtin_readcyclecounter();        \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)
extern "C" int umhs_debug_tf_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tf_stamp), sizeof(unsigned long long) * 48);
}
extern "C" int umhs_debug_tf_stamps_clear() {
  unsigned long long z[48] = {};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tf_stamp), z, sizeof(z));
}
#else
#define TF_STAMP(k_) \
  do {               \
  } while (0)
#endif

// The chain (forward recompute + dX) runs on the fp32 MFMA here (v_mfma_f32_16x16x4_f32: UMHS_BWD_TF=1, and the shapes whose bf16x3
// kernels do not hold their registers); the three-piece bf16 form of the chain lives in umhs_field_zip.h.
// (Two waves per SIMD for the part-0 kernel without specular head and with the per-ray mixing -- its accumulators alone would fit --
// was tried: 128 + 128 registers, 103 spilled, 654 vs 485 us at 141 bands.)
template <int PART, bool SPEC, int TBMAX, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void field_bwd_tf_kernel(FieldIO io, PackDesc pd, TPackDesc td, const float* __restrict__ image,
                                                              const float* __restrict__ wT_image, ImgSegs seg_f, ImgSegs seg_t,
                                                              int wt_off, const float* __restrict__ bf_image, ImgSegs seg_b, int bf_off,
                                                              BfOffs bo, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef TfSlots<TBMAX> SL;
#ifdef UMHS_TF_STAMP
  const unsigned long long k_t0 = __builtin_readcyclecounter();
#endif
  copy_segs(lds, image, seg_f);  // pd / td carry offsets local to this part's LDS image
#ifdef UMHS_TF_STAMP
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long k_ta = __builtin_readcyclecounter();
#endif
  copy_segs(lds + wt_off, wT_image, seg_t);
#ifdef UMHS_TF_STAMP
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long k_tb = __builtin_readcyclecounter();
#endif
#ifdef UMHS_TF_STAMP
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long k_tc = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef UMHS_TF_STAMP
  const unsigned long long k_td = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0)
    g_tf_stamp[PART][16] = k_ta - k_t0, g_tf_stamp[PART][17] = k_tb - k_ta, g_tf_stamp[PART][19] = k_tc - k_tb, g_tf_stamp[PART][23] = k_td - k_tc;
#endif
  const float* const wT = lds + wt_off;
#define TF_GEMM_F(OT_, KS_, INIT_, ACC_, B_, LID_) gemm_pack<OT_, KS_, NT, INIT_>(ACC_, B_, lds + pd.L[LID_].off_w, lds + pd.L[LID_].off_b, lane)
#define TF_GEMM_T(OT_, KS_, INIT_, ACC_, B_, TID_) gemm_pack<OT_, KS_, NT, INIT_>(ACC_, B_, wT + td.L[TID_].off, nullptr, lane)
  constexpr int NT = 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const v4s ident = ident_frag(lane);
  constexpr int A0 = SL::acc0(PART), NA = SL::acc1(PART) - A0, DB0 = 4 * SL::dbv0(PART), NDBP = 4 * (SL::dbv1(PART) - SL::dbv0(PART));
  v4f acc_[NA];
  float db_[NDBP];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc_[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NDBP; ++i) db_[i] = 0.0f;
  // acc_ / db_ are indexed with the absolute slots of TfSlots minus this part's first slot (compile-time indices only)
  const int C = io.C, B = io.B, TB = io.TB;
  const int64_t ntiles = (io.n + 63) / 64;
  // One wave per SIMD: nothing else hides a global load, so every per-sample input of a tile is requested one tile ahead.
  struct TileIn {
    float w[3], d[3];
    float2 e[PART == 1 ? 4 : 1];
    v4f x0, x1;  // part 0: saved feature logits, -;  part 1: d_fl, d_bo (from part 0)
    float emb[4], dsig, sel, demb[4];
    float ws, tm0, tm1;  // FUSED: weights[n] (scaled by scale_n once the tile is current), the sample's interval
    int64_t ray;         // FUSED: the sample's ray
  };
  auto fetch = [&](int64_t tile, TileIn& in) {
    int64_t n = tile * 64 + wave * 16 + j;
    const bool ok = n < io.n;
    if (!ok) n = io.n - 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) in.w[s] = io.wpos[3 * n + s];
    if (PART == 0) {
      if (SPEC) {
#pragma unroll
        for (int s = 0; s < 3; ++s) in.d[s] = io.dirs[3 * n + s];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r - 1;
        in.emb[r] = (e >= 0 && !io.bo16_in) ? io.emb_in[n * 15 + e] : 0.0f;  // slot 0 (sigma_raw) meets a zero weight column
      }
      if (io.bo16_in) {  // the aligned-row form of the saved base outputs (one 16-byte load)
        const v4f b4 = *reinterpret_cast<const v4f*>(io.bo16_in + n * 16 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) in.emb[r] = (q == 0 && r == 0) ? 0.0f : b4[r];
      }
      in.x0 = *reinterpret_cast<const v4f*>(io.feat_logits_in + n * 16 + 4 * q);
      if (FUSED) {  // raw loads only: arithmetic on a prefetched value would make the wave wait for it here, a tile too early
        in.ws = io.weights[n];
        in.tm0 = io.t0 ? io.t0[n] : 1.0f, in.tm1 = io.t0 ? io.t1[n] : 1.0f;  // (t_mid = 1: scale 1)
        in.ray = io.ray_of[n];
      }
    } else {
#pragma unroll
      for (int lv = 0; lv < 4; ++lv) in.e[lv] = *reinterpret_cast<const float2*>(io.enc + n * io.sn + (int64_t)(4 * q + lv) * io.sl);
      const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
      // rows past the end carry zero upstream gradients: every dZ of theirs is then zero
      in.x0 = ok ? *reinterpret_cast<const v4f*>(io.d_fl + n * 16 + 4 * q) : z;
      in.x1 = ok ? *reinterpret_cast<const v4f*>(io.d_bo + n * 16 + 4 * q) : z;
      in.sel = io.sel[n];
      in.dsig = ok ? io.d_sigma[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r - 1;
        in.demb[r] = (io.d_emb && ok && e >= 0) ? io.d_emb[n * 15 + e] : 0.0f;
      }
    }
  };
  TileIn cur, nxt;
#ifdef UMHS_TF_STAMP
  const unsigned long long k_t1 = __builtin_readcyclecounter();
#endif
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x, cur);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t n = tile * 64 + wave * 16 + j;
    const bool ok = n < io.n;
    if (!ok) n = io.n - 1;
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, nxt);
#ifdef UMHS_TF_STAMP
    unsigned long long stamp_[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) stamp_[k] = 0;
#endif
    TF_STAMP(0);
    v4f t4[NT][4];
    float in27[NT][7];
    float pe[3];
    pe_slots(pe, cur.w[0], cur.w[1], cur.w[2], q);
#pragma unroll
    for (int s = 0; s < 3; ++s) in27[0][s] = pe[s];
    v4f dbo4[NT][1];
    dbo4[0][0] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    // one 27->64->64->out MLP (head or feature): dW of its three layers, dX down to the base-MLP slots.  x27S: swapped
    // forms of the MLP's input; input layers keep their operand order, so a swapped tile's column c = 4q'+u is whatever lane
    // quarter q' holds in slot u (tf_col() maps it back in the slab reduce): [0] positional encoding 3q'+u (u < 3), [1] base-MLP
    // output slot c.
    auto mlp3_bwd = [&](const float(&dzo)[NT][4], const float(&a2)[NT][16], const float(&a1)[NT][16], const STile(&x27S)[2],
                        v4f* __restrict__ acc2, v4f* __restrict__ acc1, v4f* __restrict__ acc0, float* __restrict__ db2,
                        float* __restrict__ db1, float* __restrict__ db0, int t2, int t1, int t0) __attribute__((always_inline)) {
      STile zS[4], xS[4];
      STile z1[1];
      z1[0] = to_swapped<true>(dzo[0], ident, db2);
      to_swapped_n<4, false>(xS, a2[0], ident);
      dw_pairs<1, 4>(acc2, z1, xS);
      TF_STAMP(8);
      v4f g4[NT][4];
      gemm_pack<4, 4, NT, 1>(g4, dzo, wT + td.L[t2].off, nullptr, lane);
      float dz1[NT][16];
#pragma unroll
      for (int i = 0; i < 16; ++i) dz1[0][i] = a2[0][i] > 0.0f ? g4[0][i >> 2][i & 3] : 0.0f;
      TF_STAMP(9);
      to_swapped_n<4, true>(zS, dz1[0], ident, db1);
      to_swapped_n<4, false>(xS, a1[0], ident);
      TF_STAMP(10);
      dw_pairs<4, 4>(acc1, zS, xS);
      TF_STAMP(11);
      TF_GEMM_T(4, 16, 1, g4, dz1, t1);
      float dz0[NT][16];
#pragma unroll
      for (int i = 0; i < 16; ++i) dz0[0][i] = a1[0][i] > 0.0f ? g4[0][i >> 2][i & 3] : 0.0f;
      TF_STAMP(12);
      to_swapped_n<4, true>(zS, dz0[0], ident, db0);
      dw_pairs<4, 2>(acc0, zS, x27S);
      TF_STAMP(13);
      TF_GEMM_T(1, 16, 0, dbo4, dz0, t0);
      TF_STAMP(14);
    };
    if constexpr (PART == 0) {
      // This tile's upstream gradients, all band tiles: requested here, consumed after the head MLP's forward recompute (with one
      // wave per SIMD a load issued next to its use costs its whole latency: one band tile ahead was 585 us at 128 bands)
      float dall[TBMAX][4];  // FUSED: the ray's d_comp row (unscaled; [R,B] stays in L2), else this sample's d_spectral row
      band_row_load<TBMAX>(dall, FUSED ? io.d_comp + cur.ray * B + 4 * q : io.d_spectral + n * B + 4 * q,
                           FUSED ? (SPEC && ok) : ok, q, TB, B);  // (FUSED: only the specular tail needs the row)
      // FUSED: G[ray][4q .. 4q+3], requested here with the ray index the previous tile's prefetch brought (a load that depends on
      // another load inside the prefetch stalls the wave for a whole memory latency per tile: +14 us at C2) and consumed after the band loop
      v4f g4 = {0.0f, 0.0f, 0.0f, 0.0f};
      if (FUSED) {
        g4 = *reinterpret_cast<const v4f*>(io.mix_g + cur.ray * 16 + 4 * q);
        const float tm = (cur.tm0 + cur.tm1) / 2.0f;  // scale_gradients_by_distance_squared: clamp(t_mid^2, 0, 1)
        cur.ws = ok ? cur.ws * fminf(fmaxf(tm * tm, 0.0f), 1.0f) : 0.0f;
      }
      float dotacc = 0.0f;
      // =================== forward recompute: head MLP, directional hidden layer (feature logits come from the forward) ===
#pragma unroll
      for (int r = 0; r < 4; ++r) in27[0][3 + r] = cur.emb[r];
      float dir28[NT][7];
      if (SPEC) {
        float sh[4];
        sh_slots(sh, cur.d[0], cur.d[1], cur.d[2], q);
#pragma unroll
        for (int s = 0; s < 4; ++s) dir28[0][s] = sh[s];
#pragma unroll
        for (int s = 0; s < 3; ++s) dir28[0][4 + s] = pe[s];
      }
      float a1h[NT][16], a2h[NT][16];
      v4f hd4[NT][1], fl4[NT][1];
      TF_GEMM_F(4, 7, 2, t4, in27, L_H0);
      relu_to<4, NT>(a1h, t4);
      TF_GEMM_F(4, 16, 2, t4, a1h, L_H1);
      relu_to<4, NT>(a2h, t4);
      TF_GEMM_F(1, 16, 2, hd4, a2h, L_H2);
      TF_STAMP(1);
      fl4[0][0] = cur.x0;
      HeadState<NT> hs;
      head_epilogue<NT, SPEC>(hs, hd4, fl4, C, io.temperature, lane);
      float hdir[NT][4];
      if (SPEC) {
        v4f d4[NT][1];
        TF_GEMM_F(1, 7, 2, d4, dir28, L_D0);
        relu_to<1, NT>(hdir, d4);
      }
      STile x27S[2], dirS[2], hdirS[1], mS[1];  // dirS[0]: SH c, dirS[1]: the positional encoding again
      {
        const float pe4[4] = {pe[0], pe[1], pe[2], 0.0f};
        x27S[0] = to_swapped<false>(pe4, ident);
        x27S[1] = to_swapped<false>(&in27[0][3], ident);
        mS[0] = to_swapped<false>(hs.m[0], ident);
        if (SPEC) {
          dirS[0] = to_swapped<false>(&dir28[0][0], ident);
          dirS[1] = x27S[0];
          hdirS[0] = to_swapped<false>(hdir[0], ident);
        }
      }
      TF_STAMP(2);
      // =================== band tiles: mixing and the specular tail (the next tile's gradients are requested a tile ahead) ===
      // (two accumulators each for d m and d hdir, even / odd band tiles: consecutive tiles do not wait for each other's MFMAs)
      v4f dm4[NT][1], dhd4[NT][1], dm4b[NT][1], dhd4b[NT][1];
      dm4[0][0] = dhd4[0][0] = dm4b[0][0] = dhd4b[0][0] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
      float ds1 = 0.0f;
#pragma unroll
      for (int t = 0; t < TBMAX; ++t) {
        if (t < TB) {
          float dsp[NT][4];
#pragma unroll
          for (int r = 0; r < 4; ++r) dsp[0][r] = FUSED ? cur.ws * dall[t][r] : dall[t][r];
          if (!FUSED) {
            // (FUSED: d_spectral[n] = ws_n d_comp[ray(n)] is one vector per RAY times a scalar per sample, and the mixing term is
            // linear -- d m_n = ws_n (d_comp E^T)[ray] and dE = sum_rays (sum_n ws_n m_n)^T d_comp[ray] are formed per ray by
            // field_mix_grad_kernel / field_mix_dE_kernel, nothing of the mixing term is left per sample and band tile)
            gemm_pack<1, 4, NT, 0>((t & 1) ? dm4b : dm4, dsp, wT + td.L[T_MX].off + t * 256, nullptr, lane);
            STile dspS[1];
            dspS[0] = to_swapped<false>(dsp[0], ident);
            dw_pairs<1, 1>(&acc_[SL::A_MX - A0 + t], dspS, mS);  // dE^T[b][c] += sum_n d_spectral[n][b] m[n][c]
          }
          if (SPEC) {
            v4f sc[NT][1];
            gemm_pack<1, 4, NT, 2>(sc, hdir, lds + pd.L[L_D1].off_w + t * 256, lds + pd.L[L_D1].off_b + 16 * t, lane);
            float dzd[NT][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sp = sigmoidf_(sc[0][0][r]);
              if (FUSED) dotacc += dall[t][r] * (hs.s1[0] * sp);
              ds1 += dsp[0][r] * sp;
              dzd[0][r] = dsp[0][r] * hs.s1[0] * sp * (1.0f - sp);
            }
            gemm_pack<1, 4, NT, 0>((t & 1) ? dhd4b : dhd4, dzd, wT + td.L[T_D1].off + t * 256, nullptr, lane);
            STile dzdS[1];
            dzdS[0] = to_swapped<true>(dzd[0], ident, &db_[SL::D_D1 - DB0 + t]);
            dw_pairs<1, 1>(&acc_[SL::A_D1 - A0 + t], dzdS, hdirS);
          }
        }
      }
      TF_STAMP(3);
      dm4[0][0] += dm4b[0][0], dhd4[0][0] += dhd4b[0][0];
      if (FUSED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dotacc += hs.m[0][r] * g4[r];  // classes 4q+r (m is zero from class C on)
        dotacc = xq_sum(dotacc);
        if (ok && q == 0) io.dots[n] = dotacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) dm4[0][0][r] = cur.ws * g4[r];
        // per-ray sums of ws_n m_n for dE: this 16-sample tile's share of its first / last ray, rays strictly inside written directly
        const int rayj = (int)cur.ray;
        const int rf = __builtin_amdgcn_readlane(rayj, 0), rl = __builtin_amdgcn_readlane(rayj, 15);
        const int64_t g = tile * 4 + wave;
        auto row_sum = [&](int ray, float(&out)[4]) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            out[r] = row_sum16((rayj == ray) ? cur.ws * hs.m[0][r] : 0.0f);
          }
        };
#ifndef UMHS_ABL_NO_MS
        if (tile * 64 + wave * 16 < io.n) {
          float a[4];
          row_sum(rf, a);
          if (j == 0) *reinterpret_cast<v4f*>(io.part_ms + (g * 2 + 0) * 16 + 4 * q) = v4f{a[0], a[1], a[2], a[3]};
          if (rl != rf) {
            row_sum(rl, a);
            if (j == 0) *reinterpret_cast<v4f*>(io.part_ms + (g * 2 + 1) * 16 + 4 * q) = v4f{a[0], a[1], a[2], a[3]};
            for (int m = rf + 1; m < rl; ++m) {
              row_sum(m, a);
              if (j == 0) *reinterpret_cast<v4f*>(io.mws16 + (int64_t)m * 16 + 4 * q) = v4f{a[0], a[1], a[2], a[3]};
            }
          }
        }
#endif
      }
      ds1 = xq_sum(ds1);
      // =================== head outputs: sigmoid scalars, temperature softmax, specular gate ==========================
      float dhs[NT][4], dfl[NT][4];
      {
        const float inv_t = 1.0f / io.temperature;
        float da[4], dot = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dmr = dm4[0][0][r];
          const float dsg = dmr * hs.ab[0][r];
          dhs[0][r] = dsg * hs.sg[0][r] * (1.0f - hs.sg[0][r]);
          da[r] = (4 * q + r < C) ? dmr * hs.sg[0][r] : 0.0f;
          dot += hs.ab[0][r] * da[r];
        }
        dot = xq_sum(dot);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          float g = (c < C) ? hs.ab[0][r] * (da[r] - dot) * inv_t : 0.0f;
          if (SPEC && c == C) g = ds1 * hs.s1[0] * (1.0f - hs.s1[0]);
          dfl[0][r] = g;
          if (c >= C) dhs[0][r] = 0.0f;
        }
      }
      if (ok) *reinterpret_cast<v4f*>(io.d_fl + n * 16 + 4 * q) = v4f{dfl[0][0], dfl[0][1], dfl[0][2], dfl[0][3]};
      if (SPEC) {  // mlp_directional hidden layer
        float dz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = hdir[0][r] > 0.0f ? dhd4[0][0][r] : 0.0f;
        STile dzS[1];
        dzS[0] = to_swapped<true>(dz, ident, &db_[SL::D_D0 - DB0]);
        dw_pairs<1, 2>(&acc_[SL::A_D0 - A0], dzS, dirS);
      }
      TF_STAMP(7);
      mlp3_bwd(dhs, a2h, a1h, x27S, &acc_[SL::A_H2 - A0], &acc_[SL::A_H1 - A0], &acc_[SL::A_H0 - A0], &db_[SL::D_H2 - DB0], &db_[SL::D_H1 - DB0], &db_[SL::D_H0 - DB0], T_H2,
               T_H1, T_H0);
      if (ok) *reinterpret_cast<v4f*>(io.d_bo + n * 16 + 4 * q) = dbo4[0][0];
    } else {
      // =================== forward recompute: mlp_base (its outputs are the feature MLP's inputs), feature MLP's hidden layers ===
      float encf[NT][8];
#pragma unroll
      for (int lv = 0; lv < 4; ++lv) encf[0][2 * lv] = cur.e[lv].x, encf[0][2 * lv + 1] = cur.e[lv].y;
      float h[NT][16];
      TF_GEMM_F(4, 8, 2, t4, encf, L_B0);
      relu_to<4, NT>(h, t4);
      v4f bo4[NT][1];
      TF_GEMM_F(1, 16, 2, bo4, h, L_B1);
#pragma unroll
      for (int r = 0; r < 4; ++r) in27[0][3 + r] = bo4[0][0][r];  // slot 0 (sigma_raw) meets a zero weight column
      TF_STAMP(1);
      float a1f[NT][16], a2f[NT][16];
      TF_GEMM_F(4, 7, 2, t4, in27, L_F0);
      relu_to<4, NT>(a1f, t4);
      TF_GEMM_F(4, 16, 2, t4, a1f, L_F1);
      relu_to<4, NT>(a2f, t4);
      TF_STAMP(2);
      STile x27S[2];
      {
        const float pe4[4] = {pe[0], pe[1], pe[2], 0.0f};
        x27S[0] = to_swapped<false>(pe4, ident);
        x27S[1] = to_swapped<false>(&in27[0][3], ident);
      }
      float dfl[NT][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dfl[0][r] = cur.x0[r];
      TF_STAMP(7);
      mlp3_bwd(dfl, a2f, a1f, x27S, &acc_[SL::A_F2 - A0], &acc_[SL::A_F1 - A0], &acc_[SL::A_F0 - A0], &db_[SL::D_F2 - DB0], &db_[SL::D_F1 - DB0], &db_[SL::D_F0 - DB0], T_F2,
               T_F1, T_F0);
      // =================== mlp_base ======================================================================================
      float dzb1[NT][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dzb1[0][r] = ok ? (dbo4[0][0][r] + cur.x1[r]) + cur.demb[r] : 0.0f;
      if (q == 0) {  // slot 0: d sigma_raw = d sigma * selector * exp(clamp(raw, -15, 15))   (trunc_exp backward)
        dzb1[0][0] = cur.dsig * cur.sel * expf(fminf(fmaxf(bo4[0][0][0], -15.0f), 15.0f));
      }
      {
        STile z1[1], hS[4];
        z1[0] = to_swapped<true>(dzb1[0], ident, &db_[SL::D_B1 - DB0]);
        to_swapped_n<4, false>(hS, h[0], ident);
        dw_pairs<1, 4>(&acc_[SL::A_B1 - A0], z1, hS);
      }
      TF_STAMP(15);
      v4f g4[NT][4];
      gemm_pack<4, 4, NT, 1>(g4, dzb1, wT + td.L[T_B1].off, nullptr, lane);
      float dzb0[NT][16];
#pragma unroll
      for (int i = 0; i < 16; ++i) dzb0[0][i] = h[0][i] > 0.0f ? g4[0][i >> 2][i & 3] : 0.0f;
      {
        STile zS[4], eS[2];
        to_swapped_n<4, true>(zS, dzb0[0], ident, &db_[SL::D_B0 - DB0]);
        to_swapped_n<2, false>(eS, encf[0], ident);  // column c = 4q'+u <-> hash feature 8q'+u ([0]) / 8q'+4+u ([1])
        dw_pairs<4, 2>(&acc_[SL::A_B0 - A0], zS, eS);
      }
      TF_STAMP(16);
      v4f de4[NT][2];
      TF_GEMM_T(2, 16, 1, de4, dzb0, T_B0);
      TF_STAMP(17);
      if (ok && io.d_enc) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int lv = 8 * t + 2 * q + rr;  // feature e = 16t+4q+r -> level e>>1, component e&1
            *reinterpret_cast<float2*>(io.d_enc + n * io.sn + (int64_t)lv * io.sl) = make_float2(de4[0][t][2 * rr], de4[0][t][2 * rr + 1]);
          }
      }
    }
    TF_STAMP(18);
#ifdef UMHS_TF_STAMP
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long last = stamp_[0];
      for (int k = 1; k < 19; ++k)
        if (stamp_[k]) g_tf_stamp[PART][k] += stamp_[k] - last, last = stamp_[k];
      g_tf_stamp[PART][0] += 1;
    }
#endif
    cur = nxt;
  }
  // =================== sum the four waves' accumulators through LDS (the pack images are dead), one slab per workgroup ======
#ifdef UMHS_TF_STAMP
  const unsigned long long k_t2 = __builtin_readcyclecounter();
#endif
  float* const slab = slabs + (size_t)blockIdx.x * (SL::NITEMS * 256);
  constexpr int NMINE = NA + NDBP / 4;  // this part's items: its accumulators, then its bias-sum quadruples
#pragma unroll
  for (int c0 = 0; c0 < NMINE; c0 += TF_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TF_CHUNK; ++i) {
      const int it = c0 + i;
      if (it < NMINE) {
        v4f v;
        if (it < NA) {
          v = acc_[it < NA ? it : 0];
        } else {
          const int k = it < NA ? 0 : 4 * (it - NA);
          v = v4f{db_[k], db_[k + 1], db_[k + 2], db_[k + 3]};
        }
        *reinterpret_cast<v4f*>(lds + ((wave * TF_CHUNK + i) * 64 + lane) * 4) = v;
      }
    }
    __syncthreads();
    const int nit = NMINE - c0 < TF_CHUNK ? NMINE - c0 : TF_CHUNK;
    for (int e = tid; e < nit * 64; e += 256) {
      v4f s = *reinterpret_cast<const v4f*>(lds + e * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const v4f*>(lds + (w * TF_CHUNK * 64 + e) * 4);
      const int it = c0 + (e >> 6);  // this part's item -> absolute slab item
      const int abs_item = it < NA ? A0 + it : SL::NACC + SL::dbv0(PART) + (it - NA);
      *reinterpret_cast<v4f*>(slab + (abs_item * 64 + (e & 63)) * 4) = s;
    }
  }
#ifdef UMHS_TF_STAMP
  if (blockIdx.x == 0 && tid == 0) {
    const unsigned long long k_t3 = __builtin_readcyclecounter();
    g_tf_stamp[PART][20] += k_t1 - k_t0, g_tf_stamp[PART][21] += k_t2 - k_t1, g_tf_stamp[PART][22] += k_t3 - k_t2;
  }
#endif
}
#undef TF_GEMM_F
#undef TF_GEMM_T

#include "umhs_field_zip.h"

template <typename K>
static int set_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return UMHS_ERR_UNSUPPORTED;
  static std::atomic<size_t> granted[16];  // per instantiation (a function-local static of a template) x device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
  if (dev >= 0 && granted[dev].load(std::memory_order_relaxed) >= bytes) return UMHS_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
      hipSuccess)
    return UMHS_ERR_LAUNCH;
  if (dev >= 0) granted[dev].store(bytes, std::memory_order_relaxed);
  return UMHS_OK;
}


static unsigned tf_grid(int64_t n) {
  const int64_t ntiles = (n + 63) / 64;
  return (unsigned)(ntiles < 256 ? ntiles : 256);
}

struct TfPart {  // one transpose-free kernel: descriptors rebased to its own LDS image + how to assemble that image
  PackDesc pd;
  TPackDesc td;
  ImgSegs seg_f, seg_t, seg_b;
  BfOffs bo;
  int wt_off, bf_off;
  size_t lds;
};


// ---- the two main kernels of the backward, per band-tile bound TBMAX; each part compiles in its own translation unit (see the top of the file)
// zipped: the three-piece bf16 chain = the kernels with the zipped instruction schedule (umhs_field_zip.h); else the fp32 chain
// (field_bwd_tf_kernel).  Measured (rocprofv3, C2) against the unzipped bf16x3 kernels they replaced: part 0 114.9 -> 110.9 us,
// part 1 114.5 -> 106.9 us; whole backward C3 604 -> 577 us, C5 457 -> 433 us.
struct TfLaunch {
  FieldIO io;
  const float *img, *wT, *bfimg;
  float* slabs;
  unsigned grid;
  umhs_stream_t stream;
};
template <int TBMAX>
int launch_tf_p0z(const TfPart& pt, const TfLaunch& a, bool spec, bool fused);  // zipped (three-piece bf16 chain)
template <int TBMAX>
int launch_tf_p0f(const TfPart& pt, const TfLaunch& a, bool spec, bool fused);  // fp32 chain
template <int TBMAX>
int launch_tf_p1(const TfPart& pt, const TfLaunch& a, bool zipped);

#define TF_ARGS_ a.io, pt.pd, pt.td, a.img, a.wT, pt.seg_f, pt.seg_t, pt.wt_off, a.bfimg, pt.seg_b, pt.bf_off, pt.bo, a.slabs
#define LAUNCH_K_(...)                                                                                          \
  do {                                                                                                          \
    int rc_ = set_lds(__VA_ARGS__, pt.lds);                                                                     \
    if (rc_) return rc_;                                                                                        \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(a.grid), dim3(256), pt.lds, umhs_s(a.stream), TF_ARGS_);             \
    return UMHS_OK;                                                                                             \
  } while (0)
#define INSTANTIATE_(fn_, ...)         \
  template int fn_<2>(__VA_ARGS__);    \
  template int fn_<4>(__VA_ARGS__);    \
  template int fn_<8>(__VA_ARGS__);    \
  template int fn_<12>(__VA_ARGS__);   \
  template int fn_<16>(__VA_ARGS__)
// (instantiated: what run_field_bwd can select -- the zipped part 0 with the specular head up to 4 band tiles, 8 in the folded form
// only; the fp32 chain with it up to 12)
#if UMHS_TU_P0Z
template <int TBMAX, bool FU>
static int launch_tf_p0z_(const TfPart& pt, const TfLaunch& a, bool spec) {
  if (spec) {
    if constexpr (TBMAX < 8 || (TBMAX == 8 && FU)) LAUNCH_K_(field_bwd_tfz0_kernel<true, TBMAX, FU>);
    return UMHS_ERR_UNSUPPORTED;
  }
  if constexpr (TBMAX < 16 || FU) LAUNCH_K_(field_bwd_tfz0_kernel<false, TBMAX, FU>);
  return UMHS_ERR_UNSUPPORTED;
}
template <int TBMAX>
int launch_tf_p0z(const TfPart& pt, const TfLaunch& a, bool spec, bool fused) {
  return fused ? launch_tf_p0z_<TBMAX, true>(pt, a, spec) : launch_tf_p0z_<TBMAX, false>(pt, a, spec);
}
INSTANTIATE_(launch_tf_p0z, const TfPart&, const TfLaunch&, bool, bool);
#endif
#if UMHS_TU_P0F
template <int TBMAX, bool FU>
static int launch_tf_p0f_(const TfPart& pt, const TfLaunch& a, bool spec) {
  if (spec) {
    if constexpr (TBMAX <= 12) LAUNCH_K_(field_bwd_tf_kernel<0, true, TBMAX, FU>);
    return UMHS_ERR_UNSUPPORTED;
  }
  LAUNCH_K_(field_bwd_tf_kernel<0, false, TBMAX, FU>);
}
template <int TBMAX>
int launch_tf_p0f(const TfPart& pt, const TfLaunch& a, bool spec, bool fused) {
  return fused ? launch_tf_p0f_<TBMAX, true>(pt, a, spec) : launch_tf_p0f_<TBMAX, false>(pt, a, spec);
}
INSTANTIATE_(launch_tf_p0f, const TfPart&, const TfLaunch&, bool, bool);
#endif
#if UMHS_TU_P1
template <int TBMAX>
int launch_tf_p1(const TfPart& pt, const TfLaunch& a, bool zipped) {  // (part 1 does not depend on the specular head)
  if (zipped) LAUNCH_K_(field_bwd_tfz1_kernel<TBMAX>);
  LAUNCH_K_(field_bwd_tf_kernel<1, false, TBMAX, false>);
}
INSTANTIATE_(launch_tf_p1, const TfPart&, const TfLaunch&, bool);
#endif
#undef INSTANTIATE_
#undef LAUNCH_K_
#undef TF_ARGS_

#if UMHS_TU_MAIN  // ---- everything below: the small kernels, the forward launchers, the host side of the C-ABI ----------------------

// ---- the mixing term of the folded compositing backward, per RAY (umhs_field_bwd_composited) ---------------------------------
// G[r][c] = sum_b d_comp[r][b] E[c][b]   (c < C, zero above): d m_n = ws_n G[ray(n)]
__global__ __launch_bounds__(256) void field_mix_grad_kernel(const float* __restrict__ d_comp, const float* __restrict__ E, int64_t R,
                                                            int B, int C, float* __restrict__ G) {
  // 16 rays per workgroup: the rays' gradient rows and the endmembers staged in LDS (odd row stride: the 16 class rows a wave reads
  // sit in 16 banks), thread = (ray, class)
  extern __shared__ float sm[];
  const int BS = B | 1;
  float* sE = sm;             // [16][BS]
  float* sD = sm + 16 * BS;   // [16][BS]
  const int tid = threadIdx.x;
  const int64_t ray0 = (int64_t)blockIdx.x * 16;
  for (int i = tid; i < 16 * B; i += 256) {
    const int r = i / B, b = i - r * B;
    sE[r * BS + b] = r < C ? E[(int64_t)r * B + b] : 0.0f;
    sD[r * BS + b] = ray0 + r < R ? d_comp[(ray0 + r) * B + b] : 0.0f;
  }
  __syncthreads();
  const int rr = tid >> 4, c = tid & 15;
  float a0 = 0.0f, a1 = 0.0f;
  int b = 0;
  for (; b + 1 < B; b += 2) a0 += sD[rr * BS + b] * sE[c * BS + b], a1 += sD[rr * BS + b + 1] * sE[c * BS + b + 1];
  if (b < B) a0 += sD[rr * BS + b] * sE[c * BS + b];
  if (ray0 + rr < R) G[(ray0 + rr) * 16 + c] = a0 + a1;
}
// dE[c][b] = sum_r M[r][c] d_comp[r][b] with M[r][c] = sum over the samples of ray r of ws_n m_n[c] (finished here from part 0's tile
// partials, as field_heads_finish_kernel does).  Stage 1: one workgroup per 32 rays -> partial[chunk][c][b]; stage 2 adds the chunks.
constexpr int MIX_CHUNK = 32;
__global__ __launch_bounds__(256) void field_mix_dE_kernel(const float* __restrict__ part_ms, const float* __restrict__ mws16,
                                                          const int64_t* __restrict__ ray_of, const int64_t* __restrict__ pinfo,
                                                          int64_t n, int64_t R, const float* __restrict__ d_comp, int B, int C,
                                                          float* __restrict__ partial) {
  __shared__ float sM[MIX_CHUNK][16];
  const int tid = threadIdx.x;
  const int64_t ray0 = (int64_t)blockIdx.x * MIX_CHUNK;
  {
    const int64_t ray = ray0 + (tid >> 3);
    const int c0 = 2 * (tid & 7);
    float acc[2] = {0.0f, 0.0f};
    if (ray < R) {
      const int64_t s0 = pinfo[2 * ray], cnt = pinfo[2 * ray + 1];
      if (cnt > 0) {
        const int64_t ts = s0 >> 4, te = (s0 + cnt - 1) >> 4;
        const bool first = ray_of[16 * ts] == ray;
        bool inside = false;
        if (ts == te) {
          const int64_t l = 16 * ts + 15 < n ? 16 * ts + 15 : n - 1;
          inside = !first && ray_of[l] != ray;
        }
        if (inside) {
          acc[0] = mws16[ray * 16 + c0], acc[1] = mws16[ray * 16 + c0 + 1];
        } else {
          for (int64_t t = ts; t <= te; ++t) {
            const float2 v = *reinterpret_cast<const float2*>(part_ms + (t * 2 + ((t == ts && !first) ? 1 : 0)) * 16 + c0);
            acc[0] += v.x, acc[1] += v.y;
          }
        }
      }
    }
    sM[tid >> 3][c0] = acc[0], sM[tid >> 3][c0 + 1] = acc[1];
  }
  __syncthreads();
  const int nr = (int)(R - ray0 < MIX_CHUNK ? R - ray0 : MIX_CHUNK);
  for (int i = tid; i < C * B; i += 256) {
    const int c = i / B, b = i - c * B;
    float a0 = 0.0f, a1 = 0.0f;
    int rr = 0;
    for (; rr + 1 < nr; rr += 2) {
      a0 += sM[rr][c] * d_comp[(ray0 + rr) * B + b];
      a1 += sM[rr + 1][c] * d_comp[(ray0 + rr + 1) * B + b];
    }
    if (rr < nr) a0 += sM[rr][c] * d_comp[(ray0 + rr) * B + b];
    partial[(size_t)blockIdx.x * C * B + i] = a0 + a1;
  }
}
__global__ __launch_bounds__(1024) void field_mix_dE_sum_kernel(const float* __restrict__ partial, int nchunks, int CB,
                                                               float* __restrict__ dE) {
  // 64 outputs per workgroup, 16 waves each adding a sixteenth of the chunks (in chunk order), then one fixed-order sum
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, pw = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
  const int per = (nchunks + 15) / 16, k0 = pw * per, k1 = min(nchunks, k0 + per);
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (i < CB) {
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      a0 += partial[(size_t)k * CB + i], a1 += partial[(size_t)(k + 1) * CB + i];
      a2 += partial[(size_t)(k + 2) * CB + i], a3 += partial[(size_t)(k + 3) * CB + i];
    }
    for (; k < k1; ++k) a0 += partial[(size_t)k * CB + i];
  }
  part[pw][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (pw != 0 || i >= CB) return;
  float sacc = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; k += 4) sacc += (part[k][lane] + part[k + 1][lane]) + (part[k + 2][lane] + part[k + 3][lane]);
  dE[i] = sacc;
}

struct GradPtrs {
  float* W[NLAYERS];
  float* b[NLAYERS];
};

// slab -> gradient tensors of the transpose-free kernels.  item -> (layer, to, ti); a swapped input tile's column c of layer kind
// `kind` is reference input column tf_col(kind, ti, c) (or -1: a padding slot).
struct TfMap {
  int nacc, ndb, nitems;
  short layer[128], to[128], ti[128];  // per accumulator item
  short db_layer[64], db_tile[64];     // per bias-sum tile
};
__device__ __forceinline__ int tf_col(int kind, int ti, int c) {
  const int qq = c >> 2, u = c & 3;
  switch (kind) {
    case IN_ENC: return 8 * qq + 4 * ti + u;
    case IN_27: return ti == 0 ? (u < 3 ? 3 * qq + u : -1) : (c >= 1 ? 12 + c - 1 : -1);
    case IN_DIR28: return ti == 0 ? c : (u < 3 ? 16 + 3 * qq + u : -1);
    default: return 16 * ti + c;  // IN_HID64 / IN_HID16 / IN_MIX: natural order
  }
}

// First pass of the slab reduction: workgroup (item, g) adds slabs [g * TF_FOLD, (g + 1) * TF_FOLD) of one item (64 lanes x 4 floats) in a
// fixed order and writes it as item `item` of slab g of `out`.  field_reduce_tf_kernel alone has one workgroup per item, ~76 of them: 76
// CUs pulling 19.5 MB at ~30 GB/s each took 16.7 us; folded by all 256 CUs first, the two passes take half of that.
constexpr int TF_FOLD = 16;
__global__ __launch_bounds__(256) void field_slab_fold_kernel(const float* __restrict__ slabs, int nslabs, int nitems, float* __restrict__ out) {
  __shared__ v4f part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, item = blockIdx.x, g = blockIdx.y;
  const size_t stride = (size_t)nitems * 256, off = (size_t)item * 256 + lane * 4;
  v4f x[TF_FOLD / 4];
#pragma unroll
  for (int k = 0; k < TF_FOLD / 4; ++k) {
    const int sl = g * TF_FOLD + wv * (TF_FOLD / 4) + k;
    x[k] = sl < nslabs ? *reinterpret_cast<const v4f*>(slabs + (size_t)sl * stride + off) : v4f{0.0f, 0.0f, 0.0f, 0.0f};
  }
  part[wv][lane] = (x[0] + x[1]) + (x[2] + x[3]);
  static_assert(TF_FOLD == 16, "four slabs per wave");
  __syncthreads();
  if (wv == 0) *reinterpret_cast<v4f*>(out + (size_t)g * stride + off) = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// 64 outputs per workgroup; its 16 waves each sum a sixteenth of the slabs (8 loads in flight), LDS combines (one thread per output
// walking all 256 slabs was 82 us at 128 bands: 90 workgroups of serial loads)
__global__ __launch_bounds__(1024) void field_reduce_tf_kernel(const float* __restrict__ slabs, int nslabs, TfMap mp, PackDesc pd,
                                                              GradPtrs gp) {
  // Workgroups [0, nacc): one accumulator item (64 lanes x 4 floats = 1 KiB per slab) each -- a wave reads the whole item of a slab
  // with one 16-byte load per lane, 16 waves take a sixteenth of the slabs each (8 loads in flight), LDS combines in a fixed order.
  // Workgroups from nacc on: 64 bias columns each (sum over the slabs and the 4 lane quarters).
  __shared__ v4f part4[16][64];
  const int lane = threadIdx.x & 63, pw = threadIdx.x >> 6;
  const size_t stride = (size_t)mp.nitems * 256;
  const int per = (nslabs + 15) / 16, w0 = pw * per, w1 = min(nslabs, w0 + per);
  if ((int)blockIdx.x < mp.nacc) {
    const int item = blockIdx.x, l = mp.layer[item];
    if (l < 0) return;  // an unused slot (workgroup-uniform)
    const size_t off = (size_t)item * 256 + lane * 4;
    v4f a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    int w = w0;
    for (; w + 7 < w1; w += 8) {
      v4f x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = *reinterpret_cast<const v4f*>(slabs + (size_t)(w + k) * stride + off);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k & 3] += x[k];
    }
    for (; w < w1; ++w) a[0] += *reinterpret_cast<const v4f*>(slabs + (size_t)w * stride + off);
    part4[pw][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (pw != 0) return;
    v4f s4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 16; k += 4) s4 += (part4[k][lane] + part4[k + 1][lane]) + (part4[k + 2][lane] + part4[k + 3][lane]);
    const LayerDesc& L = pd.L[l];
    if (!gp.W[l]) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int out = 16 * mp.to[item] + 4 * (lane >> 4) + r;
      if (l == L_MX) {  // dE^T[b][c]
        const int cls = lane & 15;
        if (out < L.OUT && cls < L.IN) gp.W[l][(size_t)cls * L.OUT + out] = s4[r];
      } else {
        const int in = tf_col(L.kind, mp.ti[item], lane & 15);
        if (out < L.OUT && in >= 0 && in < L.IN) gp.W[l][(size_t)out * L.IN + in] = s4[r];
      }
    }
    return;
  }
  float(*part)[64] = reinterpret_cast<float(*)[64]>(&part4[0][0]);
  const int k = ((int)blockIdx.x - mp.nacc) * 64 + lane, nb = mp.ndb * 16;
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 0.0f;
  const int T = k >> 4, c = k & 15;  // bias tile T (slot in db[]), column c
  if (k < nb) {
    const size_t off = (size_t)(mp.nacc + (T >> 2)) * 256 + (T & 3);
    for (int w = w0; w < w1; ++w) {
      const float* p = slabs + (size_t)w * stride + off;
      a[w & 7] += (p[(c)*4] + p[(16 + c) * 4]) + (p[(32 + c) * 4] + p[(48 + c) * 4]);
    }
  }
  part[pw][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (pw != 0 || k >= nb) return;
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; i += 4) s += (part[i][lane] + part[i + 1][lane]) + (part[i + 2][lane] + part[i + 3][lane]);
  const int l = mp.db_layer[T];
  if (l < 0) return;
  const int o = 16 * mp.db_tile[T] + c;
  if (o < pd.L[l].OUT && gp.b[l]) gp.b[l][o] = s;
}

// =============================================================================================
// host side
// =============================================================================================
static int round16(int x) { return (x + 15) & ~15; }

static int build_pack_desc(const umhs_field_cfg* cfg, const umhs_field_params* p, PackDesc* pd, int* TB_out, int part = 0) {
  // part 0: every layer, 1: mlp_base only, 2: heads only (backward kernels)
  const int B = cfg->n_bands, C = cfg->n_classes, spec = cfg->pred_specular != 0, dens = cfg->density_only != 0;
  const int TB = dens ? 0 : (B + 15) / 16;
  *TB_out = TB;
  auto set = [&](int l, const float* W, const float* b, int kind, int KS, int OT, int OUT, int IN) {
    pd->L[l].W = W, pd->L[l].b = b, pd->L[l].kind = kind, pd->L[l].KS = KS, pd->L[l].OT = OT, pd->L[l].OUT = OUT,
    pd->L[l].IN = IN;
  };
  set(L_B0, p->base_w0, p->base_b0, IN_ENC, 8, 4, 64, 32);
  set(L_B1, p->base_w1, p->base_b1, IN_HID64, 16, 1, 16, 64);
  const int h = dens ? 0 : 1;
  set(L_H0, p->head_w0, p->head_b0, IN_27, 7, 4 * h, 64, 27);
  set(L_H1, p->head_w1, p->head_b1, IN_HID64, 16, 4 * h, 64, 64);
  set(L_H2, p->head_w2, p->head_b2, IN_HID64, 16, 1 * h, C, 64);
  set(L_F0, p->feat_w0, p->feat_b0, IN_27, 7, 4 * h, 64, 27);
  set(L_F1, p->feat_w1, p->feat_b1, IN_HID64, 16, 4 * h, 64, 64);
  set(L_F2, p->feat_w2, p->feat_b2, IN_HID64, 16, 1 * h, spec ? C + 1 : C, 64);
  const int d = (!dens && spec) ? 1 : 0;
  set(L_D0, p->dir_w0, p->dir_b0, IN_DIR28, 7, 1 * d, 16, 28);
  set(L_D1, p->dir_w1, p->dir_b1, IN_HID16, 4, TB * d, B, 16);
  set(L_MX, p->endmembers, nullptr, IN_MIX, 4, TB * h, B, C);  // "W" = E [C][B]: OUT = B, IN = C
  for (int l = 0; l < NLAYERS; ++l)
    if ((part == 1 && l > L_B1) || (part == 2 && l <= L_B1)) pd->L[l].OT = 0;
  int off = 0;
  for (int l = 0; l < NLAYERS; ++l) {
    pd->L[l].off_w = off;
    off += pd->L[l].OT * ((pd->L[l].KS + 3) / 4) * 256;
  }
  pd->total_w = off;
  for (int l = 0; l < NLAYERS; ++l) {
    pd->L[l].off_b = off;
    if (l != L_MX) off += 16 * pd->L[l].OT;
  }
  pd->total = off;
  // pointers required for the layers in use
  for (int l = 0; l < NLAYERS; ++l)
    if (pd->L[l].OT > 0 && (!pd->L[l].W || (l != L_MX && !pd->L[l].b))) return UMHS_ERR_ARG;
  return UMHS_OK;
}

static int check_cfg(const umhs_field_cfg* cfg) {
  if (!cfg) return UMHS_ERR_ARG;
  if (cfg->density_only) return UMHS_OK;
  if (cfg->n_bands < 1 || cfg->n_classes < 1 || !(cfg->temperature > 0.0f)) return UMHS_ERR_ARG;
  if (cfg->n_classes > 15 || cfg->n_bands > 256) return UMHS_ERR_UNSUPPORTED;
  return UMHS_OK;
}

// Raises a kernel's dynamic-LDS limit; remembered per kernel instantiation and device, so the driver call (which showed up as
// a ~6 us bubble in front of every launch it preceded) is made once, not on every step.
// ---- forward on the bf16x3 chain: which layers convert, the compact LDS image, the global bf16x3 image ---------------------------
struct FwdBfPlan {
  PackDesc pd;  // offsets local to the compact fp32 part
  FwdBfArgs args;
  BfPlan bp;
  size_t lds;
};
static bool fwd_bf_plan(const PackDesc& pd_all, FwdBfPlan* fp, bool base_only = false) {
  // base_only: the LDS image of the density half run from the FULL image (umhs_field_base_fwd): mlp_base's two packs and biases
  const int conv[] = {L_B0, L_B1, L_H0, L_H1, L_H2, L_F0, L_F1, L_F2};
  fp->pd = pd_all;
  BfOffs& bo = fp->args.bo;
  for (int l = 0; l < NLAYERS; ++l) bo.f[l] = -1;
  for (int l = 0; l < NTLAYERS; ++l) bo.t[l] = -1;
  fp->bp.n = 0;
  int off = 0;
  for (int l : conv) {
    const LayerDesc& L = pd_all.L[l];
    if (L.OT == 0) continue;
    const int KS4 = (L.KS + 3) / 4;
    fp->bp.c[fp->bp.n++] = BfConv{0, L.off_w, KS4, L.OT, off};
    bo.f[l] = off, off += L.OT * ((KS4 + 1) / 2) * 3 * 256;
  }
  fp->bp.total = off;
  ImgSegs& sf = fp->args.seg_f;
  sf.n = 0;
  bool ok = true;
  auto add = [&](int src, int dst, int len) {
    if (len == 0) return;
    if (sf.n && sf.src[sf.n - 1] + sf.len[sf.n - 1] == src && sf.dst[sf.n - 1] + sf.len[sf.n - 1] == dst) {
      sf.len[sf.n - 1] += len;
      return;
    }
    if (sf.n == 6) {
      ok = false;
      return;
    }
    sf.src[sf.n] = src, sf.dst[sf.n] = dst, sf.len[sf.n] = len, ++sf.n;
  };
  int cur = 0;
  for (int l = 0; l < NLAYERS; ++l) {
    const LayerDesc& L = pd_all.L[l];
    if (bo.f[l] >= 0 || L.OT == 0 || base_only) continue;
    const int len = L.OT * ((L.KS + 3) / 4) * 256;
    add(L.off_w, cur, len);
    fp->pd.L[l].off_w = cur, cur += len;
  }
  for (int l = 0; l < NLAYERS; ++l) {
    const LayerDesc& L = pd_all.L[l];
    if (l == L_MX || L.OT == 0 || (base_only && l > L_B1)) continue;
    add(L.off_b, cur, 16 * L.OT);
    fp->pd.L[l].off_b = cur, cur += 16 * L.OT;
  }
  fp->args.bf_off = (cur + 3) & ~3;
  if (base_only) off = bo.f[L_H0] >= 0 ? bo.f[L_H0] : off;  // mlp_base's packs open the bf16x3 image
  fp->args.seg_b.n = 1, fp->args.seg_b.src[0] = 0, fp->args.seg_b.dst[0] = 0, fp->args.seg_b.len[0] = off;
  fp->args.bf_image = nullptr;
  fp->lds = (size_t)(fp->args.bf_off + off) * 4;
  return ok && fp->lds <= 160 * 1024;
}
// workspace of the forward: [fp32 pack image][bf16x3 image of the converted layers]
static size_t fwd_ws_need(const PackDesc& pd, bool dens) {
  size_t need = (size_t)((pd.total + 63) & ~63) * 4 + 512;
  if (!dens) {
    FwdBfPlan fp;
    fwd_bf_plan(pd, &fp);
    need += (size_t)fp.bp.total * 4;
  }
  return need;
}
static void launch_fwd_packs(const PackDesc& pd, bool dens, float* img, umhs_stream_t stream) {
  PackJob jb = {};
  jb.pd = pd, jb.img = img;
  int n = pd.total;
  if (!dens) {
    FwdBfPlan fp;
    fwd_bf_plan(pd, &fp);
    jb.bp = fp.bp, jb.bf = reinterpret_cast<uint32_t*>(img + ((pd.total + 63) & ~63)), n += fp.bp.total;
  }
  hipLaunchKernelGGL(field_pack_all_kernel, dim3((n + 255) / 256), dim3(256), 0, umhs_s(stream), jb);
}

extern "C" size_t umhs_field_fwd_workspace_bytes(const umhs_field_cfg* cfg) {
  if (check_cfg(cfg)) return 0;
  umhs_field_params dummy = {};
  const float one = 0.0f;
  const float** pp = reinterpret_cast<const float**>(&dummy);
  for (size_t i = 0; i < sizeof(dummy) / sizeof(float*); ++i) pp[i] = &one;  // layout only, never dereferenced
  PackDesc pd;
  int TB;
  if (build_pack_desc(cfg, &dummy, &pd, &TB)) return 0;
  return fwd_ws_need(pd, cfg->density_only != 0);
}

// Builds the forward pack image into the workspace ahead of time (depends on the parameters only); pass pack_ready = 1 and
// the same workspace to umhs_field_fwd afterwards.
extern "C" int umhs_field_fwd_prepare(const umhs_field_cfg* cfg, const umhs_field_params* params, void* workspace,
                                      size_t workspace_bytes, umhs_stream_t stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!params || !workspace) return UMHS_ERR_ARG;
  PackDesc pd;
  int TB;
  rc = build_pack_desc(cfg, params, &pd, &TB);
  if (rc) return rc;
  if (workspace_bytes < fwd_ws_need(pd, cfg->density_only != 0)) return UMHS_ERR_WORKSPACE;
  float* img = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  launch_fwd_packs(pd, cfg->density_only != 0, img, stream);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

static int run_field_fwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc,
                         int64_t stride_n, int64_t stride_l, const float* world_pos, const float* directions,
                         const float* selector, int64_t n, float* sigma, float* sigma_raw, float* emb,
                         float* spectral, float* spectral2, float* specular, float* abundances, float* feat_logits,
                         void* workspace, size_t workspace_bytes, int pack_ready, umhs_stream_t stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!params || !enc || !selector || !sigma || n < 0) return UMHS_ERR_ARG;
  if (((stride_n & 1) || (stride_l & 1) || ((uintptr_t)enc & 7))) return UMHS_ERR_ARG;
  const bool dens = cfg->density_only != 0, spec = cfg->pred_specular != 0;
  if (!dens && (!world_pos || !spectral || (spec && !directions))) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  PackDesc pd;
  int TB;
  rc = build_pack_desc(cfg, params, &pd, &TB);
  if (rc) return rc;
  FieldIO io = {};
  io.enc = enc, io.sn = stride_n, io.sl = stride_l, io.wpos = world_pos, io.dirs = directions, io.sel = selector;
  io.n = n, io.B = cfg->n_bands, io.C = cfg->n_classes, io.TB = TB, io.temperature = cfg->temperature;
  io.sigma = sigma, io.sigma_raw = sigma_raw, io.emb = emb, io.spectral = spectral, io.spectral2 = spectral2;
  io.specular = specular, io.abund = abundances, io.feat_logits = dens ? nullptr : feat_logits;
  const size_t lds_bytes = (size_t)((pd.total + 3) & ~3) * 4;
  // (one shape per case: the bf16x3 chain as 8 waves x 2 sample tiles wherever a prebuilt image exists and its packs fit the LDS, else
  // the fp32 chain as 4 x 2.  The 12 x 1 / 16 x 1 / fp32 8 x 1 forms of round 2 were measured -- 16 x 1: 90 vs 97 us alone, 0.875 vs
  // 0.846 ms inside the step, where it starves the side-stream kernels -- and removed in round 3.)
  const float* image = nullptr;
  if (workspace) {  // optional: prebuilt pack image (without it every workgroup gathers the image itself)
    if (workspace_bytes < fwd_ws_need(pd, dens)) return UMHS_ERR_WORKSPACE;
    float* img = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (!pack_ready) {  // else: umhs_field_fwd_prepare already built it in this workspace
      launch_fwd_packs(pd, dens, img, stream);
      UMHS_CHECK_LAUNCH();
    }
    image = img;
  }
  FwdBfPlan fp;
  FwdBfArgs no_bf = {};
  const bool bf = !dens && image && fwd_bf_plan(pd, &fp);  // no image: each workgroup gathers fp32 packs
  if (bf) fp.args.bf_image = image + ((pd.total + 63) & ~63);
  // samples per workgroup iteration: 16 x NT x waves
  const int tile_samples = bf ? 256 : 128;
  const int64_t ntiles = (n + tile_samples - 1) / tile_samples;
  const int blocks_per_cu = bf ? 1 : (lds_bytes <= 78 * 1024 ? 2 : 1);
  const unsigned grid = (unsigned)(ntiles < 256 * blocks_per_cu ? ntiles : 256 * blocks_per_cu);
  const size_t lds_launch = bf ? fp.lds : lds_bytes;
#define LAUNCH_FWD(S, D, NT_, W_, ...)                                                                                       \
  do {                                                                                                                       \
    rc = set_lds(field_fwd_kernel<S, D, NT_, W_, ##__VA_ARGS__>, lds_launch);                                                 \
    if (rc) return rc;                                                                                                       \
    hipLaunchKernelGGL((field_fwd_kernel<S, D, NT_, W_, ##__VA_ARGS__>), dim3(grid), dim3(64 * W_), lds_launch, umhs_s(stream), \
                       io, bf ? fp.pd : pd, image, bf ? fp.args : no_bf);                                                    \
  } while (0)
  if (dens)
    LAUNCH_FWD(false, true, 2, 4);
  else if (spec) {
    if (bf)
      LAUNCH_FWD(true, false, 2, 8, true);
    else
      LAUNCH_FWD(true, false, 2, 4);
  } else {
    if (bf)
      LAUNCH_FWD(false, false, 2, 8, true);
    else
      LAUNCH_FWD(false, false, 2, 4);
  }
#undef LAUNCH_FWD
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_field_fwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc,
                              int64_t stride_n, int64_t stride_l, const float* world_pos, const float* directions,
                              const float* selector, int64_t n, float* sigma, float* sigma_raw, float* emb,
                              float* spectral, float* spectral2, float* specular, float* abundances, float* feat_logits,
                              void* workspace, size_t workspace_bytes, int pack_ready, umhs_stream_t stream) {
  return run_field_fwd(cfg, params, enc, stride_n, stride_l, world_pos, directions, selector, n, sigma, sigma_raw, emb,
                       spectral, spectral2, specular, abundances, feat_logits, workspace, workspace_bytes, pack_ready, stream);
}

// ---- the forward as two launches with the rendering weights known in between (training step) --------------------------------
// umhs_field_base_fwd: mlp_base only (sigma, sigma_raw, emb) from the FULL configuration's prepared workspace (the same pack
// images umhs_field_heads_fwd uses, built once by umhs_field_fwd_prepare).
extern "C" int umhs_field_base_fwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc, int64_t stride_n,
                                   int64_t stride_l, const float* selector, int64_t n, float* sigma, float* sigma_raw, float* emb,
                                   float* base16, void* workspace, size_t workspace_bytes, int pack_ready, umhs_stream_t stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (cfg->density_only) return UMHS_ERR_UNSUPPORTED;
  if (!params || !enc || !selector || !sigma || !workspace || n < 0) return UMHS_ERR_ARG;
  if ((stride_n & 1) || (stride_l & 1) || ((uintptr_t)enc & 7)) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  PackDesc pd;
  int TB;
  rc = build_pack_desc(cfg, params, &pd, &TB);
  if (rc) return rc;
  if (workspace_bytes < fwd_ws_need(pd, false)) return UMHS_ERR_WORKSPACE;
  float* img = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  if (!pack_ready) {
    launch_fwd_packs(pd, false, img, stream);
    UMHS_CHECK_LAUNCH();
  }
  FwdBfPlan fp;
  if (!fwd_bf_plan(pd, &fp, true)) return UMHS_ERR_UNSUPPORTED;
  fp.args.bf_image = img + ((pd.total + 63) & ~63);
  FieldIO io = {};
  io.enc = enc, io.sn = stride_n, io.sl = stride_l, io.sel = selector, io.n = n, io.B = cfg->n_bands, io.C = cfg->n_classes, io.TB = 0;
  io.temperature = cfg->temperature, io.sigma = sigma, io.sigma_raw = sigma_raw, io.emb = emb, io.bo16 = base16;
  if (base16 && ((uintptr_t)base16 & 15)) return UMHS_ERR_ARG;
  const int64_t ntiles = (n + 127) / 128;
  // One workgroup per CU.  Alone the kernel is fastest with 4 (33 us at 524 k samples; 44 us with 1) -- inside the training
  // step ONE is 45-50 us faster end to end (C3 1.67 vs 1.72 ms, C5 1.49 vs 1.54, three A/B pairs): the bucket histogram of the hash-grid
  // backward runs on the side stream at that moment, and a kernel that fills every CU pushes it under the heads kernel instead.
  const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
  rc = set_lds(field_fwd_kernel<false, true, 2, 4, true>, fp.lds);
  if (rc) return rc;
  hipLaunchKernelGGL((field_fwd_kernel<false, true, 2, 4, true>), dim3(grid), dim3(256), fp.lds, umhs_s(stream), io, fp.pd,
                     (const float*)img, fp.args);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// Finishes the per-ray sums of umhs_field_heads_fwd (see field_fwd_kernel, HEADS): one workgroup per ray.  Adds the ray's tile
// partials in tile order (only its first tile can hold it as that tile's LAST ray, every later tile starts with it; a ray strictly
// inside one tile was written by the heads kernel), then mixes once per ray: spectral2[b] = sum_c (sum_n w_n m_n[c]) E[c][b].
__global__ __launch_bounds__(256) void field_heads_finish_kernel(const float* __restrict__ part_spec, const float* __restrict__ part_m,
                                                                const float* __restrict__ part_ab, const int64_t* __restrict__ ray_of,
                                                                const int64_t* __restrict__ pinfo, int64_t n, int64_t R, int B, int BP,
                                                                int C, const float* __restrict__ E, float* __restrict__ mix16,
                                                                float* __restrict__ c_spectral, float* __restrict__ c_mix,
                                                                float* __restrict__ c_specular, float* __restrict__ cab) {
  // four rays per workgroup (one wave each): four independent chains of dependent loads in flight instead of one
  __shared__ float sM[4][16];
  const int tid = threadIdx.x, sub = tid >> 6, lane = tid & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + sub;
  const bool live = ray < R;
  int64_t ts = 0, te = -1;
  bool inside = false;  // strictly inside one tile: the heads kernel wrote this ray's sums itself
  int az0 = 0;
  if (live) {
    const int64_t s0 = pinfo[2 * ray], cnt = pinfo[2 * ray + 1];
    ts = s0 >> 4, te = cnt > 0 ? (s0 + cnt - 1) >> 4 : ts - 1;
    if (cnt > 0) {
      const bool first = ray_of[16 * ts] == ray;
      if (ts == te) {
        const int64_t l = 16 * ts + 15 < n ? 16 * ts + 15 : n - 1;
        inside = !first && ray_of[l] != ray;
      }
      az0 = first ? 0 : 1;
    }
  }
  if (live && lane < 16) {
    float acc = 0.0f;
    if (inside) {
      acc = mix16[ray * 16 + lane];
    } else {
      for (int64_t t = ts; t <= te; ++t) acc += part_m[(t * 2 + (t == ts ? az0 : 0)) * 16 + lane];
      mix16[ray * 16 + lane] = acc;
    }
    sM[sub][lane] = lane < C ? acc : 0.0f;
  } else if (live && lane < 32 && cab && !inside && lane - 16 < C) {
    float acc = 0.0f;
    for (int64_t t = ts; t <= te; ++t) acc += part_ab[(t * 2 + (t == ts ? az0 : 0)) * 16 + (lane - 16)];
    cab[ray * C + (lane - 16)] = acc;
  }
  __syncthreads();
  if (!live) return;
  for (int b = lane; b < B; b += 64) {
    float mix = 0.0f;
    for (int c = 0; c < C; ++c) mix += sM[sub][c] * E[(int64_t)c * B + b];
    if (part_spec) {  // specular head: spectral = mixing + specular
      float sp = 0.0f;
      if (inside) {
        sp = c_specular[ray * B + b];
      } else {
        for (int64_t t = ts; t <= te; ++t) sp += part_spec[(t * 2 + (t == ts ? az0 : 0)) * BP + b];
        c_specular[ray * B + b] = sp;
      }
      c_mix[ray * B + b] = mix;
      c_spectral[ray * B + b] = mix + sp;
    } else {
      c_spectral[ray * B + b] = mix;
    }
  }
}

// 1 when umhs_field_base_fwd / umhs_field_heads_fwd can serve this configuration (every pack of the bf16x3 forward LDS-resident).
extern "C" int umhs_field_heads_fwd_supported(const umhs_field_cfg* cfg) {
  if (check_cfg(cfg) || cfg->density_only) return 0;
  umhs_field_params dummy = {};
  const float zero = 0.0f;
  const float** pp = reinterpret_cast<const float**>(&dummy);
  for (size_t i = 0; i < sizeof(dummy) / sizeof(float*); ++i) pp[i] = &zero;  // layout only, never dereferenced
  PackDesc pd;
  int TB;
  if (build_pack_desc(cfg, &dummy, &pd, &TB)) return 0;
  FwdBfPlan fp;
  return fwd_bf_plan(pd, &fp, true) && fwd_bf_plan(pd, &fp) ? 1 : 0;
}

// scratch of umhs_field_heads_fwd: [specular partials G*2*BP (specular head only)][w m partials G*2*16][abundance partials G*2*16]
// [mix16 R*16], G = ceil(n / 16) tiles
static size_t heads_scratch_floats(const umhs_field_cfg* cfg, int64_t n, int64_t n_rays, size_t (&off)[4]) {
  const size_t G = (size_t)((n + 15) / 16), BP = 16 * (size_t)((cfg->n_bands + 15) / 16);
  off[0] = 0;
  off[1] = off[0] + (cfg->pred_specular ? G * 2 * BP : 0);
  off[2] = off[1] + G * 2 * 16;
  off[3] = off[2] + G * 2 * 16;
  return off[3] + (size_t)n_rays * 16;
}
extern "C" size_t umhs_field_heads_fwd_scratch_bytes(const umhs_field_cfg* cfg, int64_t n, int64_t n_rays) {
  if (check_cfg(cfg) || cfg->density_only || n < 0 || n_rays < 0) return 0;
  size_t off[4];
  return heads_scratch_floats(cfg, n, n_rays, off) * sizeof(float) + 256;
}

// umhs_field_heads_fwd: everything after mlp_base from its saved outputs (emb [N,15] or the aligned [N,16] rows), with the per-ray
// sums comp_*[r] = sum over the samples n of ray r of weights[n] * stream[n] formed inside the kernel + the finish pass.  No [N,B]
// array exists: the mixing term is summed per ray as w m (16 classes) and multiplied by the endmembers once per ray, the specular
// term per band tile.  ray_indices [N] non-decreasing, packed_info [R,2] = (first sample, count) as umhs_pack_info makes them.
extern "C" int umhs_field_heads_fwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* emb, int emb_stride,
                                    const float* world_pos, const float* directions, int64_t n, const float* weights,
                                    const int64_t* ray_indices, const int64_t* packed_info, int64_t n_rays, float* abundances,
                                    float* feat_logits, float* comp_spectral, float* comp_spectral2, float* comp_specular,
                                    float* comp_abundances, void* scratch, size_t scratch_bytes, void* workspace,
                                    size_t workspace_bytes, int pack_ready, umhs_stream_t stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (cfg->density_only) return UMHS_ERR_UNSUPPORTED;
  const bool spec = cfg->pred_specular != 0;
  if (!params || !params->endmembers || !workspace || n < 0 || n_rays < 0 || !packed_info || !comp_spectral || !scratch)
    return UMHS_ERR_ARG;
  if (spec && (!comp_spectral2 || !comp_specular)) return UMHS_ERR_ARG;
  if (n > 0 && (!emb || !world_pos || !weights || !ray_indices || (spec && !directions))) return UMHS_ERR_ARG;
  if ((emb_stride != 15 && emb_stride != 16) || (emb_stride == 16 && ((uintptr_t)emb & 15))) return UMHS_ERR_ARG;
  if (n_rays == 0) return UMHS_OK;
  PackDesc pd;
  int TB;
  rc = build_pack_desc(cfg, params, &pd, &TB);
  if (rc) return rc;
  if (workspace_bytes < fwd_ws_need(pd, false)) return UMHS_ERR_WORKSPACE;
  if (scratch_bytes < umhs_field_heads_fwd_scratch_bytes(cfg, n, n_rays) || ((uintptr_t)scratch & 15)) return UMHS_ERR_WORKSPACE;
  float* img = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  if (!pack_ready) {
    launch_fwd_packs(pd, false, img, stream);
    UMHS_CHECK_LAUNCH();
  }
  size_t off[4];
  heads_scratch_floats(cfg, n, n_rays, off);
  float* const sc = reinterpret_cast<float*>(scratch);
  float *part_spec = spec ? sc + off[0] : nullptr, *part_m = sc + off[1], *part_ab = sc + off[2], *mix16 = sc + off[3];
  if (n > 0) {
    FwdBfPlan fp;
    if (!fwd_bf_plan(pd, &fp)) return UMHS_ERR_UNSUPPORTED;
    fp.args.bf_image = img + ((pd.total + 63) & ~63);
    FieldIO io = {};
    io.wpos = world_pos, io.dirs = directions, io.n = n, io.B = cfg->n_bands, io.C = cfg->n_classes, io.TB = TB;
    io.temperature = cfg->temperature, io.emb_in = emb, io.abund = abundances, io.feat_logits = feat_logits;
    if (emb_stride == 16) io.bo16_in = emb;
    io.weights = weights, io.ray_of = ray_indices, io.part = part_spec, io.part_m = part_m, io.mix16 = mix16;
    io.part_ab = comp_abundances ? part_ab : nullptr, io.comp_ab = comp_abundances;
    io.comp[0] = comp_spectral, io.comp[1] = comp_spectral2, io.comp[2] = comp_specular, io.n_streams = spec ? 3 : 1;
    const int64_t ntiles = (n + 255) / 256;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    if (spec) {
      rc = set_lds(field_fwd_kernel<true, false, 2, 8, true, true>, fp.lds);
      if (rc) return rc;
      hipLaunchKernelGGL((field_fwd_kernel<true, false, 2, 8, true, true>), dim3(grid), dim3(512), fp.lds, umhs_s(stream), io,
                         fp.pd, (const float*)img, fp.args);
    } else {
      rc = set_lds(field_fwd_kernel<false, false, 2, 8, true, true>, fp.lds);
      if (rc) return rc;
      hipLaunchKernelGGL((field_fwd_kernel<false, false, 2, 8, true, true>), dim3(grid), dim3(512), fp.lds, umhs_s(stream), io,
                         fp.pd, (const float*)img, fp.args);
    }
    UMHS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(field_heads_finish_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, umhs_s(stream), (const float*)part_spec,
                     (const float*)part_m, (const float*)part_ab, ray_indices, packed_info, n, n_rays, cfg->n_bands, 16 * TB, cfg->n_classes,
                     params->endmembers, mix16, comp_spectral, comp_spectral2, comp_specular, comp_abundances);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

struct BwdPlan {
  int TB;
  PackDesc pd_all;
  TPackDesc td;
};

static int build_bwd_plan(const umhs_field_cfg* cfg, const umhs_field_params* p, BwdPlan* pl) {
  const int B = cfg->n_bands, C = cfg->n_classes, spec = cfg->pred_specular != 0;
  int TB;
  int rc = build_pack_desc(cfg, p, &pl->pd_all, &TB, 0);
  if (rc) return rc;
  pl->TB = TB;
  // with the specular head the backward serves up to 192 bands (12 band tiles): beyond, part 0's kernel does not hold its registers
  // (110 dwords spilled at 16 tiles) and has no test behind it -- reported, and the callers keep what they can (known limits)
  if (spec && TB > 12) return UMHS_ERR_UNSUPPORTED;

  TPackDesc* td = &pl->td;
  auto sett = [&](int l, const float* W, int OUT, int IN, int KS, int OT, int rowmap) {
    td->L[l].W = W, td->L[l].OUT = OUT, td->L[l].IN = IN, td->L[l].KS = KS, td->L[l].OT = OT, td->L[l].rowmap = rowmap;
  };
  sett(T_B1, p->base_w1, 16, 64, 4, 4, 0);
  sett(T_B0, p->base_w0, 64, 32, 16, 2, 0);
  sett(T_H2, p->head_w2, C, 64, 4, 4, 0);
  sett(T_H1, p->head_w1, 64, 64, 16, 4, 0);
  sett(T_H0, p->head_w0, 64, 27, 16, 1, 1);
  sett(T_F2, p->feat_w2, spec ? C + 1 : C, 64, 4, 4, 0);
  sett(T_F1, p->feat_w1, 64, 64, 16, 4, 0);
  sett(T_F0, p->feat_w0, 64, 27, 16, 1, 1);
  sett(T_D1, p->dir_w1, B, 16, 4 * TB, spec ? 1 : 0, 0);
  sett(T_MX, p->endmembers, C, B, 4 * TB, 1, 2);  // E [C][B]
  int off = 0;
  for (int l = 0; l < NTLAYERS; ++l) {
    td->L[l].off = off;
    off += td->L[l].OT * ((td->L[l].KS + 3) / 4) * 256;
  }
  td->total = off;

  return UMHS_OK;
}

// ---- transpose-free backward: per-part LDS images and launch ------------------------------------------------------------
static int tf_tbmax(int TB) { return TB <= 2 ? 2 : (TB <= 4 ? 4 : (TB <= 8 ? 8 : (TB <= 12 ? 12 : (TB <= 16 ? 16 : 0)))); }
static int tf_nitems(int tbmax) {
  switch (tbmax) {
    case 2: return TfSlots<2>::NITEMS;
    case 4: return TfSlots<4>::NITEMS;
    case 8: return TfSlots<8>::NITEMS;
    case 12: return TfSlots<12>::NITEMS;
    default: return TfSlots<16>::NITEMS;
  }
}
static size_t bwd_slab_floats(const BwdPlan& pl, int64_t n) {  // the per-workgroup slabs
  const size_t tf = (size_t)(tf_grid(n) + (tf_grid(n) + 15) / 16) * tf_nitems(tf_tbmax(pl.TB) ? tf_tbmax(pl.TB) : 16) * 256;  // + the folded set
  return tf;
}
static int bf_image_dwords(const BwdPlan& pl);
static size_t bwd_workspace_need(const BwdPlan& pl, int64_t n) {
  return ((size_t)pl.td.total + bwd_slab_floats(pl, n) + (size_t)n * 48 + pl.pd_all.total + bf_image_dwords(pl)) * 4 + 4096 + 1024;
}

// The layers whose products run as three-piece bf16 MFMAs (every layer with >= 7 k-steps), in the order of the global bf16x3 image:
// part 0's first (head MLP, directional hidden layer, their transposes), then part 1's.
static const int BF_F0[] = {L_H0, L_H1, L_H2, L_D0}, BF_T0[] = {T_H1, T_H0};
static const int BF_F1[] = {L_B0, L_B1, L_F0, L_F1}, BF_T1[] = {T_B0, T_F1, T_F0};
static void build_bf_plan(const BwdPlan& pl, BfPlan* bp, int (&dst_f)[NLAYERS], int (&dst_t)[NTLAYERS], int (&part_range)[2][2]) {
  for (int l = 0; l < NLAYERS; ++l) dst_f[l] = -1;
  for (int l = 0; l < NTLAYERS; ++l) dst_t[l] = -1;
  bp->n = 0;
  int off = 0;
  auto addf = [&](int l) {
    const LayerDesc& L = pl.pd_all.L[l];
    if (L.OT == 0) return;
    const int KS4 = (L.KS + 3) / 4;
    bp->c[bp->n++] = BfConv{0, L.off_w, KS4, L.OT, off};
    dst_f[l] = off, off += L.OT * ((KS4 + 1) / 2) * 3 * 256;
  };
  auto addt = [&](int l) {
    const TDesc& T = pl.td.L[l];
    if (T.OT == 0) return;
    const int KS4 = (T.KS + 3) / 4;
    bp->c[bp->n++] = BfConv{1, T.off, KS4, T.OT, off};
    dst_t[l] = off, off += T.OT * ((KS4 + 1) / 2) * 3 * 256;
  };
  part_range[0][0] = off;
  for (int l : BF_F0) addf(l);
  for (int l : BF_T0) addt(l);
  part_range[0][1] = part_range[1][0] = off;
  for (int l : BF_F1) addf(l);
  for (int l : BF_T1) addt(l);
  part_range[1][1] = off;
  bp->total = off;
}
static int bf_image_dwords(const BwdPlan& pl) {
  BfPlan bp;
  int df[NLAYERS], dt[NTLAYERS], pr[2][2];
  build_bf_plan(pl, &bp, df, dt, pr);
  return bp.total;
}

static bool tf_part(const BwdPlan& pl, int part, bool bf, TfPart* pp) {
  const int fl0[] = {L_H0, L_H1, L_H2, L_D0, L_D1}, tl0[] = {T_H2, T_H1, T_H0, T_D1, T_MX};
  const int fl1[] = {L_B0, L_B1, L_F0, L_F1}, tl1[] = {T_B1, T_B0, T_F2, T_F1, T_F0};
  const int* fl = part == 0 ? fl0 : fl1;
  const int nfl = part == 0 ? 5 : 4;
  const int* tl = part == 0 ? tl0 : tl1;
  const int ntl = 5;
  pp->pd = pl.pd_all, pp->td = pl.td;
  pp->seg_f.n = pp->seg_t.n = pp->seg_b.n = 0;
  BfPlan bp;
  int dst_f[NLAYERS], dst_t[NTLAYERS], pr[2][2];
  build_bf_plan(pl, &bp, dst_f, dst_t, pr);
  for (int l = 0; l < NLAYERS; ++l) pp->bo.f[l] = (bf && dst_f[l] >= pr[part][0] && dst_f[l] < pr[part][1]) ? dst_f[l] - pr[part][0] : -1;
  for (int l = 0; l < NTLAYERS; ++l) pp->bo.t[l] = (bf && dst_t[l] >= pr[part][0] && dst_t[l] < pr[part][1]) ? dst_t[l] - pr[part][0] : -1;
  bool ok = true;
  auto add = [&](ImgSegs& sg, int src, int dst, int len) {
    if (len == 0) return;
    if (sg.n && sg.src[sg.n - 1] + sg.len[sg.n - 1] == src && sg.dst[sg.n - 1] + sg.len[sg.n - 1] == dst) {
      sg.len[sg.n - 1] += len;
      return;
    }
    if (sg.n == 6) {
      ok = false;
      return;
    }
    sg.src[sg.n] = src, sg.dst[sg.n] = dst, sg.len[sg.n] = len, ++sg.n;
  };
  int cur = 0;
  for (int i = 0; i < nfl; ++i) {  // fp32 weights (of the layers that keep them), then biases, in the kernel's own compact image
    const LayerDesc& L = pl.pd_all.L[fl[i]];
    if (pp->bo.f[fl[i]] >= 0) continue;
    const int len = L.OT * ((L.KS + 3) / 4) * 256;
    add(pp->seg_f, L.off_w, cur, len);
    pp->pd.L[fl[i]].off_w = cur, cur += len;
  }
  for (int i = 0; i < nfl; ++i) {
    const LayerDesc& L = pl.pd_all.L[fl[i]];
    add(pp->seg_f, L.off_b, cur, 16 * L.OT);
    pp->pd.L[fl[i]].off_b = cur, cur += 16 * L.OT;
  }
  pp->wt_off = (cur + 3) & ~3;
  cur = 0;
  for (int i = 0; i < ntl; ++i) {
    const TDesc& T = pl.td.L[tl[i]];
    if (pp->bo.t[tl[i]] >= 0) continue;
    const int len = T.OT * ((T.KS + 3) / 4) * 256;
    add(pp->seg_t, T.off, cur, len);
    pp->td.L[tl[i]].off = cur, cur += len;
  }
  pp->bf_off = (pp->wt_off + cur + 3) & ~3;
  cur = 0;
  if (bf) add(pp->seg_b, pr[part][0], 0, pr[part][1] - pr[part][0]), cur = pr[part][1] - pr[part][0];
  pp->lds = (size_t)(pp->bf_off + cur) * 4;
  if (pp->lds < (size_t)4 * TF_CHUNK * 256 * 4) pp->lds = (size_t)4 * TF_CHUNK * 256 * 4;  // the end-of-launch reduction's rounds
  return ok && pp->lds <= 160 * 1024;
}

template <int TBMAX>
static void fill_tf_map(TfMap* mp, bool spec, int TB) {
  typedef TfSlots<TBMAX> SL;
  static_assert(SL::NACC <= 128 && SL::NDB <= 64, "TfMap tables");
  mp->nacc = SL::NACC, mp->ndb = SL::NDB, mp->nitems = SL::NITEMS;
  for (int i = 0; i < 128; ++i) mp->layer[i] = -1, mp->to[i] = 0, mp->ti[i] = 0;
  for (int i = 0; i < 64; ++i) mp->db_layer[i] = -1, mp->db_tile[i] = 0;
  auto pairs = [&](int base, int layer, int TO, int TI) {
    for (int to = 0; to < TO; ++to)
      for (int ti = 0; ti < TI; ++ti) {
        const int i = base + to * TI + ti;
        mp->layer[i] = (short)layer, mp->to[i] = (short)to, mp->ti[i] = (short)ti;
      }
  };
  pairs(SL::A_B0, L_B0, 4, 2), pairs(SL::A_B1, L_B1, 1, 4);
  pairs(SL::A_H0, L_H0, 4, 2), pairs(SL::A_H1, L_H1, 4, 4), pairs(SL::A_H2, L_H2, 1, 4);
  pairs(SL::A_F0, L_F0, 4, 2), pairs(SL::A_F1, L_F1, 4, 4), pairs(SL::A_F2, L_F2, 1, 4);
  if (spec) pairs(SL::A_D0, L_D0, 1, 2);
  for (int t = 0; t < TB; ++t) {
    if (spec) mp->layer[SL::A_D1 + t] = L_D1, mp->to[SL::A_D1 + t] = (short)t;
    mp->layer[SL::A_MX + t] = L_MX, mp->to[SL::A_MX + t] = (short)t;
  }
  auto bias = [&](int base, int layer, int TO) {
    for (int t = 0; t < TO; ++t) mp->db_layer[base + t] = (short)layer, mp->db_tile[base + t] = (short)t;
  };
  bias(SL::D_B0, L_B0, 4), bias(SL::D_B1, L_B1, 1), bias(SL::D_H0, L_H0, 4), bias(SL::D_H1, L_H1, 4), bias(SL::D_H2, L_H2, 1);
  bias(SL::D_F0, L_F0, 4), bias(SL::D_F1, L_F1, 4), bias(SL::D_F2, L_F2, 1);
  if (spec) bias(SL::D_D0, L_D0, 1), bias(SL::D_D1, L_D1, TB);
}

// The compositing backward folded between the two parts (umhs_field_bwd_composited): part 0 forms d_spectral on the fly and emits the
// dot products, umhs_composite_bwd_dots turns them into d_sigma, part 1 consumes it.
struct BwdComp {
  const float *sigma, *t0, *t1, *weights, *d_comp, *d_acc;
  const int64_t *packed_info, *ray_of;
  int64_t n_rays;
  int grad_scaling;
  float *d_sigma, *dots;
  bool emb16;  // emb is the aligned [N,16] form umhs_field_base_fwd writes (slot 0 = sigma_raw)
  // per-ray mixing term: scratch [G R*16][part_ms tiles*2*16][mws16 R*16][dE partials chunks*C*B]
  float *mix_g, *part_ms, *mws16, *dE_part;
  const float* E;  // endmembers [C][B]
  float* dE;       // their gradient
  int B, C;
};

template <int TBMAX>
static int launch_tf(const BwdPlan& pl, const TfPart (&part)[2], int bf_mask, FieldIO io, bool spec, const float* img, const float* wT,
                     const float* bfimg, float* slabs, const GradPtrs& gp, int64_t n, umhs_stream_t stream, const BwdComp* bc) {
  typedef TfSlots<TBMAX> SL;
  const unsigned grid = tf_grid(n);
  const TfLaunch la = {io, img, wT, bfimg, slabs, grid, stream};
  int rc;
  if (bc) {
    hipLaunchKernelGGL(field_mix_grad_kernel, dim3((unsigned)((bc->n_rays + 15) / 16)), dim3(256), (size_t)32 * (bc->B | 1) * 4,
                       umhs_s(stream), bc->d_comp, bc->E, bc->n_rays, bc->B, bc->C, bc->mix_g);
    rc = (bf_mask & 1) ? launch_tf_p0z<TBMAX>(part[0], la, spec, true) : launch_tf_p0f<TBMAX>(part[0], la, spec, true);
    if (rc) return rc;
    UMHS_CHECK_LAUNCH();
    rc = umhs_composite_bwd_dots(bc->sigma, bc->t0, bc->t1, bc->packed_info, bc->n_rays, n, bc->weights, bc->dots, bc->d_acc,
                                 bc->grad_scaling, bc->d_sigma, stream);
    if (rc) return rc;
  } else {
    rc = (bf_mask & 1) ? launch_tf_p0z<TBMAX>(part[0], la, spec, false) : launch_tf_p0f<TBMAX>(part[0], la, spec, false);
    if (rc) return rc;
  }
  rc = launch_tf_p1<TBMAX>(part[1], la, bf_mask >> 1 & 1);
  if (rc) return rc;
  UMHS_CHECK_LAUNCH();
  TfMap mp;
  fill_tf_map<TBMAX>(&mp, spec, pl.TB);
  if (bc)  // the endmember gradient comes from the per-ray pass below, not from the slabs
    for (int t = 0; t < TBMAX; ++t) mp.layer[SL::A_MX + t] = -1;
  const float* rslabs = slabs;
  int nrs = (int)grid;
  if (grid > 2 * TF_FOLD) {  // fold the slabs 16 : 1 on every CU first (the workspace holds room for the folded set behind the slabs)
    float* folded = slabs + (size_t)grid * SL::NITEMS * 256;
    nrs = (int)((grid + TF_FOLD - 1) / TF_FOLD);
    hipLaunchKernelGGL(field_slab_fold_kernel, dim3(SL::NITEMS, (unsigned)nrs), dim3(256), 0, umhs_s(stream), (const float*)slabs, (int)grid,
                       SL::NITEMS, folded);
    rslabs = folded;
  }
  hipLaunchKernelGGL(field_reduce_tf_kernel, dim3(SL::NACC + (SL::NDB * 16 + 63) / 64), dim3(1024), 0, umhs_s(stream), rslabs, nrs, mp,
                     pl.pd_all, gp);
  UMHS_CHECK_LAUNCH();
  if (bc && bc->dE) {
    const int nchunks = (int)((bc->n_rays + MIX_CHUNK - 1) / MIX_CHUNK), CB = bc->C * bc->B;
    hipLaunchKernelGGL(field_mix_dE_kernel, dim3((unsigned)nchunks), dim3(256), 0, umhs_s(stream), (const float*)bc->part_ms,
                       (const float*)bc->mws16, bc->ray_of, bc->packed_info, n, bc->n_rays, bc->d_comp, bc->B, bc->C, bc->dE_part);
    hipLaunchKernelGGL(field_mix_dE_sum_kernel, dim3((unsigned)((CB + 63) / 64)), dim3(1024), 0, umhs_s(stream),
                       (const float*)bc->dE_part, nchunks, CB, bc->dE);
    UMHS_CHECK_LAUNCH();
  }
  return UMHS_OK;
}

static void launch_bwd_packs(const BwdPlan& pl, float* wT, float* img, float* bfimg, umhs_stream_t stream) {
  PackJob jb = {};
  jb.pd = pl.pd_all, jb.td = pl.td, jb.img = img, jb.wT = wT;
  int df[NLAYERS], dt[NTLAYERS], pr[2][2];
  build_bf_plan(pl, &jb.bp, df, dt, pr);
  if (jb.bp.total > 0) jb.bf = reinterpret_cast<uint32_t*>(bfimg);
  const int n = pl.pd_all.total + pl.td.total + jb.bp.total;
  hipLaunchKernelGGL(field_pack_all_kernel, dim3((n + 255) / 256), dim3(256), 0, umhs_s(stream), jb);
}

// The weight images of the backward (transposed packs + forward pack image) depend on the parameters only: a caller may build
// them ahead of time (e.g. on a side stream during the forward pass) and pass packs_ready = 1 to umhs_field_bwd with the SAME
// workspace.  The parameters must not change in between.
extern "C" int umhs_field_bwd_prepare(const umhs_field_cfg* cfg, const umhs_field_params* params, void* workspace,
                                      size_t workspace_bytes, umhs_stream_t stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (cfg->density_only) return UMHS_ERR_UNSUPPORTED;
  if (!params) return UMHS_ERR_ARG;
  BwdPlan pl;
  rc = build_bwd_plan(cfg, params, &pl);
  if (rc) return rc;
  if (!workspace || workspace_bytes < bwd_workspace_need(pl, 1)) return UMHS_ERR_WORKSPACE;
  float* wT = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float* img = wT + ((pl.td.total + 63) & ~63);
  float* bfimg = img + ((pl.pd_all.total + 63) & ~63);
  launch_bwd_packs(pl, wT, img, bfimg, stream);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" size_t umhs_field_bwd_workspace_bytes(const umhs_field_cfg* cfg, int64_t n) {
  if (check_cfg(cfg) || cfg->density_only || n <= 0) return 0;
  umhs_field_params dummy = {};
  const float one = 0.0f;
  const float** pp = reinterpret_cast<const float**>(&dummy);
  for (size_t i = 0; i < sizeof(dummy) / sizeof(float*); ++i) pp[i] = &one;  // layout only, never dereferenced
  BwdPlan pl;
  if (build_bwd_plan(cfg, &dummy, &pl)) return 0;
  return bwd_workspace_need(pl, n);
}

static int run_field_bwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc,
                         int64_t stride_n, int64_t stride_l, const float* world_pos, const float* directions,
                         const float* selector, const float* sigma_raw, const float* emb, const float* feat_logits,
                         int64_t n, const float* d_sigma, const float* d_spectral, const float* d_emb_ext, float* d_enc,
                         const umhs_field_grads* grads, void* workspace, size_t workspace_bytes, int packs_ready,
                         umhs_stream_t stream, BwdComp* bc) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (cfg->density_only) return UMHS_ERR_UNSUPPORTED;
  if (!params || !enc || !selector || !world_pos || !sigma_raw || !emb || !feat_logits || !d_sigma || (!bc && !d_spectral) || !grads || n < 0)
    return UMHS_ERR_ARG;
  if (bc && (!bc->sigma || !bc->t0 || !bc->t1 || !bc->weights || !bc->d_comp || !bc->packed_info || !bc->ray_of ||
             bc->n_rays < 0))
    return UMHS_ERR_ARG;
  const bool spec = cfg->pred_specular != 0;
  if (spec && !directions) return UMHS_ERR_ARG;
  if ((stride_n & 1) || (stride_l & 1) || ((uintptr_t)enc & 7) || ((uintptr_t)d_enc & 7)) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  BwdPlan pl;
  rc = build_bwd_plan(cfg, params, &pl);
  if (rc) return rc;
  if (!workspace || workspace_bytes < bwd_workspace_need(pl, n)) return UMHS_ERR_WORKSPACE;
  // workspace: [transposed packs][forward pack image][bf16x3 images] (independent of n: umhs_field_bwd_prepare fills them) [slabs][d_bo]
  float* wT = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float* img = wT + ((pl.td.total + 63) & ~63);
  float* bfimg = img + ((pl.pd_all.total + 63) & ~63);
  float* slabs = bfimg + ((bf_image_dwords(pl) + 63) & ~63);
  float* d_bo = slabs + ((bwd_slab_floats(pl, n) + 63) & ~(size_t)63);
  float* d_bo2 = d_bo + (((size_t)n * 16 + 63) & ~(size_t)63);
  float* d_fl = d_bo2 + (((size_t)n * 16 + 63) & ~(size_t)63);
  if (!packs_ready) {
    launch_bwd_packs(pl, wT, img, bfimg, stream);
    UMHS_CHECK_LAUNCH();
  }
  FieldIO io = {};
  io.enc = enc, io.sn = stride_n, io.sl = stride_l, io.wpos = world_pos, io.dirs = directions, io.sel = selector;
  io.n = n, io.B = cfg->n_bands, io.C = cfg->n_classes, io.TB = pl.TB, io.temperature = cfg->temperature;
  io.d_sigma = d_sigma, io.d_spectral = d_spectral, io.d_emb = d_emb_ext, io.d_enc = d_enc;
  io.emb_in = emb, io.sigma_raw_in = sigma_raw, io.d_bo = d_bo;
  GradPtrs gp;
  {
    float* const gw[NLAYERS] = {grads->base_w0, grads->base_w1, grads->head_w0, grads->head_w1, grads->head_w2,
                                grads->feat_w0, grads->feat_w1, grads->feat_w2, grads->dir_w0,  grads->dir_w1,
                                grads->endmembers};
    float* const gb[NLAYERS] = {grads->base_b0, grads->base_b1, grads->head_b0, grads->head_b1, grads->head_b2,
                                grads->feat_b0, grads->feat_b1, grads->feat_b2, grads->dir_b0,  grads->dir_b1,
                                nullptr};
    for (int l = 0; l < NLAYERS; ++l) gp.W[l] = gw[l], gp.b[l] = gb[l];
  }
  // Two kernels (umhs_field_bwd needs the forward's feature logits: part 0 starts from them, part 1 from part 0's d_fl).  The chain runs
  // as three-piece bf16 products (the kernels of umhs_field_zip.h) wherever those hold their registers, else on the fp32 MFMA
  // (field_bwd_tf_kernel); UMHS_BWD_TF=1 (A/B knob) forces the fp32 chain everywhere.  (The LDS-staged kernels of round 1 -- 351 vs
  // 228 us at C2, 1454 vs 577 us at 128 bands -- were removed in round 3.)
  static const bool fp32_chain = getenv("UMHS_BWD_TF") && atoi(getenv("UMHS_BWD_TF")) == 1;
  const int tbmax = tf_tbmax(pl.TB);
  if (tbmax == 0) return UMHS_ERR_UNSUPPORTED;  // more than 256 bands
  {
    // bf16x3: part 1 always (part 0: up to 7 band tiles with the specular head, 8 in the folded form only -- its 8-tile per-sample kernel spills to scratch --
    // and any band count without it)
    const bool p0 = spec ? (tbmax < 8 || (tbmax == 8 && bc != nullptr)) : (tbmax < 16 || bc != nullptr);
    int bf_mask = fp32_chain ? 0 : (2 | (p0 ? 1 : 0));
    TfPart part[2];
    bool ok = true;
    for (int p = 0; p < 2; ++p) {
      if ((bf_mask >> p & 1) && !tf_part(pl, p, true, &part[p])) bf_mask &= ~(1 << p);  // LDS: fall back to the fp32 chain
      if (!(bf_mask >> p & 1)) ok = ok && tf_part(pl, p, false, &part[p]);
    }
    if (ok) {
      io.feat_logits_in = feat_logits, io.d_fl = d_fl;
      if (bc) {
        bc->dots = d_bo2;  // [N] of the (unused here) second hand-off buffer
        io.weights = bc->weights, io.ray_of = bc->ray_of, io.d_comp = bc->d_comp, io.dots = bc->dots;
        io.mix_g = bc->mix_g, io.part_ms = bc->part_ms, io.mws16 = bc->mws16;
        bc->E = params->endmembers, bc->dE = grads->endmembers, bc->B = cfg->n_bands, bc->C = cfg->n_classes;
        if (bc->emb16) io.bo16_in = emb;
        io.t0 = bc->grad_scaling ? bc->t0 : nullptr, io.t1 = bc->grad_scaling ? bc->t1 : nullptr;
      }
      switch (tbmax) {
        case 2: return launch_tf<2>(pl, part, bf_mask, io, spec, img, wT, bfimg, slabs, gp, n, stream, bc);
        case 4: return launch_tf<4>(pl, part, bf_mask, io, spec, img, wT, bfimg, slabs, gp, n, stream, bc);
        case 8: return launch_tf<8>(pl, part, bf_mask, io, spec, img, wT, bfimg, slabs, gp, n, stream, bc);
        case 12: return launch_tf<12>(pl, part, bf_mask, io, spec, img, wT, bfimg, slabs, gp, n, stream, bc);
        default: return launch_tf<16>(pl, part, bf_mask, io, spec, img, wT, bfimg, slabs, gp, n, stream, bc);
      }
    }
  }
  return UMHS_ERR_UNSUPPORTED;  // (a part whose weight images exceed the LDS in either arithmetic: no configuration check_cfg admits)
}

extern "C" int umhs_field_bwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc,
                              int64_t stride_n, int64_t stride_l, const float* world_pos, const float* directions,
                              const float* selector, const float* sigma_raw, const float* emb, const float* feat_logits,
                              int64_t n, const float* d_sigma, const float* d_spectral, const float* d_emb_ext, float* d_enc,
                              const umhs_field_grads* grads, void* workspace, size_t workspace_bytes, int packs_ready,
                              umhs_stream_t stream) {
  return run_field_bwd(cfg, params, enc, stride_n, stride_l, world_pos, directions, selector, sigma_raw, emb, feat_logits, n, d_sigma,
                       d_spectral, d_emb_ext, d_enc, grads, workspace, workspace_bytes, packs_ready, stream, nullptr);
}

// 1 when umhs_field_bwd_composited can serve this configuration (the transpose-free kernels with every pack LDS-resident).
extern "C" int umhs_field_bwd_composited_supported(const umhs_field_cfg* cfg) {
  if (check_cfg(cfg) || cfg->density_only) return 0;
  umhs_field_params dummy = {};
  const float zero = 0.0f;
  const float** pp = reinterpret_cast<const float**>(&dummy);
  for (size_t i = 0; i < sizeof(dummy) / sizeof(float*); ++i) pp[i] = &zero;  // layout only, never dereferenced
  BwdPlan pl;
  if (build_bwd_plan(cfg, &dummy, &pl) || tf_tbmax(pl.TB) == 0) return 0;
  TfPart part;
  return tf_part(pl, 0, false, &part) && tf_part(pl, 1, false, &part) ? 1 : 0;
}

extern "C" size_t umhs_field_bwd_composited_scratch_bytes(const umhs_field_cfg* cfg, int64_t n, int64_t n_rays) {
  if (check_cfg(cfg) || cfg->density_only || n < 0 || n_rays < 0) return 0;
  const size_t G = (size_t)((n + 15) / 16), chunks = (size_t)((n_rays + 31) / 32);
  return ((size_t)n_rays * 32 + G * 32 + chunks * cfg->n_classes * cfg->n_bands) * sizeof(float) + 256;
}

// umhs_field_bwd with the value half of the compositing backward folded in (training step after umhs_field_heads_fwd): instead of
// d_spectral [N,B] it takes the gradient of the per-ray band sums d_comp_spectral [R,B] (+ d_accumulation [R]) and what the
// renderer knows -- sigma, intervals, packed_info, ray_indices, weights -- and returns d_sigma [N] besides everything umhs_field_bwd
// returns.  Per sample: d_spectral[n][b] = scale_n weights[n] d_comp[ray(n)][b] on the fly; spectral[n][b] recomputed for
// dw_n = d_acc[r] + sum_b d_comp[r][b] spectral[n][b]; umhs_composite_bwd_dots; then the density half of the field backward.
// Neither spectral nor d_spectral exists as an [N,B] array.  feat_logits is required.
extern "C" int umhs_field_bwd_composited(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc,
                                         int64_t stride_n, int64_t stride_l, const float* world_pos, const float* directions,
                                         const float* selector, const float* sigma_raw, const float* emb, int emb_stride,
                                         const float* feat_logits, int64_t n, const float* sigma, const float* t_starts, const float* t_ends,
                                         const int64_t* packed_info, int64_t n_rays, const int64_t* ray_indices, const float* weights,
                                         const float* d_comp_spectral, const float* d_accumulation, int grad_scaling, float* d_sigma,
                                         float* d_enc, const umhs_field_grads* grads, void* scratch, size_t scratch_bytes,
                                         void* workspace, size_t workspace_bytes, int packs_ready, umhs_stream_t stream) {
  if (!cfg || n < 0 || n_rays < 0 || !scratch || ((uintptr_t)scratch & 15)) return UMHS_ERR_ARG;
  if (scratch_bytes < umhs_field_bwd_composited_scratch_bytes(cfg, n, n_rays)) return UMHS_ERR_WORKSPACE;
  BwdComp bc = {};
  {
    float* sc = reinterpret_cast<float*>(scratch);
    const size_t G = (size_t)((n + 15) / 16);
    bc.mix_g = sc, bc.part_ms = bc.mix_g + (size_t)n_rays * 16, bc.mws16 = bc.part_ms + G * 2 * 16;
    bc.dE_part = bc.mws16 + (size_t)n_rays * 16;
  }
  bc.sigma = sigma, bc.t0 = t_starts, bc.t1 = t_ends, bc.weights = weights, bc.d_comp = d_comp_spectral, bc.d_acc = d_accumulation;
  bc.packed_info = packed_info, bc.ray_of = ray_indices, bc.n_rays = n_rays, bc.grad_scaling = grad_scaling, bc.d_sigma = d_sigma;
  if ((emb_stride != 15 && emb_stride != 16) || (emb_stride == 16 && ((uintptr_t)emb & 15))) return UMHS_ERR_ARG;
  bc.emb16 = emb_stride == 16;
  return run_field_bwd(cfg, params, enc, stride_n, stride_l, world_pos, directions, selector, sigma_raw, emb, feat_logits, n, d_sigma,
                       nullptr, nullptr, d_enc, grads, workspace, workspace_bytes, packs_ready, stream, &bc);
}
#endif  // UMHS_TU_MAIN
