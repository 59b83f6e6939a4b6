// "Zipped" instruction schedule of the transpose-free field backward (included by umhs_field.hip; needs its helpers).
//
// The transpose-free kernels run ONE wave per SIMD (their dW accumulators fill the AGPR half of the register file), so nothing but
// the wave's own instruction order can overlap the vector ALU with the matrix pipe.  The ISA of field_bwd_tf_kernel shows the two
// strictly taking turns (tools/isa_timeline.py): runs of 50-130 VALU instructions (ReLU masks, the bf16 splits of the three-piece
// products, packs of transposed tiles) with the matrix pipe idle, then runs of 12-48 back-to-back MFMAs with the VALU idle -- per
// 16-sample tile ~1,030 VALU instructions (x 4 cycles of issue for a lone wave) + 404 MFMAs (~7 k cycles of pipe) = the ~14 k
// cycles it measures, PMC MFMA busy 0.32-0.39.  hipcc's scheduler does not interleave them (max-ilp strategy: -3 %), and the dW
// products are opaque inline asm to it anyway.
//
// A tile's work splits into a serial CHAIN (forward recompute -> dX: gemm, then mask / ReLU + bf16 split of its result, then the next
// gemm ...) and OFF-CHAIN work that only has to happen eventually: the 60 transposing MFMAs, the packs of their results, the 120 dW
// MFMAs.  Here the chain's VALU blocks are written as short pinned steps (one bf16 piece of one value pair each, ~4 instructions)
// with a SLOT after every step, and a compile-time list scheduler (make_plan below) assigns the off-chain operations to slots: at
// most one MFMA and one VALU operation per slot, each as early as its operands allow.  So a dW / transposing MFMA issues behind every
// few VALU instructions of the chain, and runs under the next few.  Operations that do not fit into their own tile's slots run in
// the first slots of the wave's next tile (their operand tiles live in registers across the loop edge; a flush follows the loop).
// The order of everything is pinned with __builtin_amdgcn_sched_barrier(0); the chain's gemms run bare between the blocks.
#pragma once

#include <type_traits>

#include "umhs_zip_plan.h"

#define ZSB() __builtin_amdgcn_sched_barrier(0)
#define ZNOP2() asm volatile("s_nop 1")

// one dW product as an instruction of its own (accumulator in AGPRs, see dw_row): the schedule places these between VALU steps
__device__ __forceinline__ void dwm(v4f& acc, const v4s& a, const v4s& b) {
  asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// The same with both operand tiles in AGPRs.  A tile that waits long for its products (the jobs carried into the wave's next tile) does
// not fit the 256 VGPRs next to the chain's working set; parked in an AGPR by the register allocator it comes back through
// v_accvgpr_read RIGHT in front of the inline-asm MFMA, and the two wait states a VALU write needs before an MFMA reads the register
// are not inserted for inline asm (the first version of this kernel computed wrong gradients that way).  Such tiles are moved to
// AGPRs once, when they are packed (pin_agpr), and the MFMA reads them there.
__device__ __forceinline__ void dwm_a(v4f& acc, const v4s& a, const v4s& b) {
  asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b));
}
__device__ __forceinline__ v4s pin_agpr(v4s x) {
  asm volatile("; tile -> AGPR" : "+a"(x));
  return x;
}
// product I of dw_pairs<TO, TI> in dw_row's order: per output row the passes hi*hi, hi*lo, lo*hi over the ti's
template <int I, int TO, int TI, bool AGPR>
__device__ __forceinline__ void dw_one(v4f* __restrict__ acc, const STile* __restrict__ Z, const STile* __restrict__ X) {
  constexpr int to = I / (3 * TI), r = I % (3 * TI), p = r / TI, ti = r % TI;
  if (AGPR)
    dwm_a(acc[to * TI + ti], p == 2 ? Z[to].lo : Z[to].hi, p == 1 ? X[ti].lo : X[ti].hi);
  else
    dwm(acc[to * TI + ti], p == 2 ? Z[to].lo : Z[to].hi, p == 1 ? X[ti].lo : X[ti].hi);
}

// gemm_bf with the B operand's bf16 pieces already split (B[piece][S] = k-slots 8S .. 8S+7 of this lane), one sample tile, and with
// its LDS traffic under control.  In gemm_bf the weight fragments are requested BF_PF fragments ahead in the source, but the ISA shows
// them loaded right in front of their MFMAs with s_waitcnt lgkmcnt(0) in between (under register pressure the scheduler sinks the reads
// to their uses; the masked scheduling barrier there pins only their order): in-kernel stamps give 33-36 cycles per v_mfma_f32_16x16x32
// for every gemm of the backward instead of the pipe's 16 -- half of a tile's 14 k cycles.  Here every step is pinned with a full
// scheduling barrier -- [reads of fragment f + PF] [the six products of fragment f] -- and the first PF fragments and the bias tile are
// requested by gemm_bfp_pre BEFORE the VALU block in front of the gemm (they depend on the lane only), whose ~700 cycles cover them.
#ifndef ZIP_PF
#define ZIP_PF 2
#endif
constexpr int ZPF = ZIP_PF;  // fragments in flight
template <int OT, int K8>
struct GemmPre {
  static constexpr int NF = OT * K8, PF = ZPF < NF ? ZPF : NF;
  v4u A[PF][3];
  v4f bias[OT];
};
template <int OT, int K8>
__device__ __forceinline__ void gemm_frag_load(v4u (&A)[3], const uint32_t* __restrict__ w, int f, int lane) {
  const int t = f % OT, S = f / OT;
  const uint32_t* pw = w + (((t * K8 + S) * 3) * 64 + lane) * 4;
#pragma unroll
  for (int p = 0; p < 3; ++p) A[p] = *reinterpret_cast<const v4u*>(pw + p * 256);
}
template <int OT, int K8, int INIT>
__device__ __forceinline__ void gemm_bfp_pre(GemmPre<OT, K8>& pre, const uint32_t* __restrict__ w, const float* __restrict__ bias, int lane) {
  ZSB();
#pragma unroll
  for (int f = 0; f < GemmPre<OT, K8>::PF; ++f) gemm_frag_load<OT, K8>(pre.A[f], w, f, lane);
  if (INIT == 2) {
#pragma unroll
    for (int t = 0; t < OT; ++t) pre.bias[t] = *reinterpret_cast<const v4f*>(bias + 16 * t + 4 * (lane >> 4));
  }
  ZSB();
}
template <int OT, int K8, int INIT>
__device__ __forceinline__ void gemm_bfp(v4f (&acc)[OT], const v4u (&B)[3][K8], const GemmPre<OT, K8>& pre, const uint32_t* __restrict__ w,
                                         int lane) {
  constexpr int NF = OT * K8, PF = GemmPre<OT, K8>::PF;
  if (INIT != 0) {
#pragma unroll
    for (int t = 0; t < OT; ++t) acc[t] = INIT == 2 ? pre.bias[t] : v4f{0.0f, 0.0f, 0.0f, 0.0f};
  }
  auto mf = [](const v4u& a, const v4u& bb, const v4f& c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, bb), c, 0, 0, 0);
  };
  v4u A[PF + 1][3];
#pragma unroll
  for (int f = 0; f < PF; ++f)
#pragma unroll
    for (int p = 0; p < 3; ++p) A[f][p] = pre.A[f][p];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    ZSB();
    if (f + PF < NF) gemm_frag_load<OT, K8>(A[(f + PF) % (PF + 1)], w, f + PF, lane);
    ZSB();
    const int t = f % OT, S = f / OT, k = f % (PF + 1);
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};  // hh, hm, mh, hl, lh, mm
#pragma unroll
    for (int p = 0; p < 6; ++p) acc[t] = mf(A[k][PA[p]], B[PB[p]][S], acc[t]);
  }
  ZSB();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Part 1 (feature_mlp + mlp_base): the slots of its chain, the off-chain operations, the plan
// ---------------------------------------------------------------------------------------------------------------------------------

template <int TBMAX>
__global__ __launch_bounds__(256, 1) void field_bwd_tfz1_kernel(FieldIO io, PackDesc pd, TPackDesc td, const float* __restrict__ image,
                                                                const float* __restrict__ wT_image, ImgSegs seg_f, ImgSegs seg_t,
                                                                int wt_off, const float* __restrict__ bf_image, ImgSegs seg_b, int bf_off,
                                                                BfOffs bo, float* __restrict__ slabs) {
  using namespace zp1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef TfSlots<TBMAX> SL;
  copy_segs(lds, image, seg_f);
  copy_segs(lds + wt_off, wT_image, seg_t);
  copy_segs(lds + bf_off, bf_image, seg_b);
  __syncthreads();
  const float* const wT = lds + wt_off;
  const uint32_t* const wbf = reinterpret_cast<const uint32_t*>(lds + bf_off);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const v4s ident = ident_frag(lane);
  constexpr int PART = 1;
  constexpr int A0 = SL::acc0(PART), NA = SL::acc1(PART) - A0, DB0 = 4 * SL::dbv0(PART), NDBP = 4 * (SL::dbv1(PART) - SL::dbv0(PART));
  v4f acc_[NA];
  float db_[NDBP];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc_[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NDBP; ++i) db_[i] = 0.0f;
  const int64_t ntiles = (io.n + 63) / 64;
  struct TileIn {
    float w[3];
    float2 e[4];
    v4f x0, x1;  // d_fl, d_bo (from part 0)
    float dsig, sel, demb[4];
  };
  auto fetch = [&](int64_t tile, TileIn& in) {
    int64_t n = tile * 64 + wave * 16 + j;
    const bool ok = n < io.n;
    if (!ok) n = io.n - 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) in.w[s] = io.wpos[3 * n + s];
#pragma unroll
    for (int lv = 0; lv < 4; ++lv) in.e[lv] = *reinterpret_cast<const float2*>(io.enc + n * io.sn + (int64_t)(4 * q + lv) * io.sl);
    const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
    in.x0 = ok ? *reinterpret_cast<const v4f*>(io.d_fl + n * 16 + 4 * q) : z;  // rows past the end: zero upstream gradients
    in.x1 = ok ? *reinterpret_cast<const v4f*>(io.d_bo + n * 16 + 4 * q) : z;
    in.sel = io.sel[n];
    in.dsig = ok ? io.d_sigma[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = 4 * q + r - 1;
      in.demb[r] = (io.d_emb && ok && e >= 0) ? io.d_emb[n * 15 + e] : 0.0f;
    }
  };
  // ---- per-wave state the off-chain operations work on (register arrays, compile-time indices only) ----------------------------
  uint32_t pH[NTILES][2], pM[NTILES][2];  // hi / mid pieces of a tile's two value pairs (the transposing MFMAs' A operands)
  v4f trh[NTILES], trl[NTILES];           // transposed pieces between MFMA and pack
  STile S[NTILES];                        // packed swapped tiles (dW operands)
#pragma unroll
  for (int t = 0; t < NTILES; ++t) S[t].hi = S[t].lo = v4s{0, 0, 0, 0};
  auto run_op = [&](auto idc) __attribute__((always_inline)) {
    constexpr int pos = decltype(idc)::value;
#ifdef UMHS_ABL_NO_OPS  // timing-only ablation (tools/alt_build.py): the chain alone, no transposes / packs / dW products
    if constexpr (false) {
#else
    if constexpr (pos >= 0) {
#endif
      constexpr zip::Op o = ORDER.op[pos];
      if constexpr (o.job == J_TR) {
        constexpr int tile = o.idx >> 1, piece = o.idx & 1;
        const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
        if (piece == 0)
          trh[tile] = MFMA_BF(__builtin_bit_cast(v4s, make_uint2(pH[tile][0], pH[tile][1])), ident, z);
        else
          trl[tile] = MFMA_BF(__builtin_bit_cast(v4s, make_uint2(pM[tile][0], pM[tile][1])), ident, z);
      } else if constexpr (o.job == J_PACK) {
        constexpr int tile = o.idx;
        if constexpr (tile_colsum(tile)) {
          constexpr int slot = tile == T_ZO ? SL::D_F2 : (tile <= T_Z13 ? SL::D_F1 + (tile - T_Z10) : (tile <= T_Z03 ? SL::D_F0 + (tile - T_Z00)
                              : (tile == T_ZB1 ? SL::D_B1 : SL::D_B0 + (tile - T_ZB00))));
          db_[slot - DB0] += ((trh[tile][0] + trh[tile][1]) + (trh[tile][2] + trh[tile][3])) + ((trl[tile][0] + trl[tile][1]) + (trl[tile][2] + trl[tile][3]));
        }
        S[tile].hi = pack_hi16(trh[tile]), S[tile].lo = pack_hi16(trl[tile]);
        if constexpr (tile_in_agpr(tile)) S[tile].hi = pin_agpr(S[tile].hi), S[tile].lo = pin_agpr(S[tile].lo);
      } else {
        constexpr int jd = o.job - J_DW_F2;
        constexpr int abase = jd == 0 ? SL::A_F2 : (jd == 1 ? SL::A_F1 : (jd == 2 ? SL::A_F0 : (jd == 3 ? SL::A_B1 : SL::A_B0)));
        dw_one<o.idx, DW_TO[jd], DW_TI[jd], tile_in_agpr(DW_Z[jd])>(&acc_[abase - A0], &S[DW_Z[jd]], &S[DW_X[jd]]);
      }
    }
  };
  // slot K of a tile: this tile's operations planned for K, and the previous tile's planned for NSLOTS + K
  auto slot = [&](auto kc) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    ZSB();
    run_op(std::integral_constant<int, PLAN.v_at[K + NSLOTS]>{});
    run_op(std::integral_constant<int, PLAN.v_at[K]>{});
    run_op(std::integral_constant<int, PLAN.m_at[K + NSLOTS]>{});
    run_op(std::integral_constant<int, PLAN.m_at[K]>{});
    ZSB();
  };
#define ZSLOT(K_) slot(std::integral_constant<int, (K_)>{})
  // three-piece split of the value pair (x0, x1) in P3 = 5 pinned steps, a slot behind each (a two-piece one: the first P2 = 3)
  static_assert(P3 == 5 && P2 == 3 && M_AT == 2, "the split steps below");
  auto split3 = [&](auto kc, float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    h = cvt_pk_bf(x0, x1);
    ZSLOT(K);
    const float h0 = __uint_as_float(h << 16), h1 = __uint_as_float(h & 0xffff0000u);
    ZSLOT(K + 1);
    const float r0 = x0 - h0, r1 = x1 - h1;
    m = cvt_pk_bf(r0, r1);
    ZSLOT(K + 2);
    const float m0 = __uint_as_float(m << 16), m1 = __uint_as_float(m & 0xffff0000u);
    ZSLOT(K + 3);
    l = cvt_pk_bf(r0 - m0, r1 - m1);
    ZSLOT(K + 4);
  };
  auto split2 = [&](auto kc, float x0, float x1, uint32_t& h, uint32_t& m) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    h = cvt_pk_bf(x0, x1);
    ZSLOT(K);
    const float h0 = __uint_as_float(h << 16), h1 = __uint_as_float(h & 0xffff0000u);
    ZSLOT(K + 1);
    m = cvt_pk_bf(x0 - h0, x1 - h1);
    ZSLOT(K + 2);
  };
#define IC(K_) std::integral_constant<int, (K_)> {}
  TileIn cur, nxt;
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x, cur);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t n = tile * 64 + wave * 16 + j;
    const bool ok = n < io.n;
    if (!ok) n = io.n - 1;
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, nxt);
    ZSB();
#ifdef UMHS_TF_STAMP
    unsigned long long stamp_[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) stamp_[k] = 0;
#endif
    TF_STAMP(0);
    GemmPre<4, 1> preB0;
    gemm_bfp_pre<4, 1, 2>(preB0, wbf + bo.f[L_B0], lds + pd.L[L_B0].off_b, lane);
    // ---- positional encoding, hash features ---------------------------------------------------------------------------------------
    float pe[3];
    pe_slots(pe, cur.w[0], cur.w[1], cur.w[2], q);
    ZSLOT(S_PE);
    float encf[8];
#pragma unroll
    for (int lv = 0; lv < 4; ++lv) encf[2 * lv] = cur.e[lv].x, encf[2 * lv + 1] = cur.e[lv].y;
    ZSLOT(S_PE + 1);
    // (the hi / mid pieces go straight into the tile tables: a transpose may run from the slot behind its tile's last piece on)
#define HP(TILE0_, P_) pH[TILE0_ + ((P_) >> 1)][(P_) & 1]
#define MP(TILE0_, P_) pM[TILE0_ + ((P_) >> 1)][(P_) & 1]
    uint32_t eL[4];
    split3(IC(S_ENC + 0 * P3), encf[0], encf[1], HP(T_E0, 0), MP(T_E0, 0), eL[0]);
    split3(IC(S_ENC + 1 * P3), encf[2], encf[3], HP(T_E0, 1), MP(T_E0, 1), eL[1]);
    split3(IC(S_ENC + 2 * P3), encf[4], encf[5], HP(T_E0, 2), MP(T_E0, 2), eL[2]);
    split3(IC(S_ENC + 3 * P3), encf[6], encf[7], HP(T_E0, 3), MP(T_E0, 3), eL[3]);
    uint32_t iH[4], iM[4], iL[4];  // (in27 pair 0 = (pe0, pe1) is also the first pair of tile X0)
    split3(IC(S_I0), pe[0], pe[1], pH[T_X0][0], pM[T_X0][0], iL[0]);
    iH[0] = pH[T_X0][0], iM[0] = pM[T_X0][0];
    split2(IC(S_X0), pe[2], 0.0f, pH[T_X0][1], pM[T_X0][1]);
    v4f t4[4];
    {
      v4u B[3][1];
      B[0][0] = v4u{HP(T_E0, 0), HP(T_E0, 1), HP(T_E0, 2), HP(T_E0, 3)}, B[1][0] = v4u{MP(T_E0, 0), MP(T_E0, 1), MP(T_E0, 2), MP(T_E0, 3)};
      B[2][0] = v4u{eL[0], eL[1], eL[2], eL[3]};
      ZSB();
      TF_STAMP(1);
      gemm_bfp<4, 1, 2>(t4, B, preB0, wbf + bo.f[L_B0], lane);
      ZSB();
      TF_STAMP(2);
    }
    // ---- base hidden layer -----------------------------------------------------------------------------------------------------------
    GemmPre<1, 2> preB1;
    gemm_bfp_pre<1, 2, 2>(preB1, wbf + bo.f[L_B1], lds + pd.L[L_B1].off_b, lane);
    float h[16];
    uint32_t lo3[8];  // third pieces of the block being split (the gemm behind the block is their only reader)
#define RELU_SPLIT3(P_, BASE_, OUT_, TILE0_)                                                                                  \
  OUT_[2 * P_] = relu1(t4[(2 * P_) >> 2][(2 * P_) & 3]), OUT_[2 * P_ + 1] = relu1(t4[(2 * P_ + 1) >> 2][(2 * P_ + 1) & 3]); \
  split3(IC(BASE_ + P3 * P_), OUT_[2 * P_], OUT_[2 * P_ + 1], HP(TILE0_, P_), MP(TILE0_, P_), lo3[P_])
#define RELU_BLOCK(BASE_, OUT_, TILE0_) \
  RELU_SPLIT3(0, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(1, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(2, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(3, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(4, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(5, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(6, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(7, BASE_, OUT_, TILE0_)
  // B operand of a 64-wide layer from the pieces of the four tiles TILE0_ .. TILE0_ + 3
#define B16(TILE0_)                                                                                              \
  v4u B[3][2];                                                                                                   \
  _Pragma("unroll") for (int S2 = 0; S2 < 2; ++S2) {                                                             \
    B[0][S2] = v4u{HP(TILE0_, 4 * S2), HP(TILE0_, 4 * S2 + 1), HP(TILE0_, 4 * S2 + 2), HP(TILE0_, 4 * S2 + 3)}; \
    B[1][S2] = v4u{MP(TILE0_, 4 * S2), MP(TILE0_, 4 * S2 + 1), MP(TILE0_, 4 * S2 + 2), MP(TILE0_, 4 * S2 + 3)}; \
    B[2][S2] = v4u{lo3[4 * S2], lo3[4 * S2 + 1], lo3[4 * S2 + 2], lo3[4 * S2 + 3]};                             \
  }
    RELU_BLOCK(S_H, h, T_HS0);
    v4f bo4[1];
    {
      B16(T_HS0);
      ZSB();
      TF_STAMP(3);
      gemm_bfp<1, 2, 2>(bo4, B, preB1, wbf + bo.f[L_B1], lane);
      ZSB();
      TF_STAMP(4);
    }
    // ---- feature MLP: input = (positional encoding, base outputs; slot 0 (sigma_raw) meets a zero weight column) -------------------------
    GemmPre<4, 1> preF0;
    gemm_bfp_pre<4, 1, 2>(preF0, wbf + bo.f[L_F0], lds + pd.L[L_F0].off_b, lane);
    split3(IC(S_I1 + 0 * P3), pe[2], bo4[0][0], iH[1], iM[1], iL[1]);
    split3(IC(S_I1 + 1 * P3), bo4[0][1], bo4[0][2], iH[2], iM[2], iL[2]);
    split3(IC(S_I1 + 2 * P3), bo4[0][3], 0.0f, iH[3], iM[3], iL[3]);
    split2(IC(S_X1 + 0 * P2), bo4[0][0], bo4[0][1], pH[T_X1][0], pM[T_X1][0]);
    split2(IC(S_X1 + 1 * P2), bo4[0][2], bo4[0][3], pH[T_X1][1], pM[T_X1][1]);
    {
      v4u B[3][1];
      B[0][0] = v4u{iH[0], iH[1], iH[2], iH[3]}, B[1][0] = v4u{iM[0], iM[1], iM[2], iM[3]}, B[2][0] = v4u{iL[0], iL[1], iL[2], iL[3]};
      ZSB();
      TF_STAMP(5);
      gemm_bfp<4, 1, 2>(t4, B, preF0, wbf + bo.f[L_F0], lane);
      ZSB();
      TF_STAMP(6);
    }
    GemmPre<4, 2> preF1;
    gemm_bfp_pre<4, 2, 2>(preF1, wbf + bo.f[L_F1], lds + pd.L[L_F1].off_b, lane);
    float a1f[16];
    RELU_BLOCK(S_A1, a1f, T_A10);
    {
      B16(T_A10);
      ZSB();
      TF_STAMP(7);
      gemm_bfp<4, 2, 2>(t4, B, preF1, wbf + bo.f[L_F1], lane);
      ZSB();
      TF_STAMP(8);
    }
    float a2f[16];
#define RELU_SPLIT2(P_, BASE_, OUT_, TILE0_)                                                        \
  OUT_[2 * P_] = relu1(t4[(2 * P_) >> 2][(2 * P_) & 3]), OUT_[2 * P_ + 1] = relu1(t4[(2 * P_ + 1) >> 2][(2 * P_ + 1) & 3]); \
  split2(IC(BASE_ + P2 * P_), OUT_[2 * P_], OUT_[2 * P_ + 1], pH[TILE0_ + (P_ >> 1)][P_ & 1], pM[TILE0_ + (P_ >> 1)][P_ & 1])
    RELU_SPLIT2(0, S_A2, a2f, T_A20);
    RELU_SPLIT2(1, S_A2, a2f, T_A20);
    RELU_SPLIT2(2, S_A2, a2f, T_A20);
    RELU_SPLIT2(3, S_A2, a2f, T_A20);
    RELU_SPLIT2(4, S_A2, a2f, T_A20);
    RELU_SPLIT2(5, S_A2, a2f, T_A20);
    RELU_SPLIT2(6, S_A2, a2f, T_A20);
    RELU_SPLIT2(7, S_A2, a2f, T_A20);
    // ---- backward: d(feature logits) from part 0 ----------------------------------------------------------------------------------------
    float dfl[1][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dfl[0][r] = cur.x0[r];
    split2(IC(S_O + 0 * P2), dfl[0][0], dfl[0][1], pH[T_ZO][0], pM[T_ZO][0]);
    split2(IC(S_O + 1 * P2), dfl[0][2], dfl[0][3], pH[T_ZO][1], pM[T_ZO][1]);
    v4f g4[1][4];
    ZSB();
    TF_STAMP(9);
    gemm_pack<4, 4, 1, 1>(g4, dfl, wT + td.L[T_F2].off, nullptr, lane);
    ZSB();
    TF_STAMP(10);
    float dz[16];
#define MASK_SPLIT3(P_, BASE_, ACT_, TILE0_)                                                                                       \
  dz[2 * P_] = ACT_[2 * P_] > 0.0f ? g4[0][(2 * P_) >> 2][(2 * P_) & 3] : 0.0f,                                                     \
         dz[2 * P_ + 1] = ACT_[2 * P_ + 1] > 0.0f ? g4[0][(2 * P_ + 1) >> 2][(2 * P_ + 1) & 3] : 0.0f;                             \
  split3(IC(BASE_ + P3 * P_), dz[2 * P_], dz[2 * P_ + 1], HP(TILE0_, P_), MP(TILE0_, P_), lo3[P_])
#define MASK_BLOCK(BASE_, ACT_, TILE0_) \
  MASK_SPLIT3(0, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(1, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(2, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(3, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(4, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(5, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(6, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(7, BASE_, ACT_, TILE0_)
    GemmPre<4, 2> preT1;
    gemm_bfp_pre<4, 2, 1>(preT1, wbf + bo.t[T_F1], nullptr, lane);
    MASK_BLOCK(S_Z1, a2f, T_Z10);
    {
      B16(T_Z10);
      ZSB();
      TF_STAMP(11);
      gemm_bfp<4, 2, 1>(g4[0], B, preT1, wbf + bo.t[T_F1], lane);
      ZSB();
      TF_STAMP(12);
    }
    GemmPre<1, 2> preT0;
    gemm_bfp_pre<1, 2, 1>(preT0, wbf + bo.t[T_F0], nullptr, lane);
    MASK_BLOCK(S_Z0, a1f, T_Z00);
    v4f dbo4[1];
    {
      B16(T_Z00);
      ZSB();
      TF_STAMP(13);
      gemm_bfp<1, 2, 1>(dbo4, B, preT0, wbf + bo.t[T_F0], lane);
      ZSB();
      TF_STAMP(14);
    }
    // ---- mlp_base ------------------------------------------------------------------------------------------------------------------------
    float dzb1[1][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dzb1[0][r] = ok ? (dbo4[0][r] + cur.x1[r]) + cur.demb[r] : 0.0f;
    if (q == 0)  // slot 0: d sigma_raw = d sigma * selector * exp(clamp(raw, -15, 15))   (trunc_exp backward)
      dzb1[0][0] = cur.dsig * cur.sel * expf(fminf(fmaxf(bo4[0][0], -15.0f), 15.0f));
    ZSLOT(S_B1);
    split2(IC(S_B1 + 1 + 0 * P2), dzb1[0][0], dzb1[0][1], pH[T_ZB1][0], pM[T_ZB1][0]);
    split2(IC(S_B1 + 1 + 1 * P2), dzb1[0][2], dzb1[0][3], pH[T_ZB1][1], pM[T_ZB1][1]);
    ZSB();
    gemm_pack<4, 4, 1, 1>(g4, dzb1, wT + td.L[T_B1].off, nullptr, lane);
    ZSB();
    TF_STAMP(15);
    GemmPre<2, 2> preTB;
    gemm_bfp_pre<2, 2, 1>(preTB, wbf + bo.t[T_B0], nullptr, lane);
    MASK_BLOCK(S_ZB, h, T_ZB00);
    v4f de4[2];
    {
      B16(T_ZB00);
      ZSB();
      TF_STAMP(16);
      gemm_bfp<2, 2, 1>(de4, B, preTB, wbf + bo.t[T_B0], lane);
      ZSB();
      TF_STAMP(17);
    }
    if (ok && io.d_enc) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int lv = 8 * t + 2 * q + rr;  // feature e = 16t+4q+r -> level e>>1, component e&1
          *reinterpret_cast<float2*>(io.d_enc + n * io.sn + (int64_t)lv * io.sl) = make_float2(de4[t][2 * rr], de4[t][2 * rr + 1]);
        }
    }
    // (no chain instructions separate these slots: the wait states a pack's v_perm needs before a product reads the tile are explicit)
    ZSLOT(S_END);
    ZNOP2();
    ZSLOT(S_END + 1);
    ZNOP2();
    ZSLOT(S_END + 2);
    ZNOP2();
    ZSLOT(S_END + 3);
    ZNOP2();
    ZSLOT(S_END + 4);
    ZNOP2();
    ZSLOT(S_END + 5);
    static_assert(N_END == 6, "end slots");
    TF_STAMP(18);
#ifdef UMHS_TF_STAMP
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long last = stamp_[0];
      for (int k = 1; k < 19; ++k)
        if (stamp_[k]) g_tf_stamp[1][k] += stamp_[k] - last, last = stamp_[k];
      g_tf_stamp[1][0] += 1;
    }
#endif
    cur = nxt;
  }
  // ---- operations the last tile left for a next one ---------------------------------------------------------------------------------------
  {
    auto flush = [&](auto kc) __attribute__((always_inline)) {
      constexpr int K = decltype(kc)::value;
      run_op(std::integral_constant<int, PLAN.v_at[K + NSLOTS]>{});
      run_op(std::integral_constant<int, PLAN.m_at[K + NSLOTS]>{});
    };
    auto flush_all = [&](auto self, auto kc) __attribute__((always_inline)) -> void {
      constexpr int K = decltype(kc)::value;
      if constexpr (K + NSLOTS <= PLAN.last) {
        flush(kc);
        self(self, std::integral_constant<int, K + 1>{});
      }
    };
    if ((int64_t)blockIdx.x < ntiles) flush_all(flush_all, std::integral_constant<int, 0>{});
  }
#undef RELU_SPLIT3
#undef RELU_SPLIT2
#undef MASK_SPLIT3
#undef MASK_BLOCK
#undef B16
#undef RELU_BLOCK
#undef HP
#undef MP
#undef IC
#undef ZSLOT
  // ---- sum the four waves' accumulators through LDS, one slab per workgroup (as field_bwd_tf_kernel) -----------------------------------
  float* const slab = slabs + (size_t)blockIdx.x * (SL::NITEMS * 256);
  constexpr int NMINE = NA + NDBP / 4;
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
      const int it = c0 + (e >> 6);
      const int abs_item = it < NA ? A0 + it : SL::NACC + SL::dbv0(PART) + (it - NA);
      *reinterpret_cast<v4f*>(slab + (abs_item * 64 + (e & 63)) * 4) = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Part 0 with its two MLP stretches zipped (plan zp0): head MLP forward recompute | head epilogue, directional layers, band tiles, head
// outputs exactly as field_bwd_tf_kernel<0> orders them | head MLP backward.  bf16x3 chain only.
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool SPEC, int TBMAX, bool FUSED>
__global__ __launch_bounds__(256, 1) void field_bwd_tfz0_kernel(FieldIO io, PackDesc pd, TPackDesc td, const float* __restrict__ image,
                                                                const float* __restrict__ wT_image, ImgSegs seg_f, ImgSegs seg_t,
                                                                int wt_off, const float* __restrict__ bf_image, ImgSegs seg_b, int bf_off,
                                                                BfOffs bo, float* __restrict__ slabs) {
  using namespace zp0;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef TfSlots<TBMAX> SL;
  copy_segs(lds, image, seg_f);
  copy_segs(lds + wt_off, wT_image, seg_t);
  copy_segs(lds + bf_off, bf_image, seg_b);
  __syncthreads();
  const float* const wT = lds + wt_off;
  const uint32_t* const wbf = reinterpret_cast<const uint32_t*>(lds + bf_off);
  constexpr int NT = 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const v4s ident = ident_frag(lane);
  constexpr int PART = 0;
  constexpr int A0 = SL::acc0(PART), NA = SL::acc1(PART) - A0, DB0 = 4 * SL::dbv0(PART), NDBP = 4 * (SL::dbv1(PART) - SL::dbv0(PART));
  v4f acc_[NA];
  float db_[NDBP];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc_[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NDBP; ++i) db_[i] = 0.0f;
  const int C = io.C, B = io.B, TB = io.TB;
  const int64_t ntiles = (io.n + 63) / 64;
  struct TileIn {
    float w[3], d[3];
    v4f x0;  // saved feature logits
    float emb[4];
    float ws, tm0, tm1;  // FUSED: weights[n] (scaled by scale_n once the tile is current), the sample's interval
    int64_t ray;         // FUSED: the sample's ray
  };
  auto fetch = [&](int64_t tile, TileIn& in) {
    int64_t n = tile * 64 + wave * 16 + j;
    const bool ok = n < io.n;
    if (!ok) n = io.n - 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) in.w[s] = io.wpos[3 * n + s];
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
  };
  uint32_t pH[NTILES][2], pM[NTILES][2];
  v4f trh[NTILES], trl[NTILES];
  STile S[NTILES];
#pragma unroll
  for (int t = 0; t < NTILES; ++t) S[t].hi = S[t].lo = v4s{0, 0, 0, 0};
  auto run_op = [&](auto idc) __attribute__((always_inline)) {
    constexpr int pos = decltype(idc)::value;
#ifdef UMHS_ABL_NO_OPS  // timing-only ablation (tools/alt_build.py): the chain alone, no transposes / packs / dW products
    if constexpr (false) {
#else
    if constexpr (pos >= 0) {
#endif
      constexpr zip::Op o = ORDER.op[pos];
      if constexpr (o.job == J_TR) {
        constexpr int tile = o.idx >> 1, piece = o.idx & 1;
        const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
        if (piece == 0)
          trh[tile] = MFMA_BF(__builtin_bit_cast(v4s, make_uint2(pH[tile][0], pH[tile][1])), ident, z);
        else
          trl[tile] = MFMA_BF(__builtin_bit_cast(v4s, make_uint2(pM[tile][0], pM[tile][1])), ident, z);
      } else if constexpr (o.job == J_PACK) {
        constexpr int tile = o.idx;
        if constexpr (tile_colsum(tile)) {
          constexpr int slot = tile == T_ZO ? SL::D_H2 : (tile <= T_Z13 ? SL::D_H1 + (tile - T_Z10) : SL::D_H0 + (tile - T_Z00));
          db_[slot - DB0] += ((trh[tile][0] + trh[tile][1]) + (trh[tile][2] + trh[tile][3])) + ((trl[tile][0] + trl[tile][1]) + (trl[tile][2] + trl[tile][3]));
        }
        S[tile].hi = pack_hi16(trh[tile]), S[tile].lo = pack_hi16(trl[tile]);
        if constexpr (tile_in_agpr(tile)) S[tile].hi = pin_agpr(S[tile].hi), S[tile].lo = pin_agpr(S[tile].lo);
      } else {
        constexpr int jd = o.job - J_DW_H2;
        constexpr int abase = jd == 0 ? SL::A_H2 : (jd == 1 ? SL::A_H1 : SL::A_H0);
        dw_one<o.idx, DW_TO[jd], DW_TI[jd], tile_in_agpr(DW_Z[jd])>(&acc_[abase - A0], &S[DW_Z[jd]], &S[DW_X[jd]]);
      }
    }
  };
  // slot K of a tile: this tile's operations planned for K, and the previous tile's planned for NSLOTS + K
  auto slot = [&](auto kc) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    ZSB();
    run_op(std::integral_constant<int, PLAN.v_at[K + NSLOTS]>{});
    run_op(std::integral_constant<int, PLAN.v_at[K]>{});
    run_op(std::integral_constant<int, PLAN.m_at[K + NSLOTS]>{});
    run_op(std::integral_constant<int, PLAN.m_at[K]>{});
    ZSB();
  };
#define ZSLOT(K_) slot(std::integral_constant<int, (K_)>{})
  // three-piece split of the value pair (x0, x1) in P3 = 5 pinned steps, a slot behind each (a two-piece one: the first P2 = 3)
  static_assert(P3 == 5 && P2 == 3 && M_AT == 2, "the split steps below");
  auto split3 = [&](auto kc, float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    h = cvt_pk_bf(x0, x1);
    ZSLOT(K);
    const float h0 = __uint_as_float(h << 16), h1 = __uint_as_float(h & 0xffff0000u);
    ZSLOT(K + 1);
    const float r0 = x0 - h0, r1 = x1 - h1;
    m = cvt_pk_bf(r0, r1);
    ZSLOT(K + 2);
    const float m0 = __uint_as_float(m << 16), m1 = __uint_as_float(m & 0xffff0000u);
    ZSLOT(K + 3);
    l = cvt_pk_bf(r0 - m0, r1 - m1);
    ZSLOT(K + 4);
  };
  auto split2 = [&](auto kc, float x0, float x1, uint32_t& h, uint32_t& m) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    h = cvt_pk_bf(x0, x1);
    ZSLOT(K);
    const float h0 = __uint_as_float(h << 16), h1 = __uint_as_float(h & 0xffff0000u);
    ZSLOT(K + 1);
    m = cvt_pk_bf(x0 - h0, x1 - h1);
    ZSLOT(K + 2);
  };
#define IC(K_) std::integral_constant<int, (K_)> {}
  TileIn cur, nxt;
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x, cur);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t n = tile * 64 + wave * 16 + j;
    const bool ok = n < io.n;
    if (!ok) n = io.n - 1;
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, nxt);
    ZSB();
#ifdef UMHS_TF_STAMP
    unsigned long long stamp_[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) stamp_[k] = 0;
#endif
    TF_STAMP(0);
      // This tile's upstream gradients, all band tiles: requested here, consumed after the head MLP's forward recompute (with one
      // wave per SIMD a load issued next to its use costs its whole latency: one band tile ahead was 585 us at 128 bands)
      float dall[TBMAX][4];  // FUSED: the ray's d_comp row (unscaled; [R,B] stays in L2), else this sample's d_spectral row
      band_row_load<TBMAX>(dall, FUSED ? io.d_comp + cur.ray * B + 4 * q : io.d_spectral + n * B + 4 * q,
                           FUSED ? (SPEC && ok) : ok, q, TB, B);  // (FUSED: only the specular tail needs the row)
      // FUSED: G[ray][4q .. 4q+3], requested here with the ray index the previous tile's prefetch brought (a load that depends on
      // another load inside the prefetch stalls the wave for a whole memory latency per tile: +14 us at C2) and consumed after the band loop
      v4f gmix = {0.0f, 0.0f, 0.0f, 0.0f};
      if (FUSED) {
        gmix = *reinterpret_cast<const v4f*>(io.mix_g + cur.ray * 16 + 4 * q);
        const float tm = (cur.tm0 + cur.tm1) / 2.0f;  // scale_gradients_by_distance_squared: clamp(t_mid^2, 0, 1)
        cur.ws = ok ? cur.ws * fminf(fmaxf(tm * tm, 0.0f), 1.0f) : 0.0f;
      }
      float dotacc = 0.0f;
    GemmPre<4, 1> preH0;
    gemm_bfp_pre<4, 1, 2>(preH0, wbf + bo.f[L_H0], lds + pd.L[L_H0].off_b, lane);
    // =================== forward recompute: head MLP (zipped) =====================================================================
    float pe[3];
    pe_slots(pe, cur.w[0], cur.w[1], cur.w[2], q);
    ZSLOT(S_PE);
    float in27[NT][7];
#pragma unroll
    for (int s = 0; s < 3; ++s) in27[0][s] = pe[s];
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
    ZSLOT(S_PE + 1);
#define HP(TILE0_, P_) pH[TILE0_ + ((P_) >> 1)][(P_) & 1]
#define MP(TILE0_, P_) pM[TILE0_ + ((P_) >> 1)][(P_) & 1]
    uint32_t iH[4], iM[4], iL[4];  // (in27 pair 0 = (pe0, pe1) is also the first pair of tile X0)
    split3(IC(S_I + 0 * P3), in27[0][0], in27[0][1], pH[T_X0][0], pM[T_X0][0], iL[0]);
    iH[0] = pH[T_X0][0], iM[0] = pM[T_X0][0];
    split3(IC(S_I + 1 * P3), in27[0][2], in27[0][3], iH[1], iM[1], iL[1]);
    split3(IC(S_I + 2 * P3), in27[0][4], in27[0][5], iH[2], iM[2], iL[2]);
    split3(IC(S_I + 3 * P3), in27[0][6], 0.0f, iH[3], iM[3], iL[3]);
    split2(IC(S_X + 0 * P2), in27[0][2], 0.0f, pH[T_X0][1], pM[T_X0][1]);
    split2(IC(S_X + 1 * P2), in27[0][3], in27[0][4], pH[T_X1][0], pM[T_X1][0]);
    split2(IC(S_X + 2 * P2), in27[0][5], in27[0][6], pH[T_X1][1], pM[T_X1][1]);
    v4f t4[4];
    {
      v4u B[3][1];
      B[0][0] = v4u{iH[0], iH[1], iH[2], iH[3]}, B[1][0] = v4u{iM[0], iM[1], iM[2], iM[3]}, B[2][0] = v4u{iL[0], iL[1], iL[2], iL[3]};
      ZSB();
      gemm_bfp<4, 1, 2>(t4, B, preH0, wbf + bo.f[L_H0], lane);
      ZSB();
    }
    uint32_t lo3[8];
#define RELU_SPLIT3(P_, BASE_, OUT_, TILE0_)                                                                                        \
  OUT_[0][2 * P_] = relu1(t4[(2 * P_) >> 2][(2 * P_) & 3]), OUT_[0][2 * P_ + 1] = relu1(t4[(2 * P_ + 1) >> 2][(2 * P_ + 1) & 3]); \
  split3(IC(BASE_ + P3 * P_), OUT_[0][2 * P_], OUT_[0][2 * P_ + 1], HP(TILE0_, P_), MP(TILE0_, P_), lo3[P_])
#define RELU_BLOCK(BASE_, OUT_, TILE0_) \
  RELU_SPLIT3(0, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(1, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(2, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(3, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(4, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(5, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(6, BASE_, OUT_, TILE0_);  \
  RELU_SPLIT3(7, BASE_, OUT_, TILE0_)
#define B16(TILE0_)                                                                                              \
  v4u B[3][2];                                                                                                   \
  _Pragma("unroll") for (int S2 = 0; S2 < 2; ++S2) {                                                             \
    B[0][S2] = v4u{HP(TILE0_, 4 * S2), HP(TILE0_, 4 * S2 + 1), HP(TILE0_, 4 * S2 + 2), HP(TILE0_, 4 * S2 + 3)}; \
    B[1][S2] = v4u{MP(TILE0_, 4 * S2), MP(TILE0_, 4 * S2 + 1), MP(TILE0_, 4 * S2 + 2), MP(TILE0_, 4 * S2 + 3)}; \
    B[2][S2] = v4u{lo3[4 * S2], lo3[4 * S2 + 1], lo3[4 * S2 + 2], lo3[4 * S2 + 3]};                             \
  }
    GemmPre<4, 2> preH1;
    gemm_bfp_pre<4, 2, 2>(preH1, wbf + bo.f[L_H1], lds + pd.L[L_H1].off_b, lane);
    float a1h[NT][16], a2h[NT][16];
    RELU_BLOCK(S_A1, a1h, T_A10);
    {
      B16(T_A10);
      ZSB();
      gemm_bfp<4, 2, 2>(t4, B, preH1, wbf + bo.f[L_H1], lane);
      ZSB();
    }
    GemmPre<1, 2> preH2;
    gemm_bfp_pre<1, 2, 2>(preH2, wbf + bo.f[L_H2], lds + pd.L[L_H2].off_b, lane);
    RELU_BLOCK(S_A2, a2h, T_A20);
    v4f hd4[NT][1], fl4[NT][1];
    {
      B16(T_A20);
      ZSB();
      gemm_bfp<1, 2, 2>(hd4[0], B, preH2, wbf + bo.f[L_H2], lane);
      ZSB();
    }
    TF_STAMP(1);
    // =================== head epilogue, directional layers, band tiles, head outputs: as in field_bwd_tf_kernel<0> ====================
    float dhs[NT][4];
    {
      float dfl[NT][4];
      fl4[0][0] = cur.x0;
      HeadState<NT> hs;
      head_epilogue<NT, SPEC>(hs, hd4, fl4, C, io.temperature, lane);
      float hdir[NT][4];
      if (SPEC) {
        v4f d4[NT][1];
        gemm_bf<1, 7, 1, 2>(d4, dir28, wbf + bo.f[L_D0], lds + pd.L[L_D0].off_b, lane);
        relu_to<1, NT>(hdir, d4);
      }
      TF_STAMP(2);
      STile dirS[2], hdirS[1], mS[1];  // dirS[0]: SH c, dirS[1]: the positional encoding (this stretch's own copy of that tile: the
      {                                // plan transposes the head MLP's input tiles late, next to the dW products that read them)
        mS[0] = to_swapped<false>(hs.m[0], ident);
        if (SPEC) {
          const float pe4[4] = {pe[0], pe[1], pe[2], 0.0f};
          dirS[0] = to_swapped<false>(&dir28[0][0], ident);
          dirS[1] = to_swapped<false>(pe4, ident);
          hdirS[0] = to_swapped<false>(hdir[0], ident);
        }
      }
      
      TF_STAMP(3);
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
      
      TF_STAMP(4);
      dm4[0][0] += dm4b[0][0], dhd4[0][0] += dhd4b[0][0];
      if (FUSED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dotacc += hs.m[0][r] * gmix[r];  // classes 4q+r (m is zero from class C on)
        dotacc = xq_sum(dotacc);
        if (ok && q == 0) io.dots[n] = dotacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) dm4[0][0][r] = cur.ws * gmix[r];
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
      TF_STAMP(5);
      ds1 = xq_sum(ds1);
      // =================== head outputs: sigmoid scalars, temperature softmax, specular gate ==========================
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
      TF_STAMP(6);
      if (SPEC) {  // mlp_directional hidden layer
        float dz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = hdir[0][r] > 0.0f ? dhd4[0][0][r] : 0.0f;
        STile dzS[1];
        dzS[0] = to_swapped<true>(dz, ident, &db_[SL::D_D0 - DB0]);
        dw_pairs<1, 2>(&acc_[SL::A_D0 - A0], dzS, dirS);
      }
    }
    TF_STAMP(7);
    ZSB();
    // =================== backward: head MLP (zipped) ================================================================================
    split2(IC(S_O + 0 * P2), dhs[0][0], dhs[0][1], pH[T_ZO][0], pM[T_ZO][0]);
    split2(IC(S_O + 1 * P2), dhs[0][2], dhs[0][3], pH[T_ZO][1], pM[T_ZO][1]);
    v4f g4[1][4];
    ZSB();
    gemm_pack<4, 4, 1, 1>(g4, dhs, wT + td.L[T_H2].off, nullptr, lane);
    ZSB();
    float dz[16];
#define MASK_SPLIT3(P_, BASE_, ACT_, TILE0_)                                                                                       \
  dz[2 * P_] = ACT_[0][2 * P_] > 0.0f ? g4[0][(2 * P_) >> 2][(2 * P_) & 3] : 0.0f,                                                  \
         dz[2 * P_ + 1] = ACT_[0][2 * P_ + 1] > 0.0f ? g4[0][(2 * P_ + 1) >> 2][(2 * P_ + 1) & 3] : 0.0f;                          \
  split3(IC(BASE_ + P3 * P_), dz[2 * P_], dz[2 * P_ + 1], HP(TILE0_, P_), MP(TILE0_, P_), lo3[P_])
#define MASK_BLOCK(BASE_, ACT_, TILE0_) \
  MASK_SPLIT3(0, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(1, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(2, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(3, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(4, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(5, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(6, BASE_, ACT_, TILE0_);  \
  MASK_SPLIT3(7, BASE_, ACT_, TILE0_)
    GemmPre<4, 2> preT1;
    gemm_bfp_pre<4, 2, 1>(preT1, wbf + bo.t[T_H1], nullptr, lane);
    MASK_BLOCK(S_Z1, a2h, T_Z10);
    {
      B16(T_Z10);
      ZSB();
      gemm_bfp<4, 2, 1>(g4[0], B, preT1, wbf + bo.t[T_H1], lane);
      ZSB();
    }
    GemmPre<1, 2> preT0;
    gemm_bfp_pre<1, 2, 1>(preT0, wbf + bo.t[T_H0], nullptr, lane);
    MASK_BLOCK(S_Z0, a1h, T_Z00);
    v4f dbo4[1];
    {
      B16(T_Z00);
      ZSB();
      gemm_bfp<1, 2, 1>(dbo4, B, preT0, wbf + bo.t[T_H0], lane);
      ZSB();
    }
    TF_STAMP(8);
    if (ok) *reinterpret_cast<v4f*>(io.d_bo + n * 16 + 4 * q) = dbo4[0];
    ZSLOT(S_END);
    ZNOP2();
    ZSLOT(S_END + 1);
    ZNOP2();
    ZSLOT(S_END + 2);
    ZNOP2();
    ZSLOT(S_END + 3);
    ZNOP2();
    ZSLOT(S_END + 4);
    ZNOP2();
    ZSLOT(S_END + 5);
    static_assert(N_END == 6, "end slots");
    TF_STAMP(9);
#ifdef UMHS_TF_STAMP
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long last = stamp_[0];
      for (int k = 1; k < 19; ++k)
        if (stamp_[k]) g_tf_stamp[0][k] += stamp_[k] - last, last = stamp_[k];
      g_tf_stamp[0][0] += 1;
    }
#endif
    cur = nxt;
  }
  // ---- operations the last tile left for a next one ---------------------------------------------------------------------------------------
  {
    auto flush = [&](auto kc) __attribute__((always_inline)) {
      constexpr int K = decltype(kc)::value;
      run_op(std::integral_constant<int, PLAN.v_at[K + NSLOTS]>{});
      run_op(std::integral_constant<int, PLAN.m_at[K + NSLOTS]>{});
    };
    auto flush_all = [&](auto self, auto kc) __attribute__((always_inline)) -> void {
      constexpr int K = decltype(kc)::value;
      if constexpr (K + NSLOTS <= PLAN.last) {
        flush(kc);
        self(self, std::integral_constant<int, K + 1>{});
      }
    };
    if ((int64_t)blockIdx.x < ntiles) flush_all(flush_all, std::integral_constant<int, 0>{});
  }
#undef RELU_SPLIT3
#undef RELU_SPLIT2
#undef MASK_SPLIT3
#undef MASK_BLOCK
#undef B16
#undef RELU_BLOCK
#undef HP
#undef MP
#undef IC
#undef ZSLOT
  // ---- sum the four waves' accumulators through LDS, one slab per workgroup (as field_bwd_tf_kernel) -----------------------------------
  float* const slab = slabs + (size_t)blockIdx.x * (SL::NITEMS * 256);
  constexpr int NMINE = NA + NDBP / 4;
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
      const int it = c0 + (e >> 6);
      const int abs_item = it < NA ? A0 + it : SL::NACC + SL::dbv0(PART) + (it - NA);
      *reinterpret_cast<v4f*>(slab + (abs_item * 64 + (e & 63)) * 4) = s;
    }
  }
}
