// method="rgb" (the reference's default method; umhs_field.py:280-294 = nerfstudio's NerfactoField): the two small MLPs as gfx950 kernels.
//   base : hash features [N,32] -> 64 (ReLU) -> 16 : density = trunc_exp(out0) * selector, embedding = out1..15      (mlp_base, :51,:320-327)
//   head : [SH16((d+1)/2) | embedding15] -> 64 (ReLU) -> 64 (ReLU) -> 3, Sigmoid                                      (NerfactoField.mlp_head)
// Not the hot path (SURVEY 8 scopes the spectral methods; BASELINE configs[0] is "plumbing"), so the kernels are plain: exact fp32 on
// v_mfma_f32_16x16x4_f32, one wave per 16-sample tile, the "samples on lanes" chaining of umhs_field.hip in its simplest form -- every
// GEMM is computed transposed, Y^T[out][sample] = W[out][in] X^T[in][sample], weights as the A operand and activations as the B operand:
// a result tile has its 16 samples on lane & 15 and its outputs (4 * (lane >> 4) + r) in register r, which IS the B-operand shape of the
// next layer once that layer walks its contraction index in the order (tile, r); the weights are read from LDS in that order.  The
// backward recomputes the forward per tile, runs dX the same way (A = W^T) and forms dW = dZ X^T (a contraction over the samples of the
// tile = over lanes) from [feature][sample] tiles staged in wave-private LDS; every wave keeps its dW / db in registers across all its
// tiles and writes one slab at the end, rgb_mlp_reduce_kernel sums the slabs in a fixed order (bitwise reproducible).
#include "umhs_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

constexpr int H = 64;      // hidden width (NerfactoField: hidden_dim = hidden_dim_color = 64)
constexpr int KIN = 32;    // input width of both MLPs (base: 16 levels x 2; head: 16 SH + 15 embedding + 1 zero)
constexpr int LDW1 = KIN + 1, LDWH = H + 1;  // LDS row strides of the weight images (odd: the 16 rows of an A operand hit 16 banks)

struct RgbMlp {        // torch Linear layout: weight [out][in] row-major, bias [out]
  const float *w0, *b0;  // [64][in0]
  const float *w1, *b1;  // head: [64][64]; base: [16][64]
  const float *w2, *b2;  // head: [3][64];  base: unused
  int in0;               // 32 (base) or 31 (head)
};

__device__ __forceinline__ v4f mfma(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS weight images: w[out][ld] with zero padding up to (rows, cols); bias images padded with zeros
__device__ __forceinline__ void load_w(float* dst, int ld, const float* src, int rows, int cols, int rows_pad, int cols_pad) {
  for (int i = threadIdx.x; i < rows_pad * cols_pad; i += blockDim.x) {
    const int r = i / cols_pad, c = i - r * cols_pad;
    dst[r * ld + c] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.0f;
  }
}
__device__ __forceinline__ void load_b(float* dst, const float* src, int n, int n_pad) {
  for (int i = threadIdx.x; i < n_pad; i += blockDim.x) dst[i] = i < n ? src[i] : 0.0f;
}

// First layer: contraction index in natural order.  x: wave-private LDS tile [16 samples][KIN + 1].
template <int OT>
__device__ __forceinline__ void gemm_first(v4f (&acc)[OT], const float* w, const float* bias, const float* x, int lane) {
  const int q = lane >> 4, m = lane & 15;
#pragma unroll
  for (int t = 0; t < OT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = bias[16 * t + 4 * q + r];
  }
#pragma unroll
  for (int s = 0; s < KIN / 4; ++s) {
    const float b = x[m * LDW1 + 4 * s + q];
#pragma unroll
    for (int t = 0; t < OT; ++t) acc[t] = mfma(w[(16 * t + m) * LDW1 + 4 * s + q], b, acc[t]);
  }
}
// Hidden / output layer: the input is the previous layer's result tiles (IT tiles of 16 features), walked in (tile, r) order.
template <int OT, int IT>
__device__ __forceinline__ void gemm_next(v4f (&acc)[OT], const float* w, const float* bias, const v4f (&h)[IT], int lane) {
  const int q = lane >> 4, m = lane & 15;
#pragma unroll
  for (int t = 0; t < OT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = bias[16 * t + 4 * q + r];
  }
#pragma unroll
  for (int ti = 0; ti < IT; ++ti) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int t = 0; t < OT; ++t) acc[t] = mfma(w[(16 * t + m) * LDWH + 16 * ti + 4 * q + r], h[ti][r], acc[t]);
    }
  }
}
// dX^T[in][sample] = sum_out W[out][in] dZ^T[out][sample]: A = W^T (lane holds W[out = 16 to + 4 q + r][in = 16 t + m]), no bias
template <int IT, int OT>
__device__ __forceinline__ void gemm_tr(v4f (&dx)[IT], const float* w, int ld, const v4f (&dz)[OT], int lane) {
  const int q = lane >> 4, m = lane & 15;
#pragma unroll
  for (int t = 0; t < IT; ++t) dx[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int to = 0; to < OT; ++to) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int t = 0; t < IT; ++t) dx[t] = mfma(w[(16 * to + 4 * q + r) * ld + 16 * t + m], dz[to][r], dx[t]);
    }
  }
}
// stage a "samples on lanes" tile set as [feature][sample] rows (17-float rows: conflict-free writes and transposed reads)
template <int T>
__device__ __forceinline__ void stage(float* dst, const v4f (&v)[T], int lane) {
  const int q = lane >> 4, m = lane & 15;
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(16 * t + 4 * q + r) * 17 + m] = v[t][r];
  }
  __builtin_amdgcn_wave_barrier();  // (wave-private tile: LDS operations of a wave execute in order; this only pins the compiler's order)
}
// dW[out][in] += sum_samples dZ[out][sample] X[in][sample]: A = dZ rows (lane: out = 16 to + m, sample = 4 s + q), B = X rows
template <int OT, int IT>
__device__ __forceinline__ void dw_acc(v4f (&dw)[OT][IT], const float* dz_rows, const float* x_rows, int lane) {
  const int q = lane >> 4, m = lane & 15;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float a[OT], b[IT];
#pragma unroll
    for (int to = 0; to < OT; ++to) a[to] = dz_rows[(16 * to + m) * 17 + 4 * s + q];
#pragma unroll
    for (int ti = 0; ti < IT; ++ti) b[ti] = x_rows[(16 * ti + m) * 17 + 4 * s + q];
#pragma unroll
    for (int to = 0; to < OT; ++to) {
#pragma unroll
      for (int ti = 0; ti < IT; ++ti) dw[to][ti] = mfma(a[to], b[ti], dw[to][ti]);
    }
  }
}
template <int T>
__device__ __forceinline__ void relu(v4f (&v)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[t][r] = fmaxf(v[t][r], 0.0f);
  }
}
// sum over the 16 sample lanes of a row (DPP row rotations: pure VALU)
__device__ __forceinline__ float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false));  // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false));  // row_ror:1
  return v;
}

// real spherical harmonics up to degree 3 of (d + 1) / 2 (nerfstudio SHEncoding(levels=4) on get_normalized_directions, umhs_field.py:160-162)
__device__ __forceinline__ void sh16(float x, float y, float z, float (&c)[16]) {
#pragma clang fp contract(off)
  const float xx = x * x, yy = y * y, zz = z * z;
  c[0] = 0.28209479177387814f;
  c[1] = 0.4886025119029199f * y, c[2] = 0.4886025119029199f * z, c[3] = 0.4886025119029199f * x;
  c[4] = 1.0925484305920792f * x * y, c[5] = 1.0925484305920792f * y * z, c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
  c[7] = 1.0925484305920792f * x * z, c[8] = 0.5462742152960396f * (xx - yy);
  c[9] = 0.5900435899266435f * y * (3.0f * xx - yy), c[10] = 2.890611442640554f * x * y * z;
  c[11] = 0.4570457994644658f * y * (5.0f * zz - 1.0f), c[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
  c[13] = 0.4570457994644658f * x * (5.0f * zz - 1.0f), c[14] = 1.445305721320277f * z * (xx - yy);
  c[15] = 0.5900435899266435f * x * (xx - 3.0f * yy);
}

// the wave's input tile x[16 samples][KIN] into its LDS rows.  HEAD: [SH16 | emb15 | 0]; base: the 32 hash features of the sample.
template <bool HEAD>
__device__ __forceinline__ void load_inputs(float* x, const float* in_a, const float* in_b, int64_t tile, int64_t n, int lane) {
  const int q = lane >> 4, m = lane & 15;
  const int64_t i = tile * 16 + m, ii = i < n ? i : n - 1;
  if (HEAD) {
    if (q == 0) {
      float c[16];
      sh16((in_a[3 * ii] + 1.0f) / 2.0f, (in_a[3 * ii + 1] + 1.0f) / 2.0f, (in_a[3 * ii + 2] + 1.0f) / 2.0f, c);
#pragma unroll
      for (int k = 0; k < 16; ++k) x[m * LDW1 + k] = c[k];
    } else if (q == 1) {
#pragma unroll
      for (int k = 0; k < 15; ++k) x[m * LDW1 + 16 + k] = in_b[15 * ii + k];
      x[m * LDW1 + 31] = 0.0f;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[m * LDW1 + 8 * q + k] = in_a[32 * ii + 8 * q + k];
  }
  __builtin_amdgcn_wave_barrier();
}

constexpr int WAVES = 4;
// LDS floats: weight images + per-wave tiles.  Head: W0 64x33, W1 64x65, W2 16x65, biases 64 + 64 + 16; base: W0 64x33, W1 16x65, 64 + 16
constexpr int HEAD_W = H * LDW1 + H * LDWH + 16 * LDWH + H + H + 16;
constexpr int BASE_W = H * LDW1 + 16 * LDWH + H + 16;
constexpr int TILE_X = 16 * LDW1;          // input tile
constexpr int TILE_ROWS = H * 17;          // one staged [64][17] tile

template <bool HEAD>
__device__ __forceinline__ void load_images(float* lds, const RgbMlp& p, float*& w0, float*& b0, float*& w1, float*& b1, float*& w2, float*& b2) {
  w0 = lds, b0 = w0 + H * LDW1;
  w1 = b0 + H, b1 = w1 + (HEAD ? H : 16) * LDWH;
  load_w(w0, LDW1, p.w0, H, p.in0, H, KIN);
  load_b(b0, p.b0, H, H);
  if (HEAD) {
    w2 = b1 + H, b2 = w2 + 16 * LDWH;
    load_w(w1, LDWH, p.w1, H, H, H, H);
    load_b(b1, p.b1, H, H);
    load_w(w2, LDWH, p.w2, 3, H, 16, H);
    load_b(b2, p.b2, 3, 16);
  } else {
    w2 = b2 = nullptr;
    load_w(w1, LDWH, p.w1, 16, H, 16, H);
    load_b(b1, p.b1, 16, 16);
  }
  __syncthreads();
}

// ---- forward -------------------------------------------------------------------------------------------------------------------------
// base: in_a = enc [N,32], sel [N] (or NULL) -> out_a = density [N], out_b = emb [N,15] (or NULL), out_c = sigma_raw [N] (or NULL)
// head: in_a = directions [N,3], in_b = emb [N,15] -> out_a = rgb [N,3]
template <bool HEAD>
__global__ __launch_bounds__(64 * WAVES) void rgb_mlp_fwd_kernel(RgbMlp p, const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                                   const float* __restrict__ sel, int64_t n, float* __restrict__ out_a,
                                                                   float* __restrict__ out_b, float* __restrict__ out_c) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *w0, *b0, *w1, *b1, *w2, *b2;
  load_images<HEAD>(lds, p, w0, b0, w1, b1, w2, b2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q = lane >> 4, m = lane & 15;
  float* x = lds + (HEAD ? HEAD_W : BASE_W) + wv * TILE_X;
  const int64_t tiles = (n + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wv; tile < tiles; tile += (int64_t)gridDim.x * WAVES) {
    load_inputs<HEAD>(x, in_a, in_b, tile, n, lane);
    v4f h1[4];
    gemm_first<4>(h1, w0, b0, x, lane);
    relu(h1);
    const int64_t i = tile * 16 + m;
    if constexpr (HEAD) {
      v4f h2[4], o[1];
      gemm_next<4, 4>(h2, w1, b1, h1, lane);
      relu(h2);
      gemm_next<1, 4>(o, w2, b2, h2, lane);
      if (q == 0 && i < n) {
#pragma unroll
        for (int r = 0; r < 3; ++r) out_a[3 * i + r] = 1.0f / (1.0f + expf(-o[0][r]));
      }
    } else {
      v4f o[1];
      gemm_next<1, 4>(o, w1, b1, h1, lane);
      if (i < n) {
        if (q == 0) {
          out_a[i] = expf(o[0][0]) * (sel ? sel[i] : 1.0f);  // trunc_exp forward = exp (umhs_field.py:327); average_init_density 1
          if (out_c) out_c[i] = o[0][0];
        }
        if (out_b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 4 * q + r;
            if (f >= 1) out_b[15 * i + f - 1] = o[0][r];
          }
        }
      }
    }
  }
}

// ---- backward ------------------------------------------------------------------------------------------------------------------------
// base: in_a = enc, sel; g_a = d_density [N] (or NULL), g_b = d_emb [N,15] (or NULL) -> d_in = d_enc [N,32]
// head: in_a = directions, in_b = emb; g_a = d_rgb [N,3] -> d_in = d_emb [N,15]
// slab (per wave): the parameter gradients in the parameters' own order, [w0 | b0 | w1 | b1 | (w2 | b2)]
template <bool HEAD>
__global__ __launch_bounds__(64 * WAVES, 1) void rgb_mlp_bwd_kernel(RgbMlp p, const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                                      const float* __restrict__ sel, const float* __restrict__ g_a,
                                                                      const float* __restrict__ g_b, int64_t n, float* __restrict__ d_in,
                                                                      float* __restrict__ slabs, int slab_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *w0, *b0, *w1, *b1, *w2, *b2;
  load_images<HEAD>(lds, p, w0, b0, w1, b1, w2, b2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q = lane >> 4, m = lane & 15;
  float* x = lds + (HEAD ? HEAD_W : BASE_W) + WAVES * 0 + wv * (TILE_X + 2 * TILE_ROWS);
  float* ta = x + TILE_X;         // staged [feature][sample] tile A (dZ)
  float* tb = ta + TILE_ROWS;     // staged tile B (X / activations)
  constexpr int O1 = HEAD ? 4 : 1;  // output tiles of the second layer
  v4f dw0[4][2], dw1[O1][4], dw2[HEAD ? 1 : 1][HEAD ? 4 : 1];
  float db0[4][4], db1[O1][4], db2[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) dw0[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) db0[a][r] = 0.0f;
  }
#pragma unroll
  for (int a = 0; a < O1; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) dw1[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) db1[a][r] = 0.0f;
  }
#pragma unroll
  for (int b = 0; b < (HEAD ? 4 : 1); ++b) dw2[0][b] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) db2[r] = 0.0f;

  const int64_t tiles = (n + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wv; tile < tiles; tile += (int64_t)gridDim.x * WAVES) {
    load_inputs<HEAD>(x, in_a, in_b, tile, n, lane);
    const int64_t i = tile * 16 + m;
    const bool live = i < n;
    v4f h1[4];
    gemm_first<4>(h1, w0, b0, x, lane);
    relu(h1);
    v4f dz1[4];  // gradient w.r.t. the first layer's pre-activation
    if constexpr (HEAD) {
      v4f h2[4], o[1], dzo[1], dh2[4], dh1[4];
      gemm_next<4, 4>(h2, w1, b1, h1, lane);
      relu(h2);
      gemm_next<1, 4>(o, w2, b2, h2, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = 1.0f / (1.0f + expf(-o[0][r]));
        const float g = (live && q == 0 && r < 3) ? g_a[3 * i + r] : 0.0f;
        dzo[0][r] = g * y * (1.0f - y);
        db2[r] += dzo[0][r];
      }
      stage<1>(ta, dzo, lane), stage<4>(tb, h2, lane);
      dw_acc<1, 4>(dw2, ta, tb, lane);
      gemm_tr<4, 1>(dh2, w2, LDWH, dzo, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dh2[t][r] = h2[t][r] > 0.0f ? dh2[t][r] : 0.0f, db1[t][r] += dh2[t][r];
      }
      stage<4>(ta, dh2, lane), stage<4>(tb, h1, lane);
      dw_acc<4, 4>(dw1, ta, tb, lane);
      gemm_tr<4, 4>(dh1, w1, LDWH, dh2, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dz1[t][r] = h1[t][r] > 0.0f ? dh1[t][r] : 0.0f;
      }
    } else {
      v4f o[1], dzo[1], dh1[4];
      gemm_next<1, 4>(o, w1, b1, h1, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * q + r;
        float g = 0.0f;
        if (live) {
          if (f == 0)  // trunc_exp backward: g * exp(clamp(x, -15, 15)) (umhs_field.py:17), through the selector product
            g = g_a ? g_a[i] * (sel ? sel[i] : 1.0f) * expf(fminf(fmaxf(o[0][0], -15.0f), 15.0f)) : 0.0f;
          else
            g = g_b ? g_b[15 * i + f - 1] : 0.0f;
        }
        dzo[0][r] = g;
        db1[0][r] += g;
      }
      stage<1>(ta, dzo, lane), stage<4>(tb, h1, lane);
      dw_acc<1, 4>(dw1, ta, tb, lane);
      gemm_tr<4, 1>(dh1, w1, LDWH, dzo, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dz1[t][r] = h1[t][r] > 0.0f ? dh1[t][r] : 0.0f;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) db0[t][r] += dz1[t][r];
    }
    // dW0 += dZ1 X^T: the input tile as [feature][sample] rows
    stage<4>(ta, dz1, lane);
#pragma unroll
    for (int k = 0; k < 8; ++k) tb[(8 * q + k) * 17 + m] = x[m * LDW1 + 8 * q + k];
    __builtin_amdgcn_wave_barrier();
    dw_acc<4, 2>(dw0, ta, tb, lane);
    v4f dx[2];
    gemm_tr<2, 4>(dx, w0, LDW1, dz1, lane);
    if (live) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * q + r;
          if (HEAD) {
            if (f >= 16 && f < 31) d_in[15 * i + f - 16] = dx[t][r];  // the embedding's share (directions carry no gradient)
          } else {
            d_in[32 * i + f] = dx[t][r];
          }
        }
      }
    }
  }
  // one slab per wave, in the parameters' own order
  float* sl = slabs + ((size_t)blockIdx.x * WAVES + wv) * slab_floats;
  const int in0 = p.in0;
  int off = 0;
#pragma unroll
  for (int to = 0; to < 4; ++to) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * to + 4 * q + r, c = 16 * ti + m;
        if (c < in0) sl[off + o * in0 + c] = dw0[to][ti][r];
      }
    }
  }
  off += H * in0;
#pragma unroll
  for (int to = 0; to < 4; ++to) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sres = row_sum16(db0[to][r]);
      if (m == 0) sl[off + 16 * to + 4 * q + r] = sres;
    }
  }
  off += H;
  const int rows1 = HEAD ? H : 16;
#pragma unroll
  for (int to = 0; to < O1; ++to) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[off + (16 * to + 4 * q + r) * H + 16 * ti + m] = dw1[to][ti][r];
    }
  }
  off += rows1 * H;
#pragma unroll
  for (int to = 0; to < O1; ++to) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sres = row_sum16(db1[to][r]);
      if (m == 0) sl[off + 16 * to + 4 * q + r] = sres;
    }
  }
  off += rows1;
  if constexpr (HEAD) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 4 * q + r;
        if (o < 3) sl[off + o * H + 16 * ti + m] = dw2[0][ti][r];
      }
    }
    off += 3 * H;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sres = row_sum16(db2[r]);
      if (m == 0 && 4 * q + r < 3) sl[off + 4 * q + r] = sres;
    }
  }
}

// grads[j] (+)= sum over the slabs, in slab order (fixed: the result does not depend on the launch's timing)
__global__ __launch_bounds__(256) void rgb_mlp_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int slab_floats, float* __restrict__ g0,
                                                              int n0, float* __restrict__ g1, int n1, float* __restrict__ g2, int n2,
                                                              float* __restrict__ g3, int n3, float* __restrict__ g4, int n4, float* __restrict__ g5,
                                                              int n5, int accumulate) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= slab_floats) return;
  float s = 0.0f;
  for (int k = 0; k < n_slabs; ++k) s += slabs[(size_t)k * slab_floats + j];
  float* dst;
  int o = j;
  if (o < n0) dst = g0 + o;
  else if ((o -= n0) < n1) dst = g1 + o;
  else if ((o -= n1) < n2) dst = g2 + o;
  else if ((o -= n2) < n3) dst = g3 + o;
  else if ((o -= n3) < n4) dst = g4 + o;
  else { o -= n4; dst = g5 + o; }
  *dst = accumulate ? *dst + s : s;
}

inline int param_floats(bool head, int in0) { return head ? H * in0 + H + H * H + H + 3 * H + 3 : H * in0 + H + 16 * H + 16; }
inline size_t lds_bytes(bool head, bool bwd) {
  return (size_t)((head ? HEAD_W : BASE_W) + WAVES * (TILE_X + (bwd ? 2 * TILE_ROWS : 0))) * sizeof(float);
}
inline int grid_for(int64_t n) {
  const int64_t tiles = (n + 15) / 16, wgs = (tiles + WAVES - 1) / WAVES;
  return (int)(wgs < 256 ? wgs : 256);
}

template <typename K>
inline int raise_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : 1;
}

}  // namespace

extern "C" size_t umhs_rgb_mlp_bwd_workspace_bytes(int head, int64_t n) {
  if (n <= 0) return 0;
  return (size_t)grid_for(n) * WAVES * param_floats(head != 0, head ? 31 : 32) * sizeof(float);
}

extern "C" int umhs_rgb_base_fwd(const float* enc, const float* selector, const float* w0, const float* b0, const float* w1,
                                 const float* b1, int64_t n, float* density, float* emb, float* sigma_raw, umhs_stream_t stream) {
  if (n < 0 || !w0 || !b0 || !w1 || !b1) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  if (!enc || !density) return UMHS_ERR_ARG;
  RgbMlp p{w0, b0, w1, b1, nullptr, nullptr, 32};
  const size_t lds = lds_bytes(false, false);
  if (raise_lds(rgb_mlp_fwd_kernel<false>, lds)) return UMHS_ERR_LAUNCH;
  hipLaunchKernelGGL(rgb_mlp_fwd_kernel<false>, dim3(grid_for(n)), dim3(64 * WAVES), lds, umhs_s(stream), p, enc, nullptr, selector, n,
                     density, emb, sigma_raw);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_rgb_head_fwd(const float* directions, const float* emb, const float* w0, const float* b0, const float* w1,
                                 const float* b1, const float* w2, const float* b2, int64_t n, float* rgb, umhs_stream_t stream) {
  if (n < 0 || !w0 || !b0 || !w1 || !b1 || !w2 || !b2) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  if (!directions || !emb || !rgb) return UMHS_ERR_ARG;
  RgbMlp p{w0, b0, w1, b1, w2, b2, 31};
  const size_t lds = lds_bytes(true, false);
  if (raise_lds(rgb_mlp_fwd_kernel<true>, lds)) return UMHS_ERR_LAUNCH;
  hipLaunchKernelGGL(rgb_mlp_fwd_kernel<true>, dim3(grid_for(n)), dim3(64 * WAVES), lds, umhs_s(stream), p, directions, emb, nullptr, n, rgb,
                     nullptr, nullptr);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_rgb_base_bwd(const float* enc, const float* selector, const float* w0, const float* b0, const float* w1,
                                 const float* b1, const float* d_density, const float* d_emb, int64_t n, float* d_enc, float* d_w0,
                                 float* d_b0, float* d_w1, float* d_b1, int accumulate, void* workspace, size_t workspace_bytes,
                                 umhs_stream_t stream) {
  if (n < 0 || !w0 || !b0 || !w1 || !b1 || !d_w0 || !d_b0 || !d_w1 || !d_b1) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  if (!enc || !d_enc || (!d_density && !d_emb)) return UMHS_ERR_ARG;
  const size_t need = umhs_rgb_mlp_bwd_workspace_bytes(0, n);
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return UMHS_ERR_WORKSPACE;
  RgbMlp p{w0, b0, w1, b1, nullptr, nullptr, 32};
  const size_t lds = lds_bytes(false, true);
  if (raise_lds(rgb_mlp_bwd_kernel<false>, lds)) return UMHS_ERR_LAUNCH;
  const int grid = grid_for(n), pf = param_floats(false, 32);
  float* slabs = static_cast<float*>(workspace);
  hipLaunchKernelGGL(rgb_mlp_bwd_kernel<false>, dim3(grid), dim3(64 * WAVES), lds, umhs_s(stream), p, enc, nullptr, selector, d_density, d_emb, n,
                     d_enc, slabs, pf);
  hipLaunchKernelGGL(rgb_mlp_reduce_kernel, dim3((pf + 255) / 256), dim3(256), 0, umhs_s(stream), slabs, grid * WAVES, pf, d_w0, H * 32, d_b0, H,
                     d_w1, 16 * H, d_b1, 16, (float*)nullptr, 0, (float*)nullptr, 0, accumulate);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

extern "C" int umhs_rgb_head_bwd(const float* directions, const float* emb, const float* w0, const float* b0, const float* w1,
                                 const float* b1, const float* w2, const float* b2, const float* d_rgb, int64_t n, float* d_emb,
                                 float* d_w0, float* d_b0, float* d_w1, float* d_b1, float* d_w2, float* d_b2, int accumulate,
                                 void* workspace, size_t workspace_bytes, umhs_stream_t stream) {
  if (n < 0 || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !d_w0 || !d_b0 || !d_w1 || !d_b1 || !d_w2 || !d_b2) return UMHS_ERR_ARG;
  if (n == 0) return UMHS_OK;
  if (!directions || !emb || !d_rgb || !d_emb) return UMHS_ERR_ARG;
  const size_t need = umhs_rgb_mlp_bwd_workspace_bytes(1, n);
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return UMHS_ERR_WORKSPACE;
  RgbMlp p{w0, b0, w1, b1, w2, b2, 31};
  const size_t lds = lds_bytes(true, true);
  if (raise_lds(rgb_mlp_bwd_kernel<true>, lds)) return UMHS_ERR_LAUNCH;
  const int grid = grid_for(n), pf = param_floats(true, 31);
  float* slabs = static_cast<float*>(workspace);
  hipLaunchKernelGGL(rgb_mlp_bwd_kernel<true>, dim3(grid), dim3(64 * WAVES), lds, umhs_s(stream), p, directions, emb, nullptr, d_rgb, nullptr, n,
                     d_emb, slabs, pf);
  hipLaunchKernelGGL(rgb_mlp_reduce_kernel, dim3((pf + 255) / 256), dim3(256), 0, umhs_s(stream), slabs, grid * WAVES, pf, d_w0, H * 31, d_b0, H,
                     d_w1, H * H, d_b1, H, d_w2, 3 * H, d_b2, 3, accumulate);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}
