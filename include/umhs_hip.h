/*
 * umhs_hip.h -- C ABI of libumhs_hip.so: the MI355X (gfx950) implementation of the UMHS
 * volumetric-rendering hot path (hash-grid encode, density/spectral MLPs + endmember mixing,
 * per-ray alpha compositing over B bands, spectrum->sRGB, fused Adam).
 *
 * The reference (Factral/unsupervised-hyperspectral-nerf) has no C ABI: its boundary is the
 * nerfstudio plugin API in Python, and the native code it reaches lives in tiny-cuda-nn / nerfacc.
 * This header sits where those libraries' Python bindings used to sit; each entry point cites the
 * reference call site (file:line relative to the reference repo) it replaces.  The Python mirror of
 * the plugin surface (UMHSField, SpectralRenderer, ColourSystem, UMHSModel ...) binds these symbols
 * with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns UMHS_OK (0) or a negative error code (umhs_strerror), never throws;
 *   - never allocates, never synchronises the stream; workspaces are caller-provided and sized by
 *     the matching *_workspace_bytes query;
 *   - all pointers are DEVICE pointers to contiguous row-major fp32 unless stated (int64 for
 *     ray_indices / packed_info, exactly the dtypes nerfacc hands out);
 *   - `stream` is a hipStream_t (NULL = default stream); calls are re-entrant on distinct streams;
 *   - no torch types anywhere in the signatures.
 */
#ifndef UMHS_HIP_H
#define UMHS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMHS_ABI_VERSION 11

enum {
  UMHS_OK = 0,
  UMHS_ERR_ARG = -1,         /* NULL / negative / inconsistent argument            */
  UMHS_ERR_UNSUPPORTED = -2, /* shape outside what the gfx950 kernels are built for */
  UMHS_ERR_WORKSPACE = -3,   /* workspace missing or too small                     */
  UMHS_ERR_LAUNCH = -4       /* hipGetLastError() != hipSuccess after a launch      */
};

typedef void* umhs_stream_t; /* hipStream_t */

const char* umhs_strerror(int code);
int umhs_abi_version(void);

/* ------------------------------------------------------------------------------------------ */
/* R1 (prefix): sample positions.  Replaces Frustums.get_positions + SceneContraction(L-inf) +  */
/* (p+2)/4 + in-box selector of UMHSField.get_density, umhs_field.py:302-310.                   */
/* Either (origins,directions,starts,ends) [N,3],[N,3],[N],[N] or world_pos_in [N,3] is given   */
/* (the latter is the density_fn(positions) path used by the occupancy grid,                    */
/* umhs_model.py:208,553).  contraction != 0 -> L-inf contraction then (p+2)/4;                 */
/* contraction == 0 -> (p - aabb_min)/(aabb_max-aabb_min) with aabb = 6 HOST floats.            */
/* Outputs: world_pos_out [N,3] (optional), pos01_out [N,3] (already multiplied by selector),   */
/* selector_out [N] (1.0f / 0.0f).                                                              */
/* ------------------------------------------------------------------------------------------ */
int umhs_positions_fwd(const float* origins, const float* directions, const float* starts, const float* ends,
                       const float* world_pos_in, int64_t n, int contraction, const float* aabb_host6,
                       float* world_pos_out, float* pos01_out, float* selector_out, umhs_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* R2: multiresolution hash-grid encode.  Replaces nerfstudio HashEncoding.forward (torch path) */
/* / tcnn HashGrid reached through self.mlp_base(positions_flat), umhs_field.py:320.            */
/* pos01 [N,3]; table [L*T, 2]; scalings [L] (device, = floor(16*g^l) as float32);              */
/* enc element (n,l,f) is written at enc[n*stride_n + l*stride_l + f] (f in {0,1}):             */
/*   stride_n=2L, stride_l=2  -> the reference's [N, L*F] layout;                               */
/*   stride_n=2,  stride_l=2N -> level-major [L][N][2] (what the fused field kernels prefer).   */
/* hashgrid_bwd: overwrite == 0 ACCUMULATES (d_table[idx] += w_corner * d_enc, caller zeroes   */
/* it); overwrite != 0 writes the whole [L*T,2] gradient (no memset / read-modify-write needed).*/
/* With a workspace (umhs_hashgrid_bwd_workspace_bytes; 0 = not available for this shape) the   */
/* scatter is atomics-free (radix partition into LDS-sized slot buckets); with workspace ==     */
/* NULL it falls back to global float atomics (no workspace, ~20x slower at N = 262k).          */
/* ------------------------------------------------------------------------------------------ */
int umhs_hashgrid_fwd(const float* pos01, const float* table, const float* scalings, int64_t n, int n_levels,
                      int log2_table_size, float* enc, int64_t stride_n, int64_t stride_l, umhs_stream_t stream);
/* Level-major feature compaction enc_out[l][i] = enc_in[l][index[i]] ([L][M][2] -> [L][N][2]): lets a caller that has already  */
/* encoded a superset of the samples (the sampler's density query) reuse those features instead of encoding the survivors again.  */
int umhs_enc_gather(const float* enc_in, const int64_t* index, int64_t m, int64_t n, int n_levels, float* enc_out,
                    umhs_stream_t stream);
size_t umhs_hashgrid_bwd_workspace_bytes(int64_t n, int n_levels, int log2_table_size);
/* Levels [level_begin, level_begin + n_levels) are processed (d_enc / d_table / scalings are indexed with the     */
/* absolute level): callers may run the backward per level group, e.g. to all-reduce finished slabs early.          */
int umhs_hashgrid_bwd(const float* pos01, const float* d_enc, int64_t stride_n, int64_t stride_l,
                      const float* scalings, int64_t n, int level_begin, int n_levels, int log2_table_size,
                      float* d_table, int overwrite, void* workspace, size_t workspace_bytes, umhs_stream_t stream);
/* The partitioned backward in two halves.  prepare: bucket histogram + scan, from the positions alone (may run on a side   */
/* stream while the forward pass is in flight).  apply: scatter + per-bucket reduction of levels [level_begin, +n_levels),  */

/* umhs_hashgrid_fwd for the workspace's level range [0, n_levels) AND the histogram pass of umhs_hashgrid_bwd_prepare for the same   */
/* positions, in one launch (the gather has hashed every (sample, level) anyway and leaves the vector ALU idle);                     */
/* umhs_hashgrid_bwd_prepare_counted then runs only the two small scans.  Together they equal umhs_hashgrid_fwd +                    */
/* umhs_hashgrid_bwd_prepare(level_begin 0) bit for bit.  workspace: umhs_hashgrid_bwd_workspace_bytes(n, n_levels, log2_T).          */
int umhs_hashgrid_fwd_count(const float* pos01, const float* table, const float* scalings, int64_t n, int n_levels, int log2_T,
                            float* enc, int64_t stride_n, int64_t stride_l, void* workspace, size_t workspace_bytes,
                            umhs_stream_t stream);
int umhs_hashgrid_bwd_prepare_counted(const float* pos01, const float* scalings, int64_t n, int n_levels, int log2_T, void* workspace,
                                      size_t workspace_bytes, umhs_stream_t stream);

/* a sub-range of the prepared [ws_level_begin, +ws_n_levels) in the same workspace; each level once per prepare.          */
int umhs_hashgrid_bwd_prepare(const float* pos01, const float* scalings, int64_t n, int level_begin, int n_levels, int log2_T,
                              void* workspace, size_t workspace_bytes, umhs_stream_t stream);
int umhs_hashgrid_bwd_apply(const float* pos01, const float* d_enc, int64_t stride_n, int64_t stride_l, const float* scalings,
                            int64_t n, int level_begin, int n_levels, int ws_level_begin, int ws_n_levels, int log2_T,
                            float* d_table, int overwrite, void* workspace, size_t workspace_bytes, umhs_stream_t stream);
/* umhs_hashgrid_bwd_apply in overwrite mode + the Adam step (umhs_adam_step arithmetic, grad_scale 1) of the table entries of    */
/* levels >= adam_level_begin, executed in the epilogue of the bucket reduce where their gradient is final.  For a single-GPU      */
/* trainer whose optimizer.step() follows the backward (UMHSAdam skips the range it is told was done): the HBM-bound update hides   */
/* inside the LDS-bound reduce and the gradient is not read back.  d_table still receives the gradient.  n > 0.                     */
int umhs_hashgrid_bwd_apply_adam(const float* pos01, const float* d_enc, int64_t stride_n, int64_t stride_l,
                                 const float* scalings, int64_t n, int level_begin, int n_levels, int ws_level_begin,
                                 int ws_n_levels, int log2_T, float* d_table, void* workspace, size_t workspace_bytes,
                                 float* table_params, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                                 float eps, int64_t step, int adam_level_begin, umhs_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* R3-R9, R18: fused per-sample field.  Replaces mlp_base's MLP, NeRFEncoding, SHEncoding,      */
/* mlp_head, feature_mlp, mlp_directional, softmax/sigmoid and the [N,B,C] endmember mixing of  */
/* UMHSField.get_density / get_outputs, umhs_field.py:160-261,320-328 (tcnn.Network /           */
/* nerfstudio MLP call sites :211,:219,:250).                                                   */
/* Weights are the reference's nn.Linear tensors as they sit in the state dict: W [out,in]      */
/* row-major, b [out].  hidden = 64, geo_feat_dim = 15, 16 hash levels x 2 features are fixed   */
/* (NerfactoField defaults); n_classes <= 15, n_bands <= 256.                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct umhs_field_cfg {
  int32_t n_bands;       /* B: wavelengths                                   */
  int32_t n_classes;     /* C: endmembers ("num_classes")                    */
  int32_t pred_specular; /* 1: feature_mlp has C+1 outputs, mlp_directional used */
  int32_t density_only;  /* 1: hash features -> sigma, emb only (density_fn)  */
  float temperature;     /* softmax(logits / temperature), umhs_field.py:226  */
} umhs_field_cfg;

typedef struct umhs_field_params {
  const float *base_w0, *base_b0, *base_w1, *base_b1;                       /* 32->64->16            */
  const float *head_w0, *head_b0, *head_w1, *head_b1, *head_w2, *head_b2;   /* 27->64->64->C         */
  const float *feat_w0, *feat_b0, *feat_w1, *feat_b1, *feat_w2, *feat_b2;   /* 27->64->64->C(+1)     */
  const float *dir_w0, *dir_b0, *dir_w1, *dir_b1;                           /* 28->16->B (specular)  */
  const float* endmembers;                                                  /* [C,B]                 */
} umhs_field_params;

typedef struct umhs_field_grads {
  float *base_w0, *base_b0, *base_w1, *base_b1;
  float *head_w0, *head_b0, *head_w1, *head_b1, *head_w2, *head_b2;
  float *feat_w0, *feat_b0, *feat_w1, *feat_b1, *feat_w2, *feat_b2;
  float *dir_w0, *dir_b0, *dir_w1, *dir_b1;
  float* endmembers;
} umhs_field_grads;

/* Forward.  enc is addressed with (stride_n, stride_l) as above.  world_pos [N,3] (raw, for the */
/* NeRF positional encoding, umhs_field.py:183-184), directions [N,3] (raw; (d+1)/2 applied     */
/* inside, :160), selector [N].  Outputs (any may be NULL except sigma):                        */
/*   sigma [N] = trunc_exp(raw)*selector (:327-328); sigma_raw [N]; emb [N,15];                 */
/*   spectral [N,B] (= spec + s1*specular when pred_specular, else spec), spectral2 [N,B] (spec),*/
/*   specular [N,B] (s1*specular), abundances [N,C].                                            */
/* workspace (optional, umhs_field_fwd_workspace_bytes): room for the packed weight image built once per call;   */
/* with workspace == NULL every workgroup gathers the image itself (~60 us slower per launch).                   */
size_t umhs_field_fwd_workspace_bytes(const umhs_field_cfg* cfg);
int umhs_field_fwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc, int64_t stride_n,
                   int64_t stride_l, const float* world_pos, const float* directions, const float* selector,
                   int64_t n, float* sigma, float* sigma_raw, float* emb, float* spectral, float* spectral2,
                   float* specular, float* abundances, float* feat_logits, void* workspace, size_t workspace_bytes,
                   int pack_ready, umhs_stream_t stream);
/* density_fn (SURVEY 8a R1-R3 for the sampler / occupancy-grid callers, umhs_model.py:208,553) = umhs_hashgrid_fwd followed by   */
/* umhs_field_fwd with cfg->density_only.  (A one-launch form with the gather inside the MLP kernel existed up to ABI 7: it took  */
/* as long as the two launches -- both are bound by the gather's L2 request rate -- and spilled registers; removed.)              */
/* The training step's forward as two launches with the rendering weights known in between (umhs_model.py:239-327: field ->    */
/* renderers; here the per-ray band sums of the [N,B] outputs are formed inside the heads kernel, so spectral2 / specular --     */
/* which carry no loss, umhs_model.py:373-374 -- never exist per sample):                                                        */
/*   umhs_field_base_fwd : mlp_base only (sigma, sigma_raw, emb as umhs_field_fwd writes them) from the FULL configuration's      */
/*                         workspace (umhs_field_fwd_workspace_bytes / umhs_field_fwd_prepare: one set of pack images for both).  */
/*   umhs_composite_fwd with no value streams: weights, accumulation, depth.                                                     */
/*   umhs_field_heads_fwd: everything after mlp_base from emb [N,15]; comp_*[r][b] = sum_{n in ray r} weights[n] stream[n][b]    */
/*                         (SpectralRenderer, renderers.py:18-53) for spectral / spectral2 / specular ([R,B]; the last two NULL   */
/*                         without the specular head), abundances, feat_logits [N,16]; no [N,B] array is ever written.            */
/*                         ray_indices [N] non-decreasing; packed_info [R,2] as umhs_pack_info.  The sums are taken in a          */
/*                         fixed order (per 16-sample tile, then tile by tile): same bits every run.                             */
int umhs_field_base_fwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc, int64_t stride_n,
                        int64_t stride_l, const float* selector, int64_t n, float* sigma, float* sigma_raw, float* emb,
                        float* base16, void* workspace, size_t workspace_bytes, int pack_ready, umhs_stream_t stream);
size_t umhs_field_heads_fwd_scratch_bytes(const umhs_field_cfg* cfg, int64_t n, int64_t n_rays);
/* 1 when the two-launch forward can serve this configuration (else both entries return UMHS_ERR_UNSUPPORTED and the caller keeps   */
/* umhs_field_fwd + umhs_composite_fwd, which serve every configuration check_cfg admits).                                          */
int umhs_field_heads_fwd_supported(const umhs_field_cfg* cfg);
/* emb / base16: the base MLP's outputs either as the reference's [N,15] embedding (emb, emb_stride 15; any of emb / base16 may be    */
/* NULL in umhs_field_base_fwd) or as aligned rows base16 [N,16] with sigma_raw in slot 0 (emb_stride 16: one 64-byte row per sample  */
/* instead of 15 dword stores / loads).  The mixing term is linear in the mixing input m: the kernel sums w_n m_n (16 classes) per   */
/* ray and the finish pass multiplies by the endmembers once per RAY; only the specular term is formed per sample and band.          */
/* comp_spectral2 / comp_specular: required with the specular head, ignored without.  comp_abundances [R,C], abundances [N,C],      */
/* feat_logits [N,16]: optional.  scratch: umhs_field_heads_fwd_scratch_bytes, 16-byte aligned.                                      */
int umhs_field_heads_fwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* emb, int emb_stride,
                         const float* world_pos, const float* directions, int64_t n, const float* weights,
                         const int64_t* ray_indices, const int64_t* packed_info, int64_t n_rays, float* abundances,
                         float* feat_logits, float* comp_spectral, float* comp_spectral2, float* comp_specular,
                         float* comp_abundances, void* scratch, size_t scratch_bytes, void* workspace, size_t workspace_bytes,
                         int pack_ready, umhs_stream_t stream);
/* feat_logits ([N,16]): the feature_mlp logits, saved because umhs_field_bwd runs its heads as two kernels (head MLP +    */
/* directional + mixing / feature MLP + mlp_base), the second one starting from the logits.  Optional in the forward,       */
/* REQUIRED by umhs_field_bwd (NULL there: UMHS_ERR_ARG; there is no single-kernel backward any more, ABI 7).                */
/* builds the pack image ahead of time (parameters only): then pass pack_ready = 1 with the same workspace */
int umhs_field_fwd_prepare(const umhs_field_cfg* cfg, const umhs_field_params* params, void* workspace,
                           size_t workspace_bytes, umhs_stream_t stream);

/* Backward.  Recomputes the activations per tile; the only saved forward tensors are enc, sigma_raw [N],       */
/* emb [N,15] and feat_logits [N,16] (all outputs of umhs_field_fwd).  d_sigma [N] and d_spectral [N,B] are the gradients w.r.t. the   */
/* forward's sigma / spectral outputs; d_emb_ext [N,15] (optional) is an extra gradient on emb.  Writes d_enc     */
/* (same strides as enc) and the parameter gradients (OVERWRITTEN, not accumulated).                             */
/* workspace: umhs_field_bwd_workspace_bytes.                                                                    */
size_t umhs_field_bwd_workspace_bytes(const umhs_field_cfg* cfg, int64_t n);
int umhs_field_bwd(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc, int64_t stride_n,
                   int64_t stride_l, const float* world_pos, const float* directions, const float* selector,
                   const float* sigma_raw, const float* emb, const float* feat_logits, int64_t n, const float* d_sigma,
                   const float* d_spectral, const float* d_emb_ext, float* d_enc, const umhs_field_grads* grads,
                   void* workspace, size_t workspace_bytes, int packs_ready, umhs_stream_t stream);
/* umhs_field_bwd with the value half of the compositing backward (nerfacc accumulate_along_rays backward through               */
/* SpectralRenderer, umhs_renderer.py:28-30) folded in -- the training step after umhs_field_heads_fwd.  Takes d_comp_spectral       */
/* [R,B] (gradient of the per-ray band sums) + d_accumulation [R] and the renderer's sigma / intervals / packed_info /              */
/* ray_indices / weights instead of d_spectral [N,B]; returns d_sigma [N] as well.  Per sample d_spectral = scale_n weights[n]      */
/* d_comp[ray(n)] is formed on the fly and spectral is recomputed for dw_n: neither exists as an [N,B] array.  feat_logits          */
/* required.  umhs_field_bwd_composited_supported: 1 when this configuration can take the path (else UMHS_ERR_UNSUPPORTED).         */
int umhs_field_bwd_composited_supported(const umhs_field_cfg* cfg);
size_t umhs_field_bwd_composited_scratch_bytes(const umhs_field_cfg* cfg, int64_t n, int64_t n_rays);
int umhs_field_bwd_composited(const umhs_field_cfg* cfg, const umhs_field_params* params, const float* enc, int64_t stride_n,
                              int64_t stride_l, const float* world_pos, const float* directions, const float* selector,
                              const float* sigma_raw, const float* emb, int emb_stride, const float* feat_logits, int64_t n,
                              const float* sigma,
                              const float* t_starts, const float* t_ends, const int64_t* packed_info, int64_t n_rays,
                              const int64_t* ray_indices, const float* weights, const float* d_comp_spectral,
                              const float* d_accumulation, int grad_scaling, float* d_sigma, float* d_enc,
                              const umhs_field_grads* grads, void* scratch, size_t scratch_bytes, void* workspace,
                              size_t workspace_bytes, int packs_ready, umhs_stream_t stream);
/* builds the transposed packs + forward image ahead of time (parameters only): then pass packs_ready = 1, same workspace */
int umhs_field_bwd_prepare(const umhs_field_cfg* cfg, const umhs_field_params* params, void* workspace,
                           size_t workspace_bytes, umhs_stream_t stream);

/* ---- method="rgb" (the reference's default method, umhs_field.py:280-294 = nerfstudio NerfactoField; BASELINE configs[0]) ---------- */
/* The two MLPs of the rgb field as gfx950 kernels (exact fp32 MFMA), replacing the torch.nn.functional.linear calls of rounds 1-3.     */
/*   base: hash features enc [N,32] (sample-major) -> 64 ReLU -> 16: density = trunc_exp(out0) * selector (selector NULL: 1),           */
/*         emb = out1..15 [N,15] (NULL: not written), sigma_raw = out0 [N] (NULL: not written)        -- mlp_base, umhs_field.py:51,320-327 */
/*   head: [SHEncoding(levels=4)((directions + 1) / 2) | emb15] -> 64 ReLU -> 64 ReLU -> 3, Sigmoid -> rgb [N,3]   -- NerfactoField.mlp_head */
/* Weights in torch.nn.Linear layout ([out][in] row-major, bias [out]).  The backward entries recompute the forward, write the input     */
/* gradient (d_enc [N,32] / d_emb [N,15]) and the parameter gradients (accumulate != 0: += ), bitwise reproducibly; workspace:           */
/* umhs_rgb_mlp_bwd_workspace_bytes(head, n), 16-byte aligned.  n == 0: no-op.                                                          */
size_t umhs_rgb_mlp_bwd_workspace_bytes(int head, int64_t n);
int umhs_rgb_base_fwd(const float* enc, const float* selector, const float* w0, const float* b0, const float* w1, const float* b1,
                      int64_t n, float* density, float* emb, float* sigma_raw, umhs_stream_t stream);
int umhs_rgb_head_fwd(const float* directions, const float* emb, const float* w0, const float* b0, const float* w1, const float* b1,
                      const float* w2, const float* b2, int64_t n, float* rgb, umhs_stream_t stream);
int umhs_rgb_base_bwd(const float* enc, const float* selector, const float* w0, const float* b0, const float* w1, const float* b1,
                      const float* d_density, const float* d_emb, int64_t n, float* d_enc, float* d_w0, float* d_b0, float* d_w1,
                      float* d_b1, int accumulate, void* workspace, size_t workspace_bytes, umhs_stream_t stream);
int umhs_rgb_head_bwd(const float* directions, const float* emb, const float* w0, const float* b0, const float* w1, const float* b1,
                      const float* w2, const float* b2, const float* d_rgb, int64_t n, float* d_emb, float* d_w0, float* d_b0,
                      float* d_w1, float* d_b1, float* d_w2, float* d_b2, int accumulate, void* workspace, size_t workspace_bytes,
                      umhs_stream_t stream);


/* ------------------------------------------------------------------------------------------ */
/* R11: packed transmittance/weights.  Replaces nerfacc.pack_info + render_weight_from_density, */
/* umhs_model.py:245-252 (dense twin: get_weights_spectral, umhs_renderer.py:117-139).          */
/* R12/R13: per-ray accumulation.  Replaces nerfacc.accumulate_along_rays behind                */
/* SpectralRenderer.forward (umhs_renderer.py:28-30) and nerfstudio's Accumulation/Depth        */
/* renderers (umhs_model.py:254-258).  ray_indices must be sorted ascending (nerfacc invariant). */
/* ------------------------------------------------------------------------------------------ */
int umhs_pack_info(const int64_t* ray_indices, int64_t n, int64_t n_rays, int64_t* packed_info, umhs_stream_t stream);

#define UMHS_MAX_STREAMS 4
typedef struct umhs_value_streams {
  int32_t n_streams;                     /* 0..4 value tensors composited with the same weights */
  int32_t k[UMHS_MAX_STREAMS];           /* channels of each [N,k] tensor                       */
  const float* values[UMHS_MAX_STREAMS]; /* [N,k]                                               */
  float* out[UMHS_MAX_STREAMS];          /* [R,k]                                               */
} umhs_value_streams;

/* weights [N] (out), accumulation [R] (out, optional), depth [R] (out, optional: sum w*t_mid /  */
/* (acc+1e-10), NOT yet clipped to [min t_mid, max t_mid] -- that global clip is the caller's).  */
int umhs_composite_fwd(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                       int64_t n_rays, int64_t n, const umhs_value_streams* streams, float* weights,
                       float* accumulation, float* depth, umhs_stream_t stream);

typedef struct umhs_value_grads {
  int32_t n_streams;
  int32_t k[UMHS_MAX_STREAMS];
  const float* values[UMHS_MAX_STREAMS]; /* [N,k] forward inputs                    */
  const float* d_out[UMHS_MAX_STREAMS];  /* [R,k] gradient of the composited output */
  float* d_values[UMHS_MAX_STREAMS];     /* [N,k] (out)                             */
} umhs_value_grads;

/* d_accumulation [R] optional.  grad_scaling != 0 multiplies d_sigma and d_values by              */
/* clamp(((t0+t1)/2)^2, 0, 1) (scale_gradients_by_distance_squared, umhs_model.py:241-242).        */
int umhs_composite_bwd(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                       int64_t n_rays, int64_t n, const float* weights, const umhs_value_grads* grads,
                       const float* d_accumulation, int grad_scaling, float* d_sigma, umhs_stream_t stream);
/* The density half alone, for a caller that has already formed dots[n] = sum over streams and bands of d_out[ray(n)][k] *    */
/* value[n][k] itself (umhs_field_bwd_composited does): d_sigma only, no [N,k] array read or written.                           */
int umhs_composite_bwd_dots(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                            int64_t n_rays, int64_t n, const float* weights, const float* dots, const float* d_accumulation,
                            int grad_scaling, float* d_sigma, umhs_stream_t stream);

/* R12 stand-alone: accumulate with caller-provided weights [N] -- SpectralRenderer.forward(spectral, weights,  */
/* ray_indices, num_rays) called on its own (umhs_renderer.py:15-30; dino / abundance renders,                  */
/* umhs_model.py:299-317).  out[r,:] = sum_n w[n] v[n,:];  bwd: d_weights[n] = sum_k d_out[r,k] v[n,k] (rays     */
/* with no samples leave their d_weights untouched: caller zero-fills), d_values[n,:] = w[n] d_out[r,:].         */
int umhs_accumulate_fwd(const float* weights, const int64_t* packed_info, int64_t n_rays, int64_t n,
                        const umhs_value_streams* streams, umhs_stream_t stream);
int umhs_accumulate_bwd(const float* weights, const int64_t* packed_info, int64_t n_rays, int64_t n,
                        const umhs_value_grads* grads, float* d_weights, umhs_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* R14: spectrum -> sRGB.  Replaces ColourSystem.forward, utils/spec_to_rgb.py:103-127:         */
/* rgb = clamp(gamma(spec @ M), 0, 1), M [B,3].  bwd writes d_spec [R,B] (accumulate != 0: +=). */
/* ------------------------------------------------------------------------------------------ */
int umhs_spec2rgb_fwd(const float* spec, const float* M, int64_t n_rays, int n_bands, float* rgb, umhs_stream_t stream);
int umhs_spec2rgb_bwd(const float* spec, const float* M, const float* d_rgb, int64_t n_rays, int n_bands,
                      float* d_spec, int accumulate, umhs_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* R13/R15/R16 fused per-ray epilogue + losses (the ~85 tiny torch kernels of umhs_model.py:254-313,358-370).     */
/* umhs_tmid_minmax: min / max over all samples of (t0+t1)/2 -- DepthRenderer's clip bounds (global over the      */
/*   batch); minmax2 = 2 device words in an order-preserving encoding consumed by umhs_ray_epilogue_fwd only.     */
/* umhs_ray_epilogue_fwd: rgb = ColourSystem(spectral) (spec_to_rgb.py:112-127); depth_clipped; seg_probs =       */
/*   softmax(alpha * cos(spectral, endmembers)) (clusterprobe.py:17-38, alpha = 0.2 at umhs_model.py:307);        */
/*   seg_raw = argmax * [acc > 0.5]; seg_pred = class_colors[argmax] * [acc > 0.5] (:308-313).  Any output NULL   */
/*   = skipped.  n_classes <= 16.  Gradient flows through rgb only (umhs_spec2rgb_bwd).                           */
/* umhs_loss_fwd: losses2[0] = w_spectral * MSE(spectral, gt_spectral); losses2[1] = w_rgb * MSE(rgb +           */
/*   background*(1-acc), gt_rgb) (rgb == NULL: skipped).  umhs_loss_bwd: their gradients scaled by the upstream    */
/*   gradients grad_losses2 (device [2], NULL = 1).                                                               */
/* ------------------------------------------------------------------------------------------ */
int umhs_tmid_minmax(const float* t_starts, const float* t_ends, int64_t n, float* minmax2, umhs_stream_t stream);
int umhs_ray_epilogue_fwd(const float* spectral, const float* M, const float* endmembers, const float* accumulation,
                          const float* depth, const float* tmid_minmax2, const float* class_colors, int64_t n_rays,
                          int n_bands, int n_classes, float alpha, float* rgb, float* depth_clipped, float* seg_probs,
                          float* seg_raw, float* seg_pred, umhs_stream_t stream);
int umhs_loss_fwd(const float* spectral, const float* gt_spectral, const float* rgb, const float* accumulation,
                  const float* background, const float* gt_rgb, int64_t n_rays, int n_bands, float w_spectral, float w_rgb,
                  float* losses2, umhs_stream_t stream);
int umhs_loss_bwd(const float* spectral, const float* gt_spectral, const float* rgb, const float* accumulation,
                  const float* background, const float* gt_rgb, int64_t n_rays, int n_bands, float w_spectral, float w_rgb,
                  const float* grad_losses2, float* d_spectral, float* d_rgb, float* d_accumulation, umhs_stream_t stream);

/* umhs_ray_train_tail: the per-ray tail of a TRAINING step in one launch = umhs_ray_epilogue_fwd + umhs_loss_fwd +      */
/*   umhs_loss_bwd with unit upstream gradients + umhs_spec2rgb_bwd accumulated into d_spectral (umhs_model.py:254-313,   */
/*   358-370 and their autograd).  rgb_loss = 0: method "spectral" (no rgb term, d_accumulation unused).  scratch:        */
/*   umhs_ray_train_tail_scratch_bytes() bytes, ZERO before the first call (the kernel leaves it zeroed again).          */
size_t umhs_ray_train_tail_scratch_bytes(void);
int umhs_ray_train_tail(const float* spectral, const float* M, const float* endmembers, const float* accumulation,
                        const float* depth, const float* tmid_minmax2, const float* class_colors, const float* gt_spectral,
                        const float* gt_rgb, const float* background, int64_t n_rays, int n_bands, int n_classes, float alpha,
                        float w_spectral, float w_rgb, int rgb_loss, float* rgb, float* depth_clipped, float* seg_probs,
                        float* seg_raw, float* seg_pred, float* losses2, float* d_spectral, float* d_accumulation,
                        void* scratch, size_t scratch_bytes, umhs_stream_t stream);
/* ------------------------------------------------------------------------------------------ */
/* occupancy-grid ray marcher.  Replaces nerfacc.OccGridEstimator.sampling (traverse_grids +        */
/* render_visibility_from_density, CUDA only) behind nerfstudio's VolumetricSampler, umhs_model.py:201-209,229-237. */
/* binaries: uint8 [levels][res][res][res] (x-major); level l covers the roi enlarged 2^l about its centre;          */
/* roi_aabb_host6 = 6 HOST floats (min xyz, max xyz).  Samples have size dt = max(t*cone_angle, step_size) and are   */
/* emitted while their mid-point lies in an occupied voxel; a run restarts at the voxel entry after empty space.     */
/* nears / fars: optional per-ray planes [R] (stratified jitter, collider).                                          */
/* umhs_march_walk (optional, in front of the entry points below): the voxel sequence of every ray -- pure geometry, it never     */
/* looks at the occupancy -- with one WAVE per ray: lane j starts somewhere inside the ray's range, falls onto the sequential       */
/* walk at its first voxel face and must land exactly on lane j+1's start (else the window ends there), so the lists are the        */
/* one-thread-per-ray walk bit for bit (occupied voxels only).  `walked`: umhs_march_walk_workspace_bytes(n_rays) bytes (4.6 KB per ray); pass it to*/
/* count / write / scratch, which then only replay it (walked == NULL: they walk the grid themselves, one thread per ray).          */
/* Two passes: umhs_march_count -> counts [R] (caller scans them into packed_info), umhs_march_write -> packed        */
/* t_starts / t_ends [N] fp32 and ray_indices [N] int64.  umhs_visibility: mask[n] = T_n >= early_stop_eps &&         */
/* (alpha_thre <= 0 || alpha_n >= alpha_thre) with sigma from the density-only field forward.                        */
/* ------------------------------------------------------------------------------------------ */
size_t umhs_march_walk_workspace_bytes(int64_t n_rays);
int umhs_march_walk(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                    const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane, const float* nears,
                    const float* fars, const float* jitter, float jitter_step, void* walked, size_t walked_bytes,
                    umhs_stream_t stream);
int umhs_march_count(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                     const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                     float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                     float jitter_step, int64_t* counts, const void* walked, size_t walked_bytes, umhs_stream_t stream);
int umhs_march_write(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                     const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                     float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                     float jitter_step, const int64_t* packed_info, float* t_starts, float* t_ends, int64_t* ray_indices,
                     const void* walked, size_t walked_bytes, umhs_stream_t stream);
/* Single pass instead of count + write: umhs_march_scratch counts AND parks the first `cap` samples of ray r in             */
/* scratch_t0/t1[r*cap + i]; after the caller's scan, umhs_march_compact moves them to their packed places.  A count > cap     */
/* means that ray overflowed its row (whose last slot then holds garbage): fall back to umhs_march_write for the batch.       */
int umhs_march_scratch(const float* origins, const float* directions, int64_t n_rays, const uint8_t* binaries,
                       const float* roi_aabb_host6, int levels, int resolution, float near_plane, float far_plane,
                       float step_size, float cone_angle, const float* nears, const float* fars, const float* jitter,
                       float jitter_step, int cap, int64_t* counts, float* scratch_t0, float* scratch_t1, const void* walked,
                       size_t walked_bytes, umhs_stream_t stream);
int umhs_march_compact(const int64_t* packed_info, int64_t n_rays, int cap, const float* scratch_t0, const float* scratch_t1,
                       float* t_starts, float* t_ends, int64_t* ray_indices, umhs_stream_t stream);
int umhs_visibility(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                    int64_t n_rays, int64_t n, float early_stop_eps, float alpha_thre, uint8_t* mask, umhs_stream_t stream);
/* `jitter` [R] or NULL (all three march entry points): the near plane of ray r is nears[r] (or near_plane) + jitter[r] *      */
/* jitter_step -- nerfacc's stratified start, folded into the walk instead of two fills, a multiply and an add in front of it.  */
/* umhs_visibility_count also returns kept[R], the survivors per ray; umhs_ray_prefix turns per-ray counts into packed_info     */
/* [R,2] = (exclusive prefix, count) and stats[2] = (total, longest) -- used for the marched candidates and for the survivors;   */
/* umhs_sample_midpoints = origins[ri] + directions[ri] * (t_starts + t_ends) / 2 (VolumetricSampler's sigma_fn positions);     */
/* umhs_compact_samples moves the survivors (order within the ray kept) to their packed places and gathers their ray's origin,   */
/* direction and camera index (camera_indices / out_camera_indices both NULL or both set); out_sel[j] = candidate index of       */
/* survivor j.  Together: the sampler's torch.nonzero + 3 index_select + 3 gathers + pack_info as one launch after its host sync. */
int umhs_visibility_count(const float* sigma, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                          int64_t n_rays, int64_t n, float early_stop_eps, float alpha_thre, uint8_t* mask, int64_t* kept,
                          umhs_stream_t stream);
int umhs_ray_prefix(const int64_t* counts, int64_t n_rays, int64_t* packed_info, int64_t* stats, umhs_stream_t stream);
int umhs_sample_midpoints(const float* origins, const float* directions, const int64_t* ray_indices, const float* t_starts,
                          const float* t_ends, int64_t n, float* positions, umhs_stream_t stream);
int umhs_compact_samples(const uint8_t* mask, const int64_t* packed_in, const int64_t* packed_out, int64_t n_rays,
                         const float* t_starts, const float* t_ends, const float* origins, const float* directions,
                         const int64_t* camera_indices, int64_t* out_ray_indices, float* out_t_starts, float* out_t_ends,
                         float* out_origins, float* out_directions, int64_t* out_camera_indices, int64_t* out_sel,
                         umhs_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* pixel sampler, ray generator and ground-truth gather (images resident in HBM, --images-on-gpu).    */
/* Replaces, under UMHSDataManager.next_train (umhs_datamanager.py:95-108), nerfstudio's PixelSampler.sample          */
/* (indices = long(rand[R,3] * (n, H, W)); batch[key] = stack[c, y, x]) and RayGenerator -> Cameras.generate_rays     */
/* (perspective, no distortion: d = R_c2w * ((x+.5-cx)/fx, -(y+.5-cy)/fy, -1) normalised, origin = t_c2w,             */
/* pixel_area from the +x / +y neighbour directions).  indices [R,3] int64 rows (camera, y, x); c2w [n,3,4];          */
/* intrinsics [n,4] = (fx, fy, cx, cy); stack [n,H,W,K] fp32 or uint8 (scaled by 1/255); pixel_area /                 */
/* directions_norm [R] optional.                                                                                      */
/* ------------------------------------------------------------------------------------------ */
int umhs_pixel_indices(const float* uniform, int64_t n_rays, int64_t n_images, int64_t height, int64_t width,
                       int64_t* indices, umhs_stream_t stream);
int umhs_raygen(const int64_t* indices, const float* c2w, const float* intrinsics, int64_t n_rays, int64_t n_cams,
                float* origins, float* directions, float* pixel_area, float* directions_norm, umhs_stream_t stream);
int umhs_pixel_gather(const int64_t* indices, const void* stack, int src_is_u8, int64_t n_images, int64_t height,
                      int64_t width, int n_channels, int64_t n_rays, float* out, umhs_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* image metrics of the eval path, get_image_metrics_and_images (umhs_model.py:407-453), on channel-last */
/* images [H*W, K] as rendered.  umhs_pixel_metrics: partial[b] = {sum (p-g)^2, sum of finite spectral angles          */
/* acos(clamp(<p,g>/(|p||g|))), count of finite angles} per block b < n_partial (PSNR :430,444, RMSE :452, SAM :447).   */
/* umhs_ssim: torchmetrics==1.5.2 structural_similarity_index_measure (:431,445; 11x11 gaussian sigma 1.5, k1 .01,      */
/* k2 .03): partial[] = per-block sums of the SSIM index over the (H-10)(W-10)K windows inside the image;               */
/* data_range = DEVICE float (max(a.max-a.min, b.max-b.min) for the default data_range=None).                           */
/* The caller adds the partials (fixed order: reproducible) and divides.                                                */
/* ------------------------------------------------------------------------------------------ */
int umhs_pixel_metrics(const float* pred, const float* gt, int64_t n_pixels, int n_channels, double* partial, int n_partial,
                       umhs_stream_t stream);
int64_t umhs_ssim_partials(int height, int width, int n_channels);
int umhs_ssim(const float* a, const float* b, int height, int width, int n_channels, const float* data_range,
              double* partial, int64_t n_partial, umhs_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Optimizer: torch.optim.Adam step for param group "fields" (AdamOptimizerConfig(lr=2e-2,      */
/* eps=1e-15), umhs_config.py:59-64) over one flat fp32 buffer, with the clamp_endmembers        */
/* callback (umhs_model.py:568-572) fused for elements [clamp_begin, clamp_end).  grad_scale     */
/* multiplies the gradient first (1/world_size for averaged DDP gradients).                      */
/* ------------------------------------------------------------------------------------------ */
int umhs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, int64_t step, float grad_scale, int64_t clamp_begin,
                   int64_t clamp_end, umhs_stream_t stream);
/* The same update on selected 2-float rows of the buffers only (rows: device int64 [n_rows], row r = elements 2r, 2r+1): the   */
/* coarse hash levels use a small fixed subset of their slots; every other row has g = m = v = 0 for ever and is skipped.       */
int umhs_adam_step_rows(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* rows,
                        int64_t n_rows, float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale,
                        umhs_stream_t stream);
/* umhs_adam_step_rows on `rows` and umhs_adam_step on elements [range_begin, range_begin + range_count) of the same flat buffers  */
/* (clamp range in absolute elements) in ONE launch: what is left for the optimizer when the dense hash levels were updated by     */
/* umhs_hashgrid_bwd_apply_adam.                                                                                                   */
int umhs_adam_step_rows_range(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* rows,
                              int64_t n_rows, int64_t range_begin, int64_t range_count, float lr, float beta1, float beta2,
                              float eps, int64_t step, float grad_scale, int64_t clamp_begin, int64_t clamp_end,
                              umhs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UMHS_HIP_H */
