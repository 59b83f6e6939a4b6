// image metrics of the eval path (umhs_model.py:407-453) for gfx950: PSNR / RMSE (sum of squared errors),
// SAM (torchmetrics SpectralAngleMapper(reduction="none") + nanmean, :177,447) and SSIM (torchmetrics==1.5.2
// structural_similarity_index_measure: 11x11 gaussian, sigma 1.5, k1 .01, k2 .03, data_range from the images, mean over
// the windows that lie fully inside the image).  Images are channel-last [H,W,K] exactly as the renderer produces them (the
// reference moves axes to [1,K,H,W] first).  Every block writes its partial sums (double) to its own slot; the caller
// adds the slots, so results are reproducible.  LPIPS needs pretrained weights (no network) and is out of scope.
#include "umhs_common.h"

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// partial[b] = {sum (p-g)^2, sum of finite spectral angles, number of finite angles}
__global__ __launch_bounds__(256) void pixel_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            int64_t n_pix, int K, double* __restrict__ partial) {
  __shared__ double sm[4];
  double sse = 0.0, sam = 0.0, cnt = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * 256) {
    float dot = 0.0f, pp = 0.0f, gg = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float a = pred[p * K + k], b = gt[p * K + k], e = a - b;
      sse += (double)(e * e);
      dot += a * b, pp += a * a, gg += b * b;
    }
    const float c = dot / (sqrtf(pp) * sqrtf(gg));  // 0/0 -> NaN, skipped like torch.nanmean
    if (c == c) sam += (double)acosf(fminf(fmaxf(c, -1.0f), 1.0f)), cnt += 1.0;
  }
  sse = block_sum_256(sse, sm), sam = block_sum_256(sam, sm), cnt = block_sum_256(cnt, sm);
  if (threadIdx.x == 0) partial[3 * blockIdx.x] = sse, partial[3 * blockIdx.x + 1] = sam, partial[3 * blockIdx.x + 2] = cnt;
}

extern "C" int umhs_pixel_metrics(const float* pred, const float* gt, int64_t n_pixels, int n_channels, double* partial,
                                  int n_partial, umhs_stream_t stream) {
  if (n_pixels < 0 || n_channels < 1 || !pred || !gt || !partial || n_partial < 1) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(pixel_metrics_kernel, dim3((unsigned)n_partial), dim3(256), 0, umhs_s(stream), pred, gt, n_pixels,
                     n_channels, partial);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// One block = 32x8 window positions of one channel.  Window (oy, ox) covers image rows oy..oy+10, cols ox..ox+10.
constexpr int SS_TX = 32, SS_TY = 8, SS_K = 11;
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W, int K,
                                                   const float* __restrict__ data_range, double* __restrict__ partial) {
#pragma clang fp contract(off)
  __shared__ float ta[SS_TY + SS_K - 1][SS_TX + SS_K - 1], tb[SS_TY + SS_K - 1][SS_TX + SS_K - 1];
  __shared__ float hz[5][SS_TY + SS_K - 1][SS_TX];
  __shared__ float gk[SS_K];
  __shared__ double sm[4];
  const int tid = threadIdx.x, c = blockIdx.z, x0 = blockIdx.x * SS_TX, y0 = blockIdx.y * SS_TY;
  if (tid < SS_K) {
    float s = 0.0f, g[SS_K];
    for (int i = 0; i < SS_K; ++i) {
      const float d = (float)(i - 5) / 1.5f;
      g[i] = expf(-(d * d) / 2.0f), s += g[i];
    }
    gk[tid] = g[tid] / s;
  }
  for (int i = tid; i < (SS_TY + SS_K - 1) * (SS_TX + SS_K - 1); i += 256) {
    const int ry = i / (SS_TX + SS_K - 1), rx = i % (SS_TX + SS_K - 1), y = y0 + ry, x = x0 + rx;
    const bool in = y < H && x < W;
    ta[ry][rx] = in ? a[((int64_t)y * W + x) * K + c] : 0.0f;
    tb[ry][rx] = in ? b[((int64_t)y * W + x) * K + c] : 0.0f;
  }
  __syncthreads();
  for (int i = tid; i < (SS_TY + SS_K - 1) * SS_TX; i += 256) {
    const int ry = i / SS_TX, rx = i % SS_TX;
    float s[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < SS_K; ++t) {
      const float p = ta[ry][rx + t], q = tb[ry][rx + t], w = gk[t];
      s[0] += w * p, s[1] += w * q, s[2] += w * (p * p), s[3] += w * (q * q), s[4] += w * (p * q);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) hz[j][ry][rx] = s[j];
  }
  __syncthreads();
  const int lx = tid % SS_TX, ly = tid / SS_TX;
  double v = 0.0;
  if (x0 + lx + SS_K <= W && y0 + ly + SS_K <= H) {
    float s[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < SS_K; ++t) {
      const float w = gk[t];
#pragma unroll
      for (int j = 0; j < 5; ++j) s[j] += w * hz[j][ly + t][lx];
    }
    const float L = *data_range, c1 = (0.01f * L) * (0.01f * L), c2 = (0.03f * L) * (0.03f * L);
    const float mpp = s[0] * s[0], mqq = s[1] * s[1], mpq = s[0] * s[1];
    const float vp = fmaxf(s[2] - mpp, 0.0f), vq = fmaxf(s[3] - mqq, 0.0f), cov = s[4] - mpq;
    v = (double)(((2.0f * mpq + c1) * (2.0f * cov + c2)) / ((mpp + mqq + c1) * (vp + vq + c2)));
  }
  v = block_sum_256(v, sm);
  if (tid == 0) partial[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = v;
}

extern "C" int64_t umhs_ssim_partials(int height, int width, int n_channels) {
  if (height < SS_K || width < SS_K || n_channels < 1) return 0;
  return (int64_t)((width - SS_K + 1 + SS_TX - 1) / SS_TX) * ((height - SS_K + 1 + SS_TY - 1) / SS_TY) * n_channels;
}

extern "C" int umhs_ssim(const float* a, const float* b, int height, int width, int n_channels, const float* data_range,
                         double* partial, int64_t n_partial, umhs_stream_t stream) {
  if (!a || !b || !data_range || !partial) return UMHS_ERR_ARG;
  const int64_t need = umhs_ssim_partials(height, width, n_channels);
  if (need == 0 || n_channels > 65535) return UMHS_ERR_UNSUPPORTED;
  if (n_partial < need) return UMHS_ERR_WORKSPACE;
  const dim3 grid((unsigned)((width - SS_K + 1 + SS_TX - 1) / SS_TX), (unsigned)((height - SS_K + 1 + SS_TY - 1) / SS_TY), (unsigned)n_channels);
  hipLaunchKernelGGL(ssim_kernel, grid, dim3(256), 0, umhs_s(stream), a, b, height, width, n_channels, data_range, partial);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}
