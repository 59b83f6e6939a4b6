// pixel sampler / ray generator / ground-truth gather for gfx950.
// Replaces what UMHSDataManager.next_train (umhs_datamanager.py:95-108) reaches in nerfstudio==1.1.5 with
// --images-on-gpu: PixelSampler.sample (indices (camera, y, x) -> rows of the cached image stacks) and RayGenerator ->
// Cameras.generate_rays (perspective, no distortion).  nerfstudio's source is not available offline; the arithmetic below
// restates its published behaviour (oracle/torch_ref.py generate_rays / gather_pixels).  Both kernels are HBM-bound:
// ray generation moves 24 B in + 28 B out per ray, the gather one (B+3)-float row per ray from a stack of n*H*W rows.
#include "umhs_common.h"

// indices [R,3] int64 (camera, y, x); c2w [n,3,4]; intr [n,4] = (fx, fy, cx, cy)
__global__ __launch_bounds__(256) void raygen_kernel(const int64_t* __restrict__ indices, const float* __restrict__ c2w,
                                                     const float* __restrict__ intr, int64_t n_rays, int64_t n_cams,
                                                     float* __restrict__ origins, float* __restrict__ directions,
                                                     float* __restrict__ pixel_area, float* __restrict__ dir_norm) {
#pragma clang fp contract(off)
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  int64_t c = indices[3 * r];
  c = c < 0 ? 0 : (c >= n_cams ? n_cams - 1 : c);  // indices are validated on the host; never read out of bounds
  const float y = (float)indices[3 * r + 1] + 0.5f, x = (float)indices[3 * r + 2] + 0.5f;  // pixel centres
  const float fx = intr[4 * c], fy = intr[4 * c + 1], cx = intr[4 * c + 2], cy = intr[4 * c + 3];
  const float* M = c2w + 12 * c;
  // three directions: the pixel, its +x and its +y neighbour (pixel_area = |d - dx| * |d - dy|)
  const float px[3] = {(x - cx) / fx, (x - cx + 1.0f) / fx, (x - cx) / fx};
  const float py[3] = {-(y - cy) / fy, -(y - cy) / fy, -(y - cy + 1.0f) / fy};
  float d[3][3], nrm0 = 0.0f;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float v[3], sq = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] = (px[s] * M[4 * k] + py[s] * M[4 * k + 1]) + (-1.0f) * M[4 * k + 2];
      sq += v[k] * v[k];
    }
    const float nrm = fmaxf(sqrtf(sq), 1.1920928955078125e-07f);
    if (s == 0) nrm0 = nrm;
#pragma unroll
    for (int k = 0; k < 3; ++k) d[s][k] = v[k] / nrm;
  }
  float dx = 0.0f, dy = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float a = d[0][k] - d[1][k], b = d[0][k] - d[2][k];
    dx += a * a, dy += b * b;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) origins[3 * r + k] = M[4 * k + 3], directions[3 * r + k] = d[0][k];
  if (pixel_area) pixel_area[r] = sqrtf(dx) * sqrtf(dy);
  if (dir_norm) dir_norm[r] = nrm0;
}

extern "C" int umhs_raygen(const int64_t* indices, const float* c2w, const float* intrinsics, int64_t n_rays,
                           int64_t n_cams, float* origins, float* directions, float* pixel_area, float* directions_norm,
                           umhs_stream_t stream) {
  if (n_rays == 0) return UMHS_OK;
  if (n_rays < 0 || n_cams < 1 || !indices || !c2w || !intrinsics || !origins || !directions) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, umhs_s(stream), indices, c2w,
                     intrinsics, n_rays, n_cams, origins, directions, pixel_area, directions_norm);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// out[r, :] = stack[c, y, x, :]  (stack [n,H,W,K] fp32, or uint8 scaled by 1/255 as nerfstudio's get_image_float32 does)
template <typename SRC>
__global__ __launch_bounds__(256) void pixel_gather_kernel(const int64_t* __restrict__ indices, const SRC* __restrict__ stack,
                                                           int64_t n, int64_t H, int64_t W, int K, int64_t n_rays,
                                                           float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rays * K) return;
  const int64_t r = i / K;
  const int k = (int)(i - r * K);
  int64_t c = indices[3 * r], y = indices[3 * r + 1], x = indices[3 * r + 2];
  c = c < 0 ? 0 : (c >= n ? n - 1 : c), y = y < 0 ? 0 : (y >= H ? H - 1 : y), x = x < 0 ? 0 : (x >= W ? W - 1 : x);
  const SRC v = stack[((c * H + y) * W + x) * K + k];
  if constexpr (sizeof(SRC) == 1)
    out[i] = (float)v / 255.0f;
  else
    out[i] = v;
}

extern "C" int umhs_pixel_gather(const int64_t* indices, const void* stack, int src_is_u8, int64_t n_images, int64_t height,
                                 int64_t width, int n_channels, int64_t n_rays, float* out, umhs_stream_t stream) {
  if (n_rays == 0) return UMHS_OK;
  if (n_rays < 0 || n_images < 1 || height < 1 || width < 1 || n_channels < 1 || !indices || !stack || !out) return UMHS_ERR_ARG;
  const int64_t total = n_rays * n_channels;
  if ((total + 255) / 256 > 0x7fffffffLL) return UMHS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (src_is_u8)
    hipLaunchKernelGGL(pixel_gather_kernel<uint8_t>, grid, dim3(256), 0, umhs_s(stream), indices, (const uint8_t*)stack,
                       n_images, height, width, n_channels, n_rays, out);
  else
    hipLaunchKernelGGL(pixel_gather_kernel<float>, grid, dim3(256), 0, umhs_s(stream), indices, (const float*)stack, n_images,
                       height, width, n_channels, n_rays, out);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}

// indices[r] = (long)(u[r] * (n, H, W))  -- PixelSampler.sample_method; u from torch.rand so the stream of draws is torch's
__global__ __launch_bounds__(256) void pixel_indices_kernel(const float* __restrict__ u, int64_t n_rays, float n, float H,
                                                            float W, int64_t* __restrict__ indices) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * n_rays) return;
  const int k = (int)(i % 3);
  indices[i] = (int64_t)(u[i] * (k == 0 ? n : (k == 1 ? H : W)));
}

extern "C" int umhs_pixel_indices(const float* uniform, int64_t n_rays, int64_t n_images, int64_t height, int64_t width,
                                  int64_t* indices, umhs_stream_t stream) {
  if (n_rays == 0) return UMHS_OK;
  if (n_rays < 0 || n_images < 1 || height < 1 || width < 1 || !uniform || !indices) return UMHS_ERR_ARG;
  hipLaunchKernelGGL(pixel_indices_kernel, dim3((unsigned)((3 * n_rays + 255) / 256)), dim3(256), 0, umhs_s(stream), uniform,
                     n_rays, (float)n_images, (float)height, (float)width, indices);
  UMHS_CHECK_LAUNCH();
  return UMHS_OK;
}
