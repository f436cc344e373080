// Frame preprocessing for gfx950: u8 HWC BGR frames -> bf16 NHWC network input
// (channels padded 3 -> 8, zero border), fused resize + normalise + layout.
//
// PH_PRE_CV2 replaces cv2.dnn.blobFromImage(image, 1/255, (Wn,Hn), mean 0,
// swapRB=False, crop=False) (openpose/opencv/skeleton.py:66-73): cv::resize
// INTER_LINEAR on uint8 in OpenCV's 11-bit fixed point (restated from public
// OpenCV 4.5.x resize.cpp -- unverified against a real cv2, none is available),
// then * 1/255.
// PH_PRE_NATIVE replaces OpenPose's cvMatToOpInput inside emplaceAndPop
// (openpose/native/python/skeleton.py:447): keep-aspect scale anchored at the
// top-left, zero fill, bilinear with source coordinates quantised to 1/32 px
// as cv::warpAffine does, then x/256 - 0.5.
#include "common.h"

__device__ __forceinline__ int cv_round_short(float v) {
  int r = __float2int_rn(v);  // round half to even, as cvRound
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

__global__ void preprocess_kernel(const uint8_t* __restrict__ frames, int B, int H0, int W0, int Hn, int Wn,
                                  int mode, double scale_x, double scale_y, double inv_scale,
                                  bf16* __restrict__ out, int out_Hp, int out_Wp, int out_Cs, int out_org) {
  const size_t total = (size_t)B * Hn * Wn;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int x = idx % Wn;
    size_t t = idx / Wn;
    const int y = t % Hn;
    const int b = t / Hn;
    const uint8_t* img = frames + (size_t)b * H0 * W0 * 3;
    float v[3];
    if (mode == 0) {
      if (H0 == Hn && W0 == Wn) {
        const uint8_t* p = img + ((size_t)y * W0 + x) * 3;
        for (int c = 0; c < 3; c++) v[c] = (float)p[c];
      } else {
        float fx = (float)((x + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= W0 - 1) { fx = 0.f; sx = W0 - 1; }
        float fy = (float)((y + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        const int a0 = cv_round_short((1.f - fx) * 2048.f), a1 = cv_round_short(fx * 2048.f);
        const int b0 = cv_round_short((1.f - fy) * 2048.f), b1 = cv_round_short(fy * 2048.f);
        const int sx1 = sx + 1 > W0 - 1 ? W0 - 1 : sx + 1;
        int sy0 = sy, sy1 = sy + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 > H0 - 1 ? H0 - 1 : sy0);
        sy1 = sy1 < 0 ? 0 : (sy1 > H0 - 1 ? H0 - 1 : sy1);
        const uint8_t* r0 = img + (size_t)sy0 * W0 * 3;
        const uint8_t* r1 = img + (size_t)sy1 * W0 * 3;
        for (int c = 0; c < 3; c++) {
          const int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
          const int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
          const int d = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
          v[c] = (float)(d < 0 ? 0 : (d > 255 ? 255 : d));
        }
      }
      for (int c = 0; c < 3; c++) v[c] = v[c] * (1.0f / 255.0f);
    } else {
      // source position = dst / scale, quantised to 1/32 px
      const int X = __double2int_rn(x * inv_scale * 32.0), Y = __double2int_rn(y * inv_scale * 32.0);
      const int sx = X >> 5, sy = Y >> 5;
      const float fx = (float)(X & 31) * (1.f / 32.f), fy = (float)(Y & 31) * (1.f / 32.f);
      for (int c = 0; c < 3; c++) {
        float p[4];
        for (int k = 0; k < 4; k++) {
          const int xx = sx + (k & 1), yy = sy + (k >> 1);
          p[k] = (xx >= 0 && xx < W0 && yy >= 0 && yy < H0) ? (float)img[((size_t)yy * W0 + xx) * 3 + c] : 0.f;
        }
        const float top = p[0] * (1.f - fx) + p[1] * fx, bot = p[2] * (1.f - fx) + p[3] * fx;
        const float val = top * (1.f - fy) + bot * fy;
        v[c] = floorf(val + 0.5f);  // back to u8 as warpAffine does
        v[c] = v[c] * (1.0f / 256.0f) - 0.5f;
      }
    }
    bf16x8 o;
    for (int c = 0; c < 3; c++) o[c] = (bf16)v[c];
    for (int c = 3; c < 8; c++) o[c] = (bf16)0.f;
    *(bf16x8*)(out + (size_t)out_org + ((size_t)(b * out_Hp + y) * out_Wp + x) * out_Cs) = o;
  }
}

int ph_launch_preprocess(const uint8_t* frames, int B, int H0, int W0, int Hn, int Wn, int mode, bf16* out,
                         int out_Hp, int out_Wp, int out_Cs, int out_org, hipStream_t stream) {
  const double scale_x = (double)W0 / Wn, scale_y = (double)H0 / Hn;
  double s = (Wn - 1) / (double)(W0 - 1 > 0 ? W0 - 1 : 1);
  const double s2 = (Hn - 1) / (double)(H0 - 1 > 0 ? H0 - 1 : 1);
  s = s < s2 ? s : s2;
  const size_t total = (size_t)B * Hn * Wn;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(256), 0, stream, frames, B, H0, W0, Hn, Wn, mode,
                     scale_x, scale_y, 1.0 / s, out, out_Hp, out_Wp, out_Cs, out_org);
  PH_CHECK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// Measurement helper (tools/hbm_copy.py, not part of include/posehip.h): achievable HBM bandwidth of a plain
// 16-byte-per-lane streaming copy, the yardstick for the HBM-bound kernels.
// ---------------------------------------------------------------------------
__global__ void ph_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

extern "C" int ph_debug_copy_bandwidth(size_t bytes, int reps, float* gb_per_s) {
  void *a = nullptr, *b = nullptr;
  PH_CHECK(hipMalloc(&a, bytes));
  PH_CHECK(hipMalloc(&b, bytes));
  PH_CHECK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  PH_CHECK(hipEventCreate(&e0));
  PH_CHECK(hipEventCreate(&e1));
  const size_t n = bytes / 16;
  hipLaunchKernelGGL(ph_copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n);
  PH_CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(ph_copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n);
  PH_CHECK(hipEventRecord(e1, 0));
  PH_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  PH_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *gb_per_s = (float)(2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9);  // read + write
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  return 0;
}
