// Frame preprocessing for gfx950: u8 HWC BGR frames -> bf16 NHWC network input
// (channels padded 3 -> 8, zero border), fused resize + normalise + layout.
//
// PH_PRE_CV2 replaces cv2.dnn.blobFromImage(image, 1/255, (Wn,Hn), mean 0,
// swapRB=False, crop=False) (openpose/opencv/skeleton.py:66-73): cv::resize
// INTER_LINEAR on uint8 in OpenCV's 11-bit fixed point (restated from public
// OpenCV 4.5.x resize.cpp -- unverified against a real cv2, none is available),
// then * 1/255.
// PH_PRE_NATIVE replaces OpenPose's cvMatToOpInput inside emplaceAndPop
// (openpose/native/python/skeleton.py:447): resizeFixedAspectRatio = cv::warpAffine
// with the matrix diag(s, s), s = min((Wn-1)/(W0-1), (Hn-1)/(H0-1)), zero fill,
// INTER_LINEAR for s <= 1 (OpenPose asks for INTER_AREA, which warpAffine replaces)
// and INTER_CUBIC for s > 1, in OpenCV's fixed point (1/32-px phases, 15-bit
// weights, (sum + 16384) >> 15), then x/256 - 0.5.  Bit-exact against
// oracle/preprocess.py:cv_mat_to_op_input, which states the sources.
#include "common.h"

__device__ __forceinline__ int cv_round_short(float v) {
  int r = __float2int_rn(v);  // round half to even, as cvRound
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

// OpenCV interpolateCubic (A = -0.75), fp32
__device__ __forceinline__ void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

__global__ void preprocess_kernel(const uint8_t* __restrict__ frames, int B, int H0, int W0, int Hn, int Wn,
                                  int mode, double scale_x, double scale_y, double inv_scale, int cubic,
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
      // cv::warpAffine's fixed-point source coordinates: AB_BITS = 10, round_delta = 16, 1/32-px phases
      const int X = (int)((16 + __double2ll_rn(inv_scale * (double)x * 1024.0)) >> 5);
      const int Y = (int)((16 + __double2ll_rn((inv_scale * (double)y + 0.0) * 1024.0)) >> 5);
      const int ax = X & 31, ay = Y & 31;
      int acc[3] = {0, 0, 0};
      if (!cubic) {
        const int sx = X >> 5, sy = Y >> 5;
        // BilinearTab_i[ay*32 + ax]: exact products (32-ay)(32-ax)*32 ...; entry 0 is {32767, 0, 0, 1} (the
        // saturated 32768 and the library's sum fix-up)
        int w[4] = {(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32};
        if ((ax | ay) == 0) {
          w[0] = 32767;
          w[3] = 1;
        }
        for (int k = 0; k < 4; k++) {
          const int xx = sx + (k & 1), yy = sy + (k >> 1);
          // (a tap of weight zero adds nothing and is not fetched: frames that already have the net size -- phase
          // 0 in both directions -- read 2 source pixels instead of 4; the kernel is bound by the number of its
          // byte loads, not by their bytes)
          if (w[k] != 0 && xx >= 0 && xx < W0 && yy >= 0 && yy < H0) {
            const uint8_t* p = img + ((size_t)yy * W0 + xx) * 3;
            for (int c = 0; c < 3; c++) acc[c] += (int)p[c] * w[k];
          }
        }
      } else {
        const int sx = (X >> 5) - 1, sy = (Y >> 5) - 1;
        float cx[4], cy[4];
        cubic_coeffs((float)ax * (1.0f / 32), cx);
        cubic_coeffs((float)ay * (1.0f / 32), cy);
        int w[16], isum = 0;
        for (int i = 0; i < 4; i++)
          for (int j = 0; j < 4; j++) {
            w[i * 4 + j] = cv_round_short((cy[i] * cx[j]) * 32768.0f);
            isum += w[i * 4 + j];
          }
        if (isum != 32768) {  // initInterTab2D's fix-up over rows/columns 2..3 of the 4x4 weights
          const int diff = isum - 32768;
          int Mk = 10, mk = 10;
          for (int k1 = 2; k1 < 4; k1++)
            for (int k2 = 2; k2 < 4; k2++) {
              const int q = k1 * 4 + k2;
              if (w[q] < w[mk]) mk = q;
              else if (w[q] > w[Mk]) Mk = q;
            }
          if (diff < 0) w[Mk] -= diff;
          else w[mk] -= diff;
        }
        for (int i = 0; i < 4; i++) {
          const int yy = sy + i;
          if (yy < 0 || yy >= H0) continue;
          for (int j = 0; j < 4; j++) {
            const int xx = sx + j;
            if (xx < 0 || xx >= W0) continue;
            const uint8_t* p = img + ((size_t)yy * W0 + xx) * 3;
            for (int c = 0; c < 3; c++) acc[c] += (int)p[c] * w[i * 4 + j];
          }
        }
      }
      for (int c = 0; c < 3; c++) {
        const int d = (acc[c] + 16384) >> 15;  // FixedPtCast<int, uchar, 15>
        v[c] = (float)(d < 0 ? 0 : (d > 255 ? 255 : d)) * (1.0f / 256.0f) - 0.5f;
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
  // OpenPose resizeGetScaleFactor, then the inverse matrix entry as cv::warpAffine computes it (D = 1/(s*s), s*D)
  double s = (Wn - 1) / (double)(W0 - 1 > 0 ? W0 - 1 : 1);
  const double s2 = (Hn - 1) / (double)(H0 - 1 > 0 ? H0 - 1 : 1);
  s = s < s2 ? s : s2;
  double D = s * s;
  D = D != 0 ? 1. / D : 0;
  const double inv = s * D;
  const size_t total = (size_t)B * Hn * Wn;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(256), 0, stream, frames, B, H0, W0, Hn, Wn, mode,
                     scale_x, scale_y, inv, s > 1.0 ? 1 : 0, out, out_Hp, out_Wp, out_Cs, out_org);
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
