// Bicubic interpolation pieces shared by the V2 decode kernels (decode_v2.hip) and the tracker-feature kernel
// (post_v2.hip).  With -ffp-contract=off every fp32 operation is the single IEEE operation the oracle
// performs, in the same order.
#pragma once
#include <hip/hip_runtime.h>

// ---------------------------------------------------------------------------
// bicubic pieces (upstream cuda.hu cubicInterpolate / bicubicInterpolate)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float cubic(float v0, float v1, float v2, float v3, float d) {
  const float a = (((-0.5f * v0 + 1.5f * v1) - 1.5f * v2) + 0.5f * v3);
  const float b = (((v0 - 2.5f * v1) + 2.0f * v2) - 0.5f * v3);
  const float c = 0.5f * (v0 - v2);
  return ((((a * d) * d) * d + (b * d) * d) - c * d) + v1;
}

struct Taps {
  int i0, i1, i2, i3;
  float d;
};

__device__ __forceinline__ Taps src_taps(int x, int n_target, int n_source) {
  const float xs = (((float)x + 0.5f) * (float)n_source) / (float)n_target - 0.5f;
  int i1 = (int)floorf(xs);
  i1 = i1 < 0 ? 0 : (i1 > n_source - 1 ? n_source - 1 : i1);
  Taps t;
  t.i1 = i1;
  t.i0 = i1 - 1 < 0 ? 0 : i1 - 1;
  t.i2 = i1 + 1 > n_source - 1 ? n_source - 1 : i1 + 1;
  t.i3 = t.i2 + 1 > n_source - 1 ? n_source - 1 : t.i2 + 1;
  t.d = xs - (float)i1;
  return t;
}

// packed taps kept in LDS tables (one fp32 division per output column/row per block
// instead of one per pixel)
struct __attribute__((aligned(4))) TapsP {
  unsigned short i0, i1, i2, i3;
  float d;
};
__device__ __forceinline__ TapsP pack_taps(const Taps& t) {
  TapsP p;
  p.i0 = (unsigned short)t.i0;
  p.i1 = (unsigned short)t.i1;
  p.i2 = (unsigned short)t.i2;
  p.i3 = (unsigned short)t.i3;
  p.d = t.d;
  return p;
}

__device__ __forceinline__ float cubic_row(const float* r, const TapsP& tx) {
  return cubic(r[tx.i0], r[tx.i1], r[tx.i2], r[tx.i3], tx.d);
}
__device__ __forceinline__ float bicubic_tp(const float* src, int ws, const TapsP& tx, const TapsP& ty) {
  const float t0 = cubic_row(src + ty.i0 * ws, tx), t1 = cubic_row(src + ty.i1 * ws, tx);
  const float t2 = cubic_row(src + ty.i2 * ws, tx), t3 = cubic_row(src + ty.i3 * ws, tx);
  return cubic(t0, t1, t2, t3, ty.d);
}

// one pixel of the upsampled map from a source plane (LDS or global)
__device__ __forceinline__ float bicubic_at(const float* src, int hs, int ws, int Ht, int Wt, int x, int y) {
  const TapsP tx = pack_taps(src_taps(x, Wt, ws)), ty = pack_taps(src_taps(y, Ht, hs));
  return bicubic_tp(src, ws, tx, ty);
}


// OpenPose --heatmaps_scale (upstream poseExtractorNet.cpp, restated from the published 1.7.0 source; unverified
// here): applied to the RESIZED maps that datum.poseHeatMaps exports.  heat = body part / background channel
// ([0,1] data), otherwise PAF channel ([-1,1] data).  0: [-1,1]   1: [0,1]   2: integer-rounded [0,255]   3: as is
__device__ __forceinline__ float heatmap_scale_value(float v, bool heat, int mode) {
  if (mode == 3) return v;
  if (heat) {
    const float t = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    if (mode == 0) return t * 2.f - 1.f;
    if (mode == 2) return (float)(int)(t * 255.f + 0.5f);
    return t;
  }
  const float t = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
  if (mode == 1) return t * 0.5f + 0.5f;
  if (mode == 2) {
    const float r = (float)(int)((t * 128.5f + 128.5f) + 0.5f);
    return r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
  }
  return t;
}
