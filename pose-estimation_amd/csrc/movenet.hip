// MoveNet pre-/post-model steps (BASELINE configs[3]).
//
// In the reference tree (movenet/utils/image_sequence_crop.py; the model itself is a remote TF-Hub graph):
//   crop_and_resize   :153-159 + the int32 cast of :171 -> movenet_crop_resize_kernel (HBM-bound gather)
//   run_inference's coordinate update :178-186 and determine_crop_region :94-150 (with init_crop_region :15-40,
//   torso_visible :43-56, determine_torso_and_body_range :59-91) -> movenet_update_kernel, one thread per stream,
//   float64 in the reference's operation order (bit-exact vs oracle/movenet.py).
// Not in the reference tree: the MultiPose head decode BASELINE configs[3] asks for; specified in
// oracle/movenet.py:multipose_decode (parity unpinned) -> movenet_centers_kernel + movenet_people_kernel.
// Compiled with -ffp-contract=off; fp32 '/' and sqrtf are the correctly rounded ones.
#include "common.h"

// ---- crop_and_resize (tf.image.crop_and_resize, bilinear, extrapolation 0) + cast to int32 --------------
__global__ void movenet_crop_resize_kernel(const uint8_t* __restrict__ frames, int B, int H, int W,
                                           const double* __restrict__ crop, int S, int32_t* __restrict__ out) {
  const size_t total = (size_t)B * S * S;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % S;
    const size_t t = i / S;
    const int y = t % S;
    const int b = t / S;
    const double* cr = crop + (size_t)b * 6;  // y_min, x_min, y_max, x_max, height, width
    const float y1 = (float)cr[0], x1 = (float)cr[1], y2 = (float)cr[2], x2 = (float)cr[3];
    float in_y, in_x;
    if (S > 1) {
      const float hs = ((y2 - y1) * (float)(H - 1)) / (float)(S - 1);
      const float ws = ((x2 - x1) * (float)(W - 1)) / (float)(S - 1);
      in_y = y1 * (float)(H - 1) + (float)y * hs;
      in_x = x1 * (float)(W - 1) + (float)x * ws;
    } else {
      in_y = (0.5f * (y1 + y2)) * (float)(H - 1);
      in_x = (0.5f * (x1 + x2)) * (float)(W - 1);
    }
    int32_t* o = out + i * 3;
    if (in_y < 0.f || in_y > (float)(H - 1) || in_x < 0.f || in_x > (float)(W - 1)) {
      o[0] = o[1] = o[2] = 0;
      continue;
    }
    const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
    const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
    const float yl = in_y - (float)top, xl = in_x - (float)left;
    const uint8_t* f = frames + (size_t)b * H * W * 3;
    const uint8_t* ptl = f + ((size_t)top * W + left) * 3;
    const uint8_t* ptr_ = f + ((size_t)top * W + right) * 3;
    const uint8_t* pbl = f + ((size_t)bot * W + left) * 3;
    const uint8_t* pbr = f + ((size_t)bot * W + right) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float tl = (float)ptl[c], tr = (float)ptr_[c], bl = (float)pbl[c], br = (float)pbr[c];
      const float tp = tl + (tr - tl) * xl;
      const float bt = bl + (br - bl) * xl;
      o[c] = (int32_t)(tp + (bt - tp) * yl);
    }
  }
}

// ---- coordinate update + next crop region, one thread per stream -------------------------------------------
__device__ void movenet_init_crop(double ih, double iw, double* o) {
  double bh, bw, y_min, x_min;
  if (iw > ih) {
    bh = iw / ih;
    bw = 1.0;
    y_min = (ih / 2 - iw / 2) / ih;
    x_min = 0.0;
  } else {
    bh = 1.0;
    bw = ih / iw;
    y_min = 0.0;
    x_min = (iw / 2 - ih / 2) / iw;
  }
  o[0] = y_min;
  o[1] = x_min;
  o[2] = y_min + bh;
  o[3] = x_min + bw;
  o[4] = bh;
  o[5] = bw;
}

__global__ void movenet_update_kernel(float* __restrict__ kp, const double* __restrict__ crop_in, int B, int ih_, int iw_,
                                      double* __restrict__ crop_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double ih = (double)ih_, iw = (double)iw_;
  float* k = kp + (size_t)b * 51;
  const double* ci = crop_in + (size_t)b * 6;
  double ty[17], tx[17];
  for (int j = 0; j < 17; j++) {  // crop -> image coordinates (normalised), stored as float32
    k[3 * j] = (float)((ci[0] * ih + ci[4] * ih * (double)k[3 * j]) / ih);
    k[3 * j + 1] = (float)((ci[1] * iw + ci[5] * iw * (double)k[3 * j + 1]) / iw);
    ty[j] = (double)k[3 * j] * ih;
    tx[j] = (double)k[3 * j + 1] * iw;
  }
  double* o = crop_out + (size_t)b * 6;
  const float thr = 0.2f;  // MIN_CROP_KEYPOINT_SCORE, compared as the float32 score against the Python float 0.2
  auto sc = [&](int j) { return (double)k[3 * j + 2]; };
  const bool visible = (sc(11) > 0.2 || sc(12) > 0.2) && (sc(5) > 0.2 || sc(6) > 0.2);
  (void)thr;
  if (!visible) {
    movenet_init_crop(ih, iw, o);
    return;
  }
  const double cy = (ty[11] + ty[12]) / 2, cx = (tx[11] + tx[12]) / 2;
  double mty = 0.0, mtx = 0.0, mby = 0.0, mbx = 0.0;
  const int torso[4] = {5, 6, 11, 12};
  for (int q = 0; q < 4; q++) {
    const double dy = fabs(cy - ty[torso[q]]), dx = fabs(cx - tx[torso[q]]);
    mty = dy > mty ? dy : mty;
    mtx = dx > mtx ? dx : mtx;
  }
  for (int j = 0; j < 17; j++) {
    if (sc(j) < 0.2) continue;
    const double dy = fabs(cy - ty[j]), dx = fabs(cx - tx[j]);
    mby = dy > mby ? dy : mby;
    mbx = dx > mbx ? dx : mbx;
  }
  double half = fmax(fmax(mtx * 1.9, mty * 1.9), fmax(mby * 1.2, mbx * 1.2));
  const double room = fmax(fmax(cx, iw - cx), fmax(cy, ih - cy));
  half = half < room ? half : room;
  if (half > (iw > ih ? iw : ih) / 2) {
    movenet_init_crop(ih, iw, o);
    return;
  }
  const double y0 = cy - half, x0 = cx - half, len = half * 2;
  o[0] = y0 / ih;
  o[1] = x0 / iw;
  o[2] = (y0 + len) / ih;
  o[3] = (x0 + len) / iw;
  o[4] = (y0 + len) / ih - y0 / ih;
  o[5] = (x0 + len) / iw - x0 / iw;
}

// ---- MultiPose head decode (specification: oracle/movenet.py:multipose_decode) -----------------------------
#define MV_T 256

// Centres: the frame's centre map sits in LDS; every thread keeps the local maxima (>= all in-range neighbours,
// above the floor) among its cells, and max_people rounds of a block-wide arg-max (higher score, then lower
// raster index) pick them in order.  (A one-pass rank-counting variant and a 1024-thread people kernel with
// shuffle reductions were both measured slower: 86 vs 30 us and 66 vs 52 us at batch 64.)
__global__ __launch_bounds__(MV_T) void movenet_centers_kernel(const float* __restrict__ center, int G, int maxp,
                                                                float min_center, int* __restrict__ c_idx,
                                                                float* __restrict__ c_score) {
  extern __shared__ float s_c[];  // the G x G centre map of this frame
  __shared__ float r_v[MV_T];
  __shared__ int r_i[MV_T];
  const int b = blockIdx.x, n = G * G;
  const float* c = center + (size_t)b * n;
  for (int i = threadIdx.x; i < n; i += MV_T) s_c[i] = c[i];
  __syncthreads();
  float mine[64];  // cells threadIdx.x + k * MV_T (G <= 128); -inf where the cell is no candidate
  int nk = 0;
  for (int i = threadIdx.x; i < n; i += MV_T, nk++) {
    const int y = i / G, x = i - y * G;
    const float v = s_c[i];
    bool ok = v > min_center;
    for (int dy = -1; dy <= 1 && ok; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int yy = y + dy, xx = x + dx;
        if ((dy || dx) && yy >= 0 && yy < G && xx >= 0 && xx < G && !(v >= s_c[yy * G + xx])) ok = false;
      }
    mine[nk] = ok ? v : -INFINITY;
  }
  for (int p = 0; p < maxp; p++) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = 0; k < nk; k++)
      if (mine[k] > bv) {  // ascending index within a thread: the first maximum stays
        bv = mine[k];
        bi = threadIdx.x + k * MV_T;
      }
    r_v[threadIdx.x] = bv;
    r_i[threadIdx.x] = bi;
    __syncthreads();
    for (int st = MV_T / 2; st > 0; st >>= 1) {
      if (threadIdx.x < st) {
        const float ov = r_v[threadIdx.x + st];
        const int oi = r_i[threadIdx.x + st];
        if (ov > r_v[threadIdx.x] || (ov == r_v[threadIdx.x] && oi < r_i[threadIdx.x])) {
          r_v[threadIdx.x] = ov;
          r_i[threadIdx.x] = oi;
        }
      }
      __syncthreads();
    }
    const float wv = r_v[0];
    const int wi = r_i[0];
    __syncthreads();
    if (threadIdx.x == 0) {
      c_idx[b * maxp + p] = wv > -INFINITY ? wi : -1;
      c_score[b * maxp + p] = wv > -INFINITY ? wv : 0.f;
    }
    if (wv > -INFINITY && (wi % MV_T) == (int)threadIdx.x) mine[wi / MV_T] = -INFINITY;
  }
}

// People: one block per (person slot, frame); every thread scans its grid cells for all 17 joints, then a
// block-wide (weight, cell) arg-max per joint through LDS.
__global__ __launch_bounds__(MV_T) void movenet_people_kernel(const float* __restrict__ heat,
                                                               const float* __restrict__ regr,
                                                               const float* __restrict__ offs,
                                                               const float* __restrict__ boxs, int G, int maxp,
                                                               const int* __restrict__ c_idx,
                                                               const float* __restrict__ c_score,
                                                               float* __restrict__ out) {
  __shared__ float r_v[17][MV_T];
  __shared__ int r_i[17][MV_T];
  const int p = blockIdx.x, b = blockIdx.y, n = G * G;
  float* o = out + ((size_t)b * maxp + p) * 56;
  const int ci = c_idx[b * maxp + p];
  if (ci < 0) {
    for (int i = threadIdx.x; i < 56; i += MV_T) o[i] = 0.f;
    return;
  }
  const int cy = ci / G, cx = ci - cy * G;
  const float* hm = heat + (size_t)b * n * 17;
  const float* rg = regr + ((size_t)b * n + ci) * 34;
  float ry[17], rx[17], bv[17];
  int bi[17];
#pragma unroll
  for (int j = 0; j < 17; j++) {
    ry[j] = (float)cy + rg[2 * j];
    rx[j] = (float)cx + rg[2 * j + 1];
    bv[j] = -INFINITY;
    bi[j] = 0x7fffffff;
  }
  for (int i = threadIdx.x; i < n; i += MV_T) {
    const int y = i / G, x = i - y * G;
    const float* h = hm + (size_t)i * 17;
#pragma unroll
    for (int j = 0; j < 17; j++) {
      const float dy = (float)y - ry[j], dx = (float)x - rx[j];
      const float w = h[j] / (sqrtf(dy * dy + dx * dx) + 1.8f);
      if (w > bv[j]) {  // ascending cells within a thread: the first maximum stays
        bv[j] = w;
        bi[j] = i;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 17; j++) {
    r_v[j][threadIdx.x] = bv[j];
    r_i[j][threadIdx.x] = bi[j];
  }
  __syncthreads();
  for (int st = MV_T / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int j = 0; j < 17; j++) {
        const float ov = r_v[j][threadIdx.x + st];
        const int oi = r_i[j][threadIdx.x + st];
        if (ov > r_v[j][threadIdx.x] || (ov == r_v[j][threadIdx.x] && oi < r_i[j][threadIdx.x])) {
          r_v[j][threadIdx.x] = ov;
          r_i[j][threadIdx.x] = oi;
        }
      }
    __syncthreads();
  }
  const float g = (float)G;
  if (threadIdx.x < 17) {
    const int j = threadIdx.x;
    const int k = r_i[j][0] == 0x7fffffff ? 0 : r_i[j][0];  // (all-NaN map: cell 0, as argmax would)
    const int ky = k / G, kx = k - ky * G;
    const float* of = offs + ((size_t)b * n + k) * 34;
    o[3 * j] = ((float)ky + of[2 * j]) / g;
    o[3 * j + 1] = ((float)kx + of[2 * j + 1]) / g;
    o[3 * j + 2] = hm[(size_t)k * 17 + j];
  } else if (threadIdx.x == 64) {
    const float* bs = boxs + ((size_t)b * n + ci) * 2;
    const float h2 = bs[0] * 0.5f, w2 = bs[1] * 0.5f;
    o[51] = ((float)cy - h2) / g;
    o[52] = ((float)cx - w2) / g;
    o[53] = ((float)cy + h2) / g;
    o[54] = ((float)cx + w2) / g;
    o[55] = c_score[b * maxp + p];
  }
}

int ph_launch_movenet_crop_resize(const uint8_t* frames, int B, int H, int W, const double* crop, int S, int32_t* out,
                                  hipStream_t stream) {
  const size_t total = (size_t)B * S * S;
  int blocks = (int)((total + 255) / 256);
  blocks = blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(movenet_crop_resize_kernel, dim3(blocks), dim3(256), 0, stream, frames, B, H, W, crop, S, out);
  PH_CHECK(hipGetLastError());
  return 0;
}

int ph_launch_movenet_update(float* kp, const double* crop_in, int B, int ih, int iw, double* crop_out,
                             hipStream_t stream) {
  hipLaunchKernelGGL(movenet_update_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, kp, crop_in, B, ih, iw, crop_out);
  PH_CHECK(hipGetLastError());
  return 0;
}

int ph_launch_movenet_multipose(const float* center, const float* heat, const float* regr, const float* offs,
                                const float* boxs, int B, int G, int maxp, float min_center, int* c_idx,
                                float* c_score, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(movenet_centers_kernel, dim3(B), dim3(MV_T), (size_t)G * G * sizeof(float), stream, center, G, maxp,
                     min_center, c_idx, c_score);
  PH_CHECK(hipGetLastError());
  hipLaunchKernelGGL(movenet_people_kernel, dim3(maxp, B), dim3(MV_T), 0, stream, heat, regr, offs, boxs, G, maxp, c_idx,
                     c_score, out);
  PH_CHECK(hipGetLastError());
  return 0;
}
