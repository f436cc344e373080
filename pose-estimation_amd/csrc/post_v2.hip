// Per-person post-processing on the device keypoints of the last V2 decode:
//  * torso bounding box       -- PyOpenPoseNativeBase.pose_bounding_box, openpose/native/python/skeleton.py:89-113
//  * depth-patch mean + pinhole back-projection -- get_3d_skeleton, skeleton.py:130-160
// so the demo path (demo/demo.py:854-869) needs no D2H round trip of the keypoints before the depth lookup.
// Arithmetic is IEEE in the order the reference writes it (fp32 min/max for the box, fp64 for the lift, no
// contraction), so the results are bit-identical to oracle/post_v2.py.
#include "common.h"

__global__ void torso_bbox_kernel(const float* __restrict__ kp, const int* __restrict__ n_people, int B, int maxp,
                                  int J, const int* __restrict__ ids, int n_ids, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * maxp) return;
  const int b = i / maxp, m = i - b * maxp;
  float* o = out + (size_t)i * 4;
  const int np_ = n_people[b] < maxp ? n_people[b] : maxp;
  if (m >= np_) {
    o[0] = o[1] = o[2] = o[3] = 0.f;
    return;
  }
  const float* p = kp + (size_t)i * J * 3;
  bool any = false;
  float u0 = 0.f, u1 = 0.f, v0 = 0.f, v1 = 0.f;
  for (int k = 0; k < n_ids; k++) {
    const float* q = p + ids[k] * 3;
    if (q[2] != 0.f) {
      if (!any) {
        u0 = u1 = q[0];
        v0 = v1 = q[1];
        any = true;
      } else {
        u0 = fminf(u0, q[0]);
        u1 = fmaxf(u1, q[0]);
        v0 = fminf(v0, q[1]);
        v1 = fmaxf(v1, q[1]);
      }
    }
  }
  if (any) {
    if (u0 == u1) u1 += 1.f;
    if (v0 == v1) v1 += 1.f;
  } else {
    u0 = v0 = 0.f;
    u1 = v1 = 1.f;
  }
  o[0] = u0;
  o[1] = v0;
  o[2] = u1;
  o[3] = v1;
}

__global__ void lift3d_kernel(const float* __restrict__ kp, const int* __restrict__ n_people, int B, int maxp, int J,
                              const uint16_t* __restrict__ depth, int Hd, int Wd, double fx, double fy, double cx,
                              double cy, double depth_scale, int po, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * maxp * J) return;
  const int pm = i / J;
  const int b = pm / maxp, m = pm - b * maxp;
  double* o = out + (size_t)i * 3;
  const int np_ = n_people[b] < maxp ? n_people[b] : maxp;
  if (m >= np_) {
    o[0] = o[1] = o[2] = 0.0;
    return;
  }
  const double x = (double)kp[(size_t)i * 3], y = (double)kp[(size_t)i * 3 + 1];
  // depth_img[max(0, int(y-po)):min(H, int(y+po)), max(0, int(x-po)):min(W, int(x+po))], int() truncates
  int ya = (int)(y - (double)po), yb = (int)(y + (double)po), xa = (int)(x - (double)po), xb = (int)(x + (double)po);
  ya = ya < 0 ? 0 : ya;
  xa = xa < 0 ? 0 : xa;
  yb = yb > Hd ? Hd : yb;
  xb = xb > Wd ? Wd : xb;
  const uint16_t* d = depth + (size_t)b * Hd * Wd;
  unsigned long long sum = 0;
  long long cnt = 0;
  for (int yy = ya; yy < yb; yy++)
    for (int xx = xa; xx < xb; xx++) {
      sum += d[(size_t)yy * Wd + xx];
      cnt++;
    }
  const double avg = cnt > 0 ? (double)sum / (double)cnt : __longlong_as_double(0x7ff8000000000000LL);  // nanmean([])
  const double x3 = (x - cx) / fx * avg;
  const double y3 = (y - cy) / fy * avg;
  o[0] = x3 * depth_scale;
  o[1] = y3 * depth_scale;
  o[2] = avg * depth_scale;
}

int ph_launch_torso_bbox(const float* kp, const int* n_people, int B, int maxp, int J, const int* ids, int n_ids,
                         float* out, hipStream_t stream) {
  const int n = B * maxp;
  hipLaunchKernelGGL(torso_bbox_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, kp, n_people, B, maxp, J, ids,
                     n_ids, out);
  PH_CHECK(hipGetLastError());
  return 0;
}

int ph_launch_lift3d(const float* kp, const int* n_people, int B, int maxp, int J, const uint16_t* depth, int Hd,
                     int Wd, double fx, double fy, double cx, double cy, double depth_scale, int po, double* out,
                     hipStream_t stream) {
  const int n = B * maxp * J;
  hipLaunchKernelGGL(lift3d_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, kp, n_people, B, maxp, J, depth, Hd,
                     Wd, fx, fy, cx, cy, depth_scale, po, out);
  PH_CHECK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// Tracker appearance feature: create_detections, tracking/utils_track.py:45-57, with
// resize_tensor, openpose/native/python/utils.py:101-110, evaluated per person straight from the 46x82
// network output: the crop of the exported heat maps (bicubic x8, --heatmaps_scale mapping; HWC
// [Ht][Wt][C]) under the torso box, resampled to fx x fy.  Only the fy*fx*C sampled values are ever
// interpolated; the 75 MB/frame upsampled tensor the reference crops from is never written.
//
//   s_w = Wt / img_w, s_h = Ht / img_h                      (float64, tracking/track.py:180-181)
//   ll = floor(box[0]*s_w)  tt = floor(box[1]*s_h)  rr = ceil(box[2]*s_w)  bb = ceil(box[3]*s_h)
//   crop = heatmaps[tt:bb, ll:rr]                           (Python slice rules: negative wraps, clipped)
//   out[i+1][j+1] = crop[1 + int(i / (fy/max(h-1,1)))][1 + int(j / (fx/max(w-1,1)))],  row 0 / column 0 zero
//
// One block per (person slot, frame); people of all frames are packed in frame order.
// status: 1 ok, -1 the crop has fewer than 2 rows or columns (the reference's loop raises IndexError).
// ---------------------------------------------------------------------------
#include "bicubic.h"

struct PhFeatWin {
  int c0[4], nc[4];
  int n;
};

__device__ __forceinline__ int py_slice_index(long long v, int n) {
  if (v < 0) v += n;
  return v < 0 ? 0 : (v > n ? n : (int)v);
}

__global__ __launch_bounds__(256) void pose_features_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                            const float* __restrict__ bbox,
                                                            const int* __restrict__ n_people, int maxp, PhFeatWin win,
                                                            int n_heat, int mode, int Ht, int Wt, double s_w,
                                                            double s_h, int fx, int fy, int capacity,
                                                            float* __restrict__ feat, int* __restrict__ status,
                                                            int* __restrict__ total) {
  const int b = blockIdx.y, m = blockIdx.x;
  const int np_ = n_people[b] < maxp ? n_people[b] : maxp;
  int row = 0;  // packed row of this person = people of the frames before + m
  for (int k = 0; k < b; k++) row += n_people[k] < maxp ? n_people[k] : maxp;
  if (m == 0 && b == (int)gridDim.y - 1 && threadIdx.x == 0) *total = row + np_;
  if (m >= np_) return;
  row += m;
  if (row >= capacity) return;
  const float* bx = bbox + ((size_t)b * maxp + m) * 4;
  const int ll = py_slice_index((long long)floor((double)bx[0] * s_w), Wt);
  const int tt = py_slice_index((long long)floor((double)bx[1] * s_h), Ht);
  const int rr = py_slice_index((long long)ceil((double)bx[2] * s_w), Wt);
  const int bb = py_slice_index((long long)ceil((double)bx[3] * s_h), Ht);
  const int w = rr > ll ? rr - ll : 0, h = bb > tt ? bb - tt : 0;
  int nch = 0;
  for (int k = 0; k < win.n; k++) nch += win.nc[k];
  float* out = feat + (size_t)row * fy * fx * nch;
  const size_t n_out = (size_t)fy * fx * nch;
  if (h < 2 || w < 2) {
    if (threadIdx.x == 0) status[row] = -1;
    for (size_t i = threadIdx.x; i < n_out; i += blockDim.x) out[i] = 0.f;
    return;
  }
  if (threadIdx.x == 0) status[row] = 1;
  const double x_scale = (double)fx / (double)(w - 1 > 1 ? w - 1 : 1);
  const double y_scale = (double)fy / (double)(h - 1 > 1 ? h - 1 : 1);
  const float* frame = net_out + (size_t)b * C * hs * ws;
  for (size_t i = threadIdx.x; i < n_out; i += blockDim.x) {
    const int c = (int)(i % nch);
    const int t = (int)(i / nch);
    const int j = t % fx, r = t / fx;
    if (r == 0 || j == 0) {
      out[i] = 0.f;
      continue;
    }
    const int sy = tt + 1 + (int)((double)(r - 1) / y_scale);
    const int sx = ll + 1 + (int)((double)(j - 1) / x_scale);
    int ch = c, k = 0;
    while (ch >= win.nc[k]) ch -= win.nc[k++];
    ch += win.c0[k];
    out[i] = heatmap_scale_value(bicubic_at(frame + (size_t)ch * hs * ws, hs, ws, Ht, Wt, sx, sy), ch < n_heat, mode);
  }
}

int ph_launch_pose_features(const float* net_out, int C, int hs, int ws, const float* bbox, const int* n_people, int B,
                            int maxp, const int* windows, int n_win, int n_heat, int mode, int Ht, int Wt, int img_w,
                            int img_h, int fx, int fy, int capacity, float* feat, int* status, int* total,
                            hipStream_t stream) {
  PhFeatWin win;
  win.n = n_win;
  for (int k = 0; k < 4; k++) {
    win.c0[k] = k < n_win ? windows[2 * k] : 0;
    win.nc[k] = k < n_win ? windows[2 * k + 1] : 0;
  }
  hipLaunchKernelGGL(pose_features_kernel, dim3(maxp, B), dim3(256), 0, stream, net_out, C, hs, ws, bbox, n_people, maxp,
                     win, n_heat, mode, Ht, Wt, (double)Wt / (double)img_w, (double)Ht / (double)img_h, fx, fy, capacity,
                     feat, status, total);
  PH_CHECK(hipGetLastError());
  return 0;
}
