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
