// Internal declarations for the V1 decode launcher (csrc/decode_v1.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define V1_MAX_PARTS 32
#define V1_MAX_LIMBS 32
// 4-byte entries of a (frame, part)'s pool: the contour vertices of the literal blob walk (8 per run, as before the
// fast walk existed: the one-group walk's share) + the tables, scratch vertices and fill tables of the fast walk
#define V1_VERT_POOL(run_cap) (32 * (run_cap))

struct V1Run {
  int y, xs, xe;  // row, first and last column of a run of mask pixels
  int bx;         // column of the run's first maximum of the smoothed map
  float best;     // that maximum
};

struct V1Model {
  int n_parts;                       // len(body_parts) - 1 (background excluded)
  int n_limbs;                       // len(MAPIDX)
  int pair_parts[2 * V1_MAX_LIMBS];  // body-part ids of pose_pairs[k]
  int mapidx[2 * V1_MAX_LIMBS];      // absolute PAF channels MAPIDX[k]
};

struct V1Caps {
  int run_cap, kp_cap, person_cap;
};

struct V1Sizes {
  size_t n_runs, runs, ws_verts, kps, n_kps, pairs, n_pairs, persons, n_persons;
};

struct V1Buffers {
  int* n_runs;
  V1Run* runs;
  short2* ws_verts;
  float* kps;
  int* n_kps;
  double* pairs;
  int* n_pairs;
  double* persons;
  int* n_persons;
  int* overflow;
};

int ph_v1_sizes(const V1Model& m, int B, int W, int H, const V1Caps& c, V1Sizes* s);
int ph_launch_decode_v1(const float* net_out, int C, int hs, int ws, int W, int H, float thr, const V1Model& m,
                        int B, const V1Caps& c, const V1Buffers& w, hipStream_t stream, hipEvent_t* ev);
