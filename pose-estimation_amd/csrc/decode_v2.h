// Internal declarations for the V2 decode launcher (csrc/decode_v2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct PhV2Dev {
  int n_parts, n_heat, n_pairs, max_peaks;
  int ear_pair0, ear_pair1, min_subset_cnt;
  int connector;       // 0: createPeopleVector (CPU build of OpenPose), 1: pafVectorIntoPeopleVector (CUDA build)
  float nms_threshold, inter_threshold, inter_min_above, min_subset_score;
  float close_score;   // float(double(nms_threshold) + 1e-6)
  double close_thr;    // sqrt(Wt*Ht) / 150: double in the CPU build, float in the CUDA build
  int pairs[128];
  int mapidx[128];
};

#define PH_V2_ROW_CAP 2048  // partial skeletons per frame the assembly kernel holds in LDS
#define PH_V2_MRG_LIST_CAP 16384  // live connections per frame kept as a list by connector 1 (beyond: matrix scans)
#define PH_V2_NMS_SPLIT 4  // horizontal stripes per heat map in the NMS kernel

struct PhV2Sizes {
  size_t nms_keys, nms_cnt, peaks, pair_scores, conn_idx, conn_score, n_conn, n_people, keypoints, scores, subsets;
  size_t n_accept, mrg_lists;
};

struct PhV2Buffers {
  int* nms_keys;
  int* nms_cnt;
  float* peaks;
  float* pair_scores;
  int* conn_idx;
  float* conn_score;
  int* n_conn;
  int* n_people;
  float* keypoints;
  float* scores;
  int* subsets;
  int* overflow;
  int* n_accept;    // connector 1: accepted candidate pairs per frame
  void* mrg_lists;  // connector 1: two connection lists per frame
};

int ph_v2_workspace_sizes(const PhV2Dev& m, int B, int max_people, PhV2Sizes* s);
int ph_launch_decode_v2(const float* net_out, int C, int hs, int ws, int Ht, int Wt, const PhV2Dev& m, int B,
                        float scale, int max_people, const PhV2Buffers& w, hipStream_t stream, hipEvent_t* ev);
int ph_launch_upsample_maps(const float* planes, int hs, int ws, int Ht, int Wt, int nc, int n_heat_rel, int mode,
                            float* out,
                            hipStream_t stream);
int ph_launch_argmax_maps(const float* maps, int n_maps, int n, float* vals, int* locs, hipStream_t stream);
