/* libposehip -- C ABI of the MI355X-native OpenPose bottom-up inference path.
 *
 * The reference (cheneeheng/pose-estimation) is Python and reaches this path
 * through two third-party engines, not through an FFI of its own:
 *   - cv2.dnn:     openpose/opencv/skeleton.py:50,66-75  (readNetFromCaffe,
 *                  blobFromImage, net.forward) + the in-tree decode
 *                  openpose/opencv/utils.py:9-173          ("V1" decode)
 *   - pyopenpose:  openpose/native/python/skeleton.py:424-447 (WrapperPython
 *                  configure/start/emplaceAndPop)          ("V2" decode)
 * Each entry point below names the reference call it stands in for.  The
 * Python classes that keep the reference's estimator interface
 * (pose-estimation_amd/posehip/api_opencv.py, api_native.py) bind these
 * symbols with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions: every function returns 0 on success and a negative value on
 * error (never throws); ph_last_error() returns the message of the calling
 * thread's last failure.  One ph_ctx == one device + one HIP stream; a ctx is
 * not thread-safe, distinct ctxs are independent.  "dev" pointers are device
 * addresses (e.g. torch.Tensor.data_ptr()), "host" pointers are host memory.
 */
#ifndef POSEHIP_H
#define POSEHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ph_ctx ph_ctx;

#define PH_MAX_DST 8
#define PH_OP_CONV 0
#define PH_OP_POOL 1
#define PH_ACT_NONE 0
#define PH_ACT_RELU 1
#define PH_ACT_PRELU 2
#define PH_PRE_CV2 0    /* blobFromImage(1/255, size, mean 0, no crop)           */
#define PH_PRE_NATIVE 1 /* OpenPose cvMatToOpInput: keep-aspect, x/256 - 0.5     */

/* A zero-bordered NHWC bf16 activation buffer [B][H+2*pad][W+2*pad][channels]. */
typedef struct {
  int channels; /* channel stride, multiple of 8 */
  int height, width;
  int pad;      /* border width (>= max pad of its consumer convs) */
} ph_buf_desc;

/* One fused operation of the execution plan (built by posehip/graph.py from the
 * Caffe prototxt the reference hands to readNetFromCaffe / OpenPose). */
typedef struct {
  int type;                 /* PH_OP_CONV | PH_OP_POOL */
  int in_buf, in_coff;      /* input buffer id, first channel read */
  int cin_read;             /* channels read (multiple of 8) */
  int kh, kw;               /* odd kernel, stride 1, pad = (k-1)/2 ("same") */
  int cout;
  int act;                  /* PH_ACT_* fused after bias */
  int n_dst;
  int dst_buf[PH_MAX_DST];
  int dst_coff[PH_MAX_DST]; /* multiples of 8 */
  int f32_coff;             /* >=0: also write fp32 into net_out channel f32_coff.. */
  int cin_caffe;            /* input channels of the Caffe weight blob */
  int slope_count;          /* 1 (channel_shared) or cout */
  int tile_n;               /* 0 = auto; else 32/64/96/128 */
  int pool;                 /* 1: the Caffe 2x2 stride-2 MAX Pooling that follows is fused into the
                               epilogue (3x3 convs, cin_read % 32 == 0, even H and W); the dst buffers
                               then have the pooled geometry */
  const float* weight;      /* host [cout][cin_caffe][kh][kw] */
  const float* bias;        /* host [cout] or NULL */
  const float* slope;       /* host PReLU slopes or NULL */
  const int32_t* cin_map;   /* host [cin_read]: caffe input channel or -1 (zero) */
} ph_op_desc;

/* Upstream-OpenPose decode constants ("V2"; posehip/tables.py PoseModelV2). */
typedef struct {
  int n_parts;        /* 25 (BODY_25) / 18 (COCO): NMS runs on channels [0,n_parts) */
  int n_heat;         /* n_parts + 1 (background): first PAF channel */
  int n_pairs;
  int max_peaks;      /* 127 */
  int ear_pair0, ear_pair1;
  int min_subset_cnt;
  float nms_threshold, inter_threshold, inter_min_above_threshold, min_subset_score;
  int pairs[64 * 2];  /* limb (partA, partB) */
  int mapidx[64 * 2]; /* PAF channel pair per limb, relative to n_heat */
  int connector;      /* person assembly of upstream bodyPartConnector: 1 = what a CUDA build of OpenPose runs
                         (pafPtrIntoVector + pafVectorIntoPeopleVector: one score-sorted walk over all candidate
                         pairs with assign / extend / merge-if-disjoint) -- the reference's Docker image builds
                         that; 0 = what a CPU-only build runs (per-limb sorted greedy + createPeopleVector).
                         Any other value is rejected (-1).  The struct has no spare fields: fill in EVERY member
                         (a zero-initialised struct selects 0, the CPU-build route; posehip/tables.py's models and
                         the ctypes binding default to 1) */
} ph_v2_model;

/* In-tree decode constants ("V1"; openpose/opencv/common.py MAPIDX :26-28 and
 * the body_parts / pose_pairs of get_body_parts_and_pose_pairs :31-102). */
typedef struct {
  int n_parts;             /* len(body_parts) - 1: heat maps decoded (background excluded) */
  int n_limbs;             /* len(MAPIDX) */
  int pair_parts[32 * 2];  /* body_parts[pose_pairs[k][0..1]] */
  int mapidx[32 * 2];      /* absolute PAF channels MAPIDX[k][0..1] */
} ph_v1_model;

/* ---- lifecycle ---------------------------------------------------------- */
/* stream: a hipStream_t to run on (e.g. torch's current stream) or NULL to
 * create a private one.  Stands in for cv2.dnn.readNetFromCaffe's Net object /
 * op.WrapperPython() (opencv/skeleton.py:50; native/python/skeleton.py:424). */
int ph_create(ph_ctx** out, int device, void* stream);
void ph_destroy(ph_ctx* ctx);
int ph_sync(ph_ctx* ctx);
const char* ph_last_error(void);
/* 1 when the library was built for the device it runs on (gfx950). */
int ph_device_ok(int device);
/* Page-locked host memory for result buffers that are filled again and again (hipHostMalloc / hipHostFree): device to
 * host copies go straight to it instead of through the runtime's staging buffer.  No counterpart in the reference
 * (cv2 / pyopenpose hand back fresh NumPy arrays, opencv/skeleton.py:154); used by Engine.decode_v1.  NULL on failure. */
void* ph_host_alloc(int64_t bytes);
int ph_host_free(void* ptr);

/* ---- network (replaces net.setInput/net.forward, opencv/skeleton.py:74-75;
 *      the Caffe forward inside emplaceAndPop, native/python/skeleton.py:447) */
int ph_net_build(ph_ctx* ctx, int batch, const ph_buf_desc* bufs, int n_bufs, const ph_op_desc* ops, int n_ops,
                 int out_channels, int out_h, int out_w);
/* frames: u8 [B][H0][W0][3] BGR, host (on_device=0) or device memory.
 * Replaces cv2.dnn.blobFromImage (opencv/skeleton.py:66-73) / cvMatToOpInput. */
int ph_preprocess(ph_ctx* ctx, const uint8_t* frames, int on_device, int B, int H0, int W0, int mode);
int ph_forward(ph_ctx* ctx, int B);
/* forward with a HIP event recorded on the ctx stream around every op.  ms (n_ops
 * slots, device time per op) may be NULL: then nothing is waited for and
 * ph_forward_times() reads the last timed forward's durations later. */
int ph_forward_timed(ph_ctx* ctx, int B, float* ms);
int ph_forward_times(ph_ctx* ctx, float* ms);
/* algorithmic FLOPs of op i for one frame (2*K*K*Cin*Cout*H*W, unpadded) */
double ph_op_flops(ph_ctx* ctx, int op);
/* name of the kernel instantiation op i launches (as rocprofv3 --kernel-trace prints it, without
 * the argument list), e.g. "conv3x3_halo_kernel<4, false>"; measurement only, no reference counterpart */
int ph_op_kernel(ph_ctx* ctx, int op, char* name, int cap);
/* net output, fp32 NCHW [B][C][Hs][Ws] (what predict() returns, opencv/skeleton.py:75) */
int ph_get_net_out(ph_ctx* ctx, int B, float* host_out);
float* ph_net_out_dev(ph_ctx* ctx);
/* inject a net output (predict_from_hm, native/python/skeleton.py:469-481) */
int ph_set_net_out(ph_ctx* ctx, const float* src, int on_device, int B, int C, int Hs, int Ws);
/* bicubic-upsampled maps of one frame, channels [c0, c0+nc), to host f32
 * [nc][Ht][Wt]: datum.poseHeatMaps export (native/python/skeleton.py:64-76).
 * n_heat: part + background channels of the net output (the rest are PAFs); scale_mode = OpenPose's
 * --heatmaps_scale (params "heatmaps_scale", native/python/skeleton.py:531): 0 [-1,1], 1 [0,1],
 * 2 integer-rounded [0,255], 3 as is. */
int ph_upsample_maps(ph_ctx* ctx, int frame, int c0, int nc, int Ht, int Wt, int n_heat, int scale_mode,
                     float* host_out);
/* per-map global max and first raster location of it for all B*C maps of the
 * net output: cv2.minMaxLoc per heat map in postprocess_single
 * (opencv/skeleton.py:88-99).  vals f32 [B*C], locs i32 [B*C] (= y*Ws + x). */
int ph_argmax_maps(ph_ctx* ctx, int B, float* vals_host, int32_t* locs_host);
/* debug/test: copy an activation buffer (bf16 bits) to host */
int ph_get_buffer(ph_ctx* ctx, int buf, uint16_t* host_out, int64_t n_elems);

/* ---- V2 decode (replaces OpenPose resizeAndMerge + nms + bodyPartConnector
 *      inside emplaceAndPop, native/python/skeleton.py:447) ------------------
 * Decodes the ctx's net output for B frames at target resolution Ht x Wt
 * (bicubic upsample never materialised).  Host outputs, caller-allocated:
 *   n_people   int32 [B]            people stored per frame: those that pass the thresholds, at most max_people
 *                                   (upstream stops at POSE_MAX_PEOPLE the same way)
 *   keypoints  f32   [B][max_people][n_parts][3]
 *   scores     f32   [B][max_people]
 *   peaks      f32   [B][n_parts][max_peaks+1][3]   (optional, may be NULL)
 *   subsets    int32 [B][max_people][n_parts+1]     (optional, may be NULL)
 * scale: keypoint x,y multiplier (net-input -> image pixels). */
int ph_decode_v2(ph_ctx* ctx, const ph_v2_model* model, int B, int Ht, int Wt, float scale, int max_people,
                 int32_t* n_people, float* keypoints, float* scores, float* peaks, int32_t* subsets);
/* same, but leaves results in device memory for a later RCCL gather; the
 * returned pointers stay valid until the next decode on this ctx */
int ph_decode_v2_dev(ph_ctx* ctx, const ph_v2_model* model, int B, int Ht, int Wt, float scale, int max_people,
                     int32_t** n_people_dev, float** keypoints_dev, float** scores_dev);
/* asynchronous copy (on the ctx stream, no wait) of the last ph_decode_v2_dev
 * results into caller buffers of the shapes above: pinned host memory
 * (dst_on_device=0) or device memory (dst_on_device=1, e.g. the send buffer of
 * the RCCL keypoint gather).  Any pointer may be NULL.  The decode's overflow flag is copied along (to pinned
 * memory of the ctx): call ph_decode_v2_status() after the stream has been synchronised. */
int ph_decode_v2_results(ph_ctx* ctx, int32_t* n_people, float* keypoints, float* scores, int dst_on_device);
/* Overflow report of the asynchronous path: waits for the ctx stream, then returns -2 (and sets the error) when the
 * decode whose results the last ph_decode_v2_results call copied had more than 2048 partial skeletons in a frame
 * (people were dropped), 0 otherwise.  ph_decode_v2 reports the same condition through its own return value. */
int ph_decode_v2_status(ph_ctx* ctx);

/* The one entry point that allocates nothing.  It has NO counterpart in the reference: it exists for callers that
 * cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same kernels and results as ph_decode_v2_dev, but the net output is the caller's device buffer
 * (f32 [B][C][Hs][Ws]), the results go to the caller's device buffers (shapes as above) and the workspace (peak
 * tables, PAF score matrices, connection lists) lives in a 20 MiB static device array of the library, one per GPU,
 * held until the process ends.  Limit: B <= PH_NOALLOC_MAX_FRAMES and the workspace of the B frames must fit the
 * array (it does for BODY_25 / COCO with 127 peaks per part); larger calls return -3 and launch nothing.  Asynchronous on the ctx stream.
 * overflow_dev (device int32, may be NULL) receives 1 when a frame had more than 2048 partial skeletons.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two ctxs, two streams) race. */
#define PH_NOALLOC_MAX_FRAMES 8
int ph_decode_v2_noalloc(ph_ctx* ctx, const ph_v2_model* model, const float* net_out_dev, int B, int C, int Hs, int Ws,
                         int Ht, int Wt, float scale, int max_people, int32_t* n_people_dev, float* keypoints_dev,
                         float* scores_dev, int32_t* overflow_dev);

/* Per-person post-processing of the LAST ph_decode_v2(_dev) results, on the device keypoints (no D2H first):
 *  bbox  [B][max_people][4] f32 (u_min, v_min, u_max, v_max) over the `torso_ids` joints with score != 0
 *        (degenerate side -> +1, none -> 0,0,1,1): PyOpenPoseNativeBase.pose_bounding_box,
 *        openpose/native/python/skeleton.py:89-113; NULL = skip.
 *  kp3d  [B][max_people][n_parts][3] f64: mean of the u16 depth patch [y-po, y+po) x [x-po, x+po) and pinhole
 *        back-projection, scaled by depth_scale: get_3d_skeleton, skeleton.py:130-160; NULL = skip.
 *        depth: u16 [B][Hd][Wd], host or device.  Rows past n_people[b] are zero.  Synchronises the stream. */
int ph_pose_post(ph_ctx* ctx, int B, int max_people, int n_parts, const int32_t* torso_ids, int n_torso, float* bbox,
                 const uint16_t* depth, int depth_on_device, int Hd, int Wd, double fx, double fy, double cx, double cy,
                 double depth_scale, int patch_offset, double* kp3d);
/* Tracker appearance feature of every person of the LAST ph_decode_v2(_dev), from the resident net output:
 * the crop of the exported heat maps (channel windows `windows` = n_win x (c0, nc) of the net output,
 * concatenated; bicubic to Ht x Wt; scale_mode as above) under the person's torso box, resampled to fx x fy
 * by the reference's resize_tensor -- create_detections, tracking/utils_track.py:45-57, with
 * openpose/native/python/utils.py:101-110 and tracking/track.py:180-183 (s_w = Wt/img_w, s_h = Ht/img_h).
 * The upsampled maps are never materialised.  People of all frames are packed in frame order:
 *   feat   f32 [n_rows][fy][fx][C]   (C = sum of the window widths; row 0 / column 0 zero as in the reference)
 *   status i32 [n_rows]  1 ok, -1 crop with fewer than 2 rows/columns (the reference raises IndexError)
 *   n_rows = people of all B frames; returns -3 (nothing copied) if it exceeds `capacity`.
 * Synchronises the stream. */
int ph_pose_features(ph_ctx* ctx, int B, int max_people, int n_parts, const int32_t* torso_ids, int n_torso,
                     const int32_t* windows, int n_win, int n_heat, int scale_mode, int Ht, int Wt, int img_w, int img_h,
                     int fx, int fy, int capacity, float* feat, int32_t* status, int32_t* n_rows);

/* ---- MoveNet steps around the (remote, TF-Hub) model: SURVEY 8f-4, BASELINE configs[3] -------------------------
 * on_device = 1: every pointer is device memory (no allocation, no copy); 0: host memory.  All synchronise.
 *
 * tf.image.crop_and_resize (bilinear, extrapolation 0) of one box per frame + cast to int32:
 * crop_and_resize / run_inference, movenet/utils/image_sequence_crop.py:153-171.
 *   frames u8 [B][H][W][3]; crop f64 [B][6] = y_min, x_min, y_max, x_max, height, width (normalised, the
 *   reference's crop_region dict); out i32 [B][size][size][3] */
int ph_movenet_crop_resize(ph_ctx* ctx, const uint8_t* frames, int B, int H, int W, const double* crop, int size,
                           int32_t* out, int on_device);
/* per stream: model output (y, x, score) from crop to image coordinates, in place (image_sequence_crop.py:178-186),
 * then the next frame's crop region, determine_crop_region :94-150 (init_crop_region :15-40, torso_visible :43-56,
 * determine_torso_and_body_range :59-91).  keypoints f32 [B][17][3]; crop_in / crop_out f64 [B][6]. */
int ph_movenet_update(ph_ctx* ctx, float* keypoints, const double* crop_in, int B, int image_height, int image_width,
                      double* crop_out, int on_device);
/* MultiPose head decode (BASELINE configs[3]).  No counterpart in the reference tree (it only calls SinglePose
 * graphs, whose decode is inside the TF-Hub model): the algorithm is specified in oracle/movenet.py
 * (multipose_decode), parity unpinned.  Heads f32 on a G x G grid: center [B][G][G], kpt_heat [B][G][G][17],
 * kpt_regr / kpt_offs [B][G][G][34], box_size [B][G][G][2]; out f32 [B][max_people][56]. */
int ph_movenet_multipose_decode(ph_ctx* ctx, const float* center, const float* kpt_heat, const float* kpt_regr,
                                const float* kpt_offs, const float* box_size, int B, int G, int max_people,
                                float min_center, float* out, int on_device);

/* test/debug: per-limb connections of the last decode.  n_conn i32 [B][n_pairs],
 * conn_idx i32 [B][n_pairs][max_peaks][2] (upstream peak indices part*3*(max_peaks+1)
 * + i*3 + 2), conn_score f32 [B][n_pairs][max_peaks]; any pointer may be NULL. */
int ph_decode_v2_connections(ph_ctx* ctx, int32_t* n_conn, int32_t* conn_idx, float* conn_score);
/* device time (ms) of the decode kernels of the last ph_decode_v2* call:
 * [0] upsample+NMS, [1] PAF scoring+matching, [2] assembly */
int ph_decode_v2_times(ph_ctx* ctx, float* ms3);

/* ---- V1 decode (replaces PyOpenPoseOpenCV.postprocess_multi and its helpers
 *      getKeypoints / getValidPairs / getPersonwiseKeypoints,
 *      opencv/skeleton.py:102-154, opencv/utils.py:9-173) ----------------------
 * Decodes the ctx's net output for B frames at ORIGINAL-IMAGE resolution
 * W0 x H0 (bilinear-resized maps never materialised).  Host outputs:
 *   n_kps     int32 [B][n_parts]                 keypoints per part = contours of findContours(RETR_TREE),
 *                                                outer AND hole borders, in its list order; each keypoint =
 *                                                first maximum of the smoothed map under fillConvexPoly(contour)
 *   kps       f32   [B][n_parts][kp_cap][3]      (x, y, score); ids run over parts in order
 *   n_pairs   int32 [B][n_limbs]                 rows per limb, -1 = "invalid pair"
 *   pairs     f64   [B][n_limbs][kp_cap][3]      (idA, idB, mean PAF score)
 *   n_persons int32 [B]
 *   persons   f64   [B][person_cap][n_parts+1]   keypoint ids (-1 = missing) + total score
 * Returns -2 when a capacity (kp_cap contours per part, person_cap people, 8192
 * mask runs per part, or the part's pool of contour vertices: 65536 for the
 * one-group walk, as before, beside 196608 entries for the wave walk's tables)
 * is exceeded. */
int ph_decode_v1(ph_ctx* ctx, const ph_v1_model* model, int B, int W0, int H0, float conf_thres, int kp_cap,
                 int person_cap, int32_t* n_kps, float* kps, int32_t* n_pairs, double* pairs, int32_t* n_persons,
                 double* persons);
/* device time (ms) of the last ph_decode_v1: [0] resize+blur+runs, [1] blobs, [2] pairs+assembly */
int ph_decode_v1_times(ph_ctx* ctx, float* ms3);
/* of the last ph_decode_v1 (no reference counterpart; for tests and profiles): how many blobs with several runs in
 * a row -- the ones that need the literal findContours / fillConvexPoly of utils.py:14-22 -- were walked by a whole
 * wave (all border loops at once) and how many by the one-group walk (POSEHIP_V1_FAST=0, or a guard of the wave
 * walk tripped). */
int ph_decode_v1_walks(ph_ctx* ctx, int32_t* wave, int32_t* group);

#ifdef __cplusplus
}
#endif
#endif
