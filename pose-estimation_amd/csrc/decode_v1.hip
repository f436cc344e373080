// "V1" decode for gfx950: the reference's in-tree multi-person decode
//   PyOpenPoseOpenCV.postprocess_multi   openpose/opencv/skeleton.py:102-154
//   getKeypoints                         openpose/opencv/utils.py:9-23
//   getValidPairs                        openpose/opencv/utils.py:27-131
//   getPersonwiseKeypoints               openpose/opencv/utils.py:136-173
// run at ORIGINAL-IMAGE resolution (up to 1920x1080) without ever writing the
// bilinearly resized maps (415 MB/frame at 1080p if materialised):
//   K-A  per 16x256 tile: bilinear resize -> 3x3 Gaussian -> threshold -> row
//        runs of the mask with their first-raster maxima (tiles whose source
//        cells are all below threshold are skipped);
//   K-B  per (frame, part): runs sorted by (row, column) in LDS, connected
//        components of touching runs (8-connectivity) = the blobs
//        cv2.findContours traces; a blob with one run per row gives its
//        arg-max, every other blob is walked literally (border following,
//        hole borders, fillConvexPoly): by a wave that finds all its border
//        loops at once (v1_fast_blob), else by a group of 8 lanes that
//        follows one border pixel after the other; RETR_TREE list order;
//   K-C  per (frame, limb): 10-sample nearest-pixel PAF line integral in fp64,
//        per-A arg-max, ordered compaction;
//   K-D  per frame: person assembly in LDS.
// Bit-exact against oracle/decode_v1.py (same IEEE operations, same order;
// built with -ffp-contract=off).
#include "common.h"
#include "decode_v1.h"

#define V1_TILE_H 16
#define V1_TILE_W 256
#define V1_RUN_THREADS 1024

__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  if (p < 0) p = -p;
  if (p >= n) p = 2 * (n - 1) - p;
  return p;
}

// cv2.resize INTER_LINEAR coordinate (x axis: weight reset at the clamps)
struct LinX {
  int s0, s1;
  float a0, a1;
};
__device__ __forceinline__ LinX lin_x(int d, int n_dst, int n_src) {
  float fx = (float)(((double)d + 0.5) * ((double)n_src / (double)n_dst) - 0.5);
  int sx = (int)floorf(fx);
  fx = fx - (float)sx;
  if (sx < 0) {
    fx = 0.f;
    sx = 0;
  }
  if (sx >= n_src - 1) {
    fx = 0.f;
    sx = n_src - 1;
  }
  LinX r;
  r.s0 = sx;
  r.s1 = sx + 1 > n_src - 1 ? n_src - 1 : sx + 1;
  r.a0 = 1.f - fx;
  r.a1 = fx;
  return r;
}
// y axis: rows clipped, weights kept
__device__ __forceinline__ LinX lin_y(int d, int n_dst, int n_src) {
  float fy = (float)(((double)d + 0.5) * ((double)n_src / (double)n_dst) - 0.5);
  const int sy = (int)floorf(fy);
  fy = fy - (float)sy;
  LinX r;
  r.s0 = sy < 0 ? 0 : (sy > n_src - 1 ? n_src - 1 : sy);
  r.s1 = sy + 1 < 0 ? 0 : (sy + 1 > n_src - 1 ? n_src - 1 : sy + 1);
  r.a0 = 1.f - fy;
  r.a1 = fy;
  return r;
}
__device__ __forceinline__ float resize_at(const float* src, int hs, int ws, int W, int H, int x, int y) {
  const LinX cx = lin_x(x, W, ws), cy = lin_y(y, H, hs);
  const float r0 = src[cy.s0 * ws + cx.s0] * cx.a0 + src[cy.s0 * ws + cx.s1] * cx.a1;
  const float r1 = src[cy.s1 * ws + cx.s0] * cx.a0 + src[cy.s1 * ws + cx.s1] * cx.a1;
  return r0 * cy.a0 + r1 * cy.a1;
}

__device__ __forceinline__ float resize_tab(const float* src, int ws, const LinX& cx, const LinX& cy) {
  const float r0 = src[cy.s0 * ws + cx.s0] * cx.a0 + src[cy.s0 * ws + cx.s1] * cx.a1;
  const float r1 = src[cy.s1 * ws + cx.s0] * cx.a0 + src[cy.s1 * ws + cx.s1] * cx.a1;
  return r0 * cy.a0 + r1 * cy.a1;
}

// ---------------------------------------------------------------------------
// K-A: mask runs.  grid (tiles_x * tiles_y, n_parts, B), block 256
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(V1_RUN_THREADS) void v1_runs_kernel(const float* __restrict__ net_out, int C, int hs, int ws, int W,
                                                      int H, float thr, int n_parts, int run_cap,
                                                      int* __restrict__ n_runs, V1Run* __restrict__ runs,
                                                      int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tiles_x = (W + V1_TILE_W - 1) / V1_TILE_W;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int part = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int x0 = tx * V1_TILE_W, y0 = ty * V1_TILE_H;
  float* s_src = smem;                         // hs*ws
  float* s_heat = s_src + hs * ws;             // (TILE_H+2) * (TILE_W+2)
  float* s_t = s_heat + (V1_TILE_H + 2) * (V1_TILE_W + 2);  // (TILE_H+2) * TILE_W
  float* s_sm = s_t + (V1_TILE_H + 2) * V1_TILE_W;          // TILE_H * TILE_W
  __shared__ float s_max[V1_RUN_THREADS / 64];
  __shared__ LinX s_lx[V1_TILE_W + 2];   // per-tile coordinate tables: the fp64 coordinate
  __shared__ LinX s_ly[V1_TILE_H + 2];   // arithmetic runs once per column/row, not per pixel

  const float* plane = net_out + ((size_t)b * C + part) * hs * ws;
  // source cells that can reach this tile (1-px blur halo, both bilinear taps)
  const int xa = x0 > 0 ? x0 - 1 : 0, xb = x0 + V1_TILE_W < W ? x0 + V1_TILE_W : W - 1;
  const int ya = y0 > 0 ? y0 - 1 : 0, yb = y0 + V1_TILE_H < H ? y0 + V1_TILE_H : H - 1;
  const int sxa = lin_x(xa, W, ws).s0, sxb = lin_x(xb, W, ws).s1;
  const int sya = lin_y(ya, H, hs).s0, syb = lin_y(yb, H, hs).s1;
  // only the source cells this tile can reach are staged (a 16x256 tile of a 1080p map touches ~4x14 of the
  // 46x82 cells): s_src is the sub-rectangle [sya..syb] x [sxa..sxb], row stride sw
  const int sw = sxb - sxa + 1, sh = syb - sya + 1;
  float lm = -INFINITY;
  for (int i = tid; i < sw * sh; i += V1_RUN_THREADS) {
    const int yy = i / sw, xx = i - yy * sw;
    const float v = plane[(sya + yy) * ws + sxa + xx];
    s_src[i] = v;
    lm = fmaxf(lm, v);
  }
  for (int o = 32; o > 0; o >>= 1) lm = fmaxf(lm, __shfl_xor(lm, o));
  if ((tid & 63) == 0) s_max[tid >> 6] = lm;
  __syncthreads();
  float tile_max = s_max[0];
  for (int i = 1; i < V1_RUN_THREADS / 64; i++) tile_max = fmaxf(tile_max, s_max[i]);
  // resize + blur are convex combinations up to a few ulps: nothing here can exceed thr
  if (thr > 0.f && tile_max < thr * 0.999f) return;

  // 1. bilinear-resized heat map, tile + 1-px halo, reflect-101 at the image border
  constexpr int HW_ = V1_TILE_W + 2;
  for (int c = tid; c < HW_; c += V1_RUN_THREADS)
    if (x0 - 1 + c <= W) {
      LinX t = lin_x(reflect101(x0 - 1 + c, W), W, ws);
      t.s0 -= sxa;  // indices relative to the staged sub-rectangle
      t.s1 -= sxa;
      s_lx[c] = t;
    }
  if (tid < V1_TILE_H + 2 && y0 - 1 + tid <= H) {
    LinX t = lin_y(reflect101(y0 - 1 + tid, H), H, hs);
    t.s0 -= sya;
    t.s1 -= sya;
    s_ly[tid] = t;
  }
  __syncthreads();
  for (int i = tid; i < (V1_TILE_H + 2) * HW_; i += V1_RUN_THREADS) {
    const int r = i / HW_, c = i - r * HW_;
    s_heat[i] = (y0 - 1 + r <= H && x0 - 1 + c <= W) ? resize_tab(s_src, sw, s_lx[c], s_ly[r]) : 0.f;
  }
  __syncthreads();
  // 2. row pass  c*1/2 + (l + r)*1/4
  for (int i = tid; i < (V1_TILE_H + 2) * V1_TILE_W; i += V1_RUN_THREADS) {
    const int r = i / V1_TILE_W, c = i - r * V1_TILE_W;
    const float* h = s_heat + r * HW_ + c + 1;
    s_t[i] = h[0] * 0.5f + (h[-1] + h[1]) * 0.25f;
  }
  __syncthreads();
  // 3. column pass
  for (int i = tid; i < V1_TILE_H * V1_TILE_W; i += V1_RUN_THREADS) {
    const int r = i / V1_TILE_W, c = i - r * V1_TILE_W;
    const float* t = s_t + (r + 1) * V1_TILE_W + c;
    const bool inside = y0 + r < H && x0 + c < W;
    s_sm[i] = inside ? t[0] * 0.5f + (t[-V1_TILE_W] + t[V1_TILE_W]) * 0.25f : -INFINITY;
  }
  __syncthreads();
  // 4. runs, one wave per tile row (16 rows = 16 waves): lane l holds pixels l, l+64, l+128, l+192 of the row, so
  //    a ballot per 64-pixel word gives the row's mask as four 64-bit words.  The best pixel of every run (max
  //    value, first position on ties -- what a left-to-right walk with `>` finds) comes from a segmented max-scan
  //    (6 shuffle steps per word, carry across words); the lane on a run's LAST pixel emits it, its first pixel
  //    found by bit scans.  Slots are reserved per BLOCK (LDS counter, then one global atomicAdd).
  static_assert(V1_TILE_W == 256 && V1_TILE_H * 64 == V1_RUN_THREADS, "one wave per 256-pixel tile row");
  __shared__ int s_cnt, s_base;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int* cnt = n_runs + (size_t)b * n_parts + part;
  V1Run* out = runs + ((size_t)b * n_parts + part) * run_cap;
  const int r = tid >> 6, l = tid & 63;
  const float* row = s_sm + r * V1_TILE_W;
  float v[4], sv[4];
  int sp[4];
  unsigned long long mw[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v[k] = row[64 * k + l];
    mw[k] = __ballot(v[k] > thr);
  }
  float cv = 0.f;   // scan value / position carried out of the previous word (its lane 63)
  int cp = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool on = (mw[k] >> l) & 1;
    const bool prev_on = l > 0 ? ((mw[k] >> (l - 1)) & 1) : (k > 0 && (mw[k - 1] >> 63) & 1);
    float x = v[k];
    int px = 64 * k + l;
    bool f = !on || !prev_on || l == 0;   // segment head inside this word (word-crossing runs: carry below)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float ox = __shfl_up(x, d);
      const int op = __shfl_up(px, d);
      const int of = __shfl_up((int)f, d);
      if (l >= d && !f) {
        if (ox >= x) {  // the earlier pixel wins ties
          x = ox;
          px = op;
        }
        f = of;
      }
    }
    // lanes of the run that continues from the previous word: merge its carried best
    if (k > 0 && on && ((mw[k - 1] >> 63) & 1)) {
      const unsigned long long z = ~mw[k];
      const int first_zero = z ? __ffsll((long long)z) - 1 : 64;
      if (l < first_zero && cv >= x) {
        x = cv;
        px = cp;
      }
    }
    sv[k] = x;
    sp[k] = px;
    cv = __shfl(x, 63);
    cp = __shfl(px, 63);
  }
  V1Run mine[4];
  int myslot[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    myslot[k] = -1;
    const bool on = (mw[k] >> l) & 1;
    const bool next_on = l < 63 ? ((mw[k] >> (l + 1)) & 1) : (k < 3 && (mw[k + 1] & 1));
    if (on && !next_on) {  // last pixel of a run: its first pixel = one past the nearest zero to the left
      int kk = k;
      unsigned long long below = l ? (~mw[k]) & ((1ull << l) - 1ull) : 0ull;
      int start;
      while (true) {
        if (below) {
          start = 64 * kk + (63 - __clzll((long long)below)) + 1;
          break;
        }
        if (kk == 0) {
          start = 0;
          break;
        }
        kk--;
        below = ~mw[kk];
      }
      mine[k].y = y0 + r;
      mine[k].xs = x0 + start;
      mine[k].xe = x0 + 64 * k + l;
      mine[k].bx = x0 + sp[k];
      mine[k].best = sv[k];
      myslot[k] = atomicAdd(&s_cnt, 1);
    }
  }
  __syncthreads();
  if (tid == 0) s_base = s_cnt ? atomicAdd(cnt, s_cnt) : 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (myslot[k] >= 0) {
      const int slot = s_base + myslot[k];
      if (slot < run_cap) out[slot] = mine[k];
      else *overflow = 1;
    }
}

// ---------------------------------------------------------------------------
// K-A, column-streaming form (round 3; the tiled form above stays behind POSEHIP_V1_RUNS=0 for A/B runs).
// The tiled kernel evaluates every pixel of an active tile from scratch (4 source reads + 2 table reads for the
// resize, then two 3-tap passes through LDS: ~130 VALU operations and ~20 LDS accesses per pixel, 83 % VALU /
// 54 % LDS busy, 5.4 ms for 32 COCO frames at 1080p).  At these upscaling ratios (x8 ... x23) almost all of that
// is redundant: the two horizontally interpolated source rows a pixel mixes are the same for every output row
// between two source rows.  Here a WAVE owns a strip of 256 columns (lane l: columns l, l+64, l+128, l+192 -- the
// layout the run scan wants) and walks down 16 rows (short walks: a wave's rows are one dependent chain, and the chip
// hides that latency only with many waves):
//   * per lane and column the horizontal interpolation of the two current source rows lives in registers and is
//     redone only when the source row pair changes;
//   * a row costs, per pixel: the vertical mix (2 mul + 1 add), the row pass (neighbouring columns = neighbouring
//     lanes by DPP wave shifts, the word seams by one readlane), the column pass over three rows kept in registers,
//     one compare: ~16 operations, no LDS;
//   * image borders need no special case: the halo columns / rows are evaluated at their reflect-101 coordinates;
//   * rows whose source rows stay below the threshold (over the strip's source columns) are skipped;
//   * runs are found by the tiled kernel's ballot + segmented max-scan, only in rows that have mask bits, collected
//     in a per-wave LDS list and written with one global atomicAdd per flush.
// Same IEEE operations in the same order as the tiled form (and as oracle/decode_v1.py): the run set is identical.
// grid (strips_x * ceil(H / 64), n_parts, B), block 256 = 4 waves = 4 consecutive 16-row chunks of one strip.
// ---------------------------------------------------------------------------
#define V1C_ROWS 16
#define V1C_WAVES 4
#define V1C_LIST 160
// first position >= pos of the 256-bit row mask m[0..3] whose bit is set (clear = false) / clear (clear = true); 256 if none
__device__ __forceinline__ int v1c_next_bit(const unsigned long long* m, int pos, bool clear) {
  for (int k = pos >> 6; k < 4; k++) {
    unsigned long long w = clear ? ~m[k] : m[k];
    if (k == (pos >> 6)) w &= ~0ull << (pos & 63);
    if (w) return 64 * k + __ffsll((long long)w) - 1;
  }
  return 256;
}
// maximum over each row of 16 lanes, in every lane of the row (quad swaps, then the two mirrors)
__device__ __forceinline__ unsigned v1c_max16(unsigned x) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);
  x = o > x ? o : x;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);
  x = o > x ? o : x;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true);
  x = o > x ? o : x;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true);
  x = o > x ? o : x;
  return x;
}
__global__ __launch_bounds__(64 * V1C_WAVES) void v1_runs_cols_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                                       int W, int H, float thr, int n_parts, int run_cap,
                                                                       int src_cap, int* __restrict__ n_runs,
                                                                       V1Run* __restrict__ runs, int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) float s_src[];   // staged source sub-rectangle (src_cap floats)
  constexpr int NR = V1C_ROWS * V1C_WAVES;                         // rows per block
  __shared__ LinX s_ly[NR + 4];     // rows Y0 - 2 .. Y0 + NR + 1 (reflected, source rows relative to the sub-rectangle)
  __shared__ int s_hot[NR + 4];     // 1: that row's source rows can reach the threshold on this strip
  __shared__ float s_rowmax[256];
  __shared__ float s_max[V1C_WAVES];
  __shared__ V1Run s_list[V1C_WAVES][V1C_LIST];
  const int strips_x = (W + V1_TILE_W - 1) / V1_TILE_W;
  const int sx_ = blockIdx.x % strips_x, by = blockIdx.x / strips_x;
  const int part = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x0 = sx_ * V1_TILE_W, Y0 = by * NR;
  const float* plane = net_out + ((size_t)b * C + part) * hs * ws;
  // source cells the block can reach (1-px blur halo, both bilinear taps)
  const int xa = x0 > 0 ? x0 - 1 : 0, xb = x0 + V1_TILE_W < W ? x0 + V1_TILE_W : W - 1;
  const int ya_ = Y0 > 0 ? Y0 - 1 : 0, yb_ = Y0 + NR < H ? Y0 + NR : H - 1;
  const int sxa = lin_x(xa, W, ws).s0, sxb = lin_x(xb, W, ws).s1;
  const int sya = lin_y(ya_, H, hs).s0, syb = lin_y(yb_, H, hs).s1;
  const int sw = sxb - sxa + 1, sh = syb - sya + 1;
  if (sw * sh > src_cap) {   // (the host sizes the buffer for the worst block; never expected)
    if (tid == 0) *overflow = 1;
    return;
  }
  float lm = -INFINITY;
  for (int i = tid; i < sw * sh; i += 64 * V1C_WAVES) {
    const int yy = i / sw, xx = i - yy * sw;
    const float v = plane[(sya + yy) * ws + sxa + xx];
    s_src[i] = v;
    lm = fmaxf(lm, v);
  }
  for (int o = 32; o > 0; o >>= 1) lm = fmaxf(lm, __shfl_xor(lm, o));
  if (l == 0) s_max[wave] = lm;
  __syncthreads();
  float blk_max = s_max[0];
  for (int i = 1; i < V1C_WAVES; i++) blk_max = fmaxf(blk_max, s_max[i]);
  // resize + blur are convex combinations up to a few ulps: nothing here can exceed thr
  const bool prune = thr > 0.f;
  if (prune && blk_max < thr * 0.999f) return;
  for (int r = tid; r < sh; r += 64 * V1C_WAVES) {
    float m = -INFINITY;
    for (int c = 0; c < sw; c++) m = fmaxf(m, s_src[r * sw + c]);
    s_rowmax[r] = m;
  }
  __syncthreads();
  for (int i = tid; i < NR + 4; i += 64 * V1C_WAVES) {
    const int r = Y0 - 2 + i;
    LinX t;
    t.s0 = t.s1 = 0;
    t.a0 = t.a1 = 0.f;
    int hot = 0;
    if (r >= -1 && r <= H) {
      t = lin_y(reflect101(r, H), H, hs);
      t.s0 -= sya;
      t.s1 -= sya;
      // (rows Y0 - 2 and Y0 + NR + 1 only feed the flags; their source rows may lie outside the staged rectangle)
      const bool staged = t.s0 >= 0 && t.s1 < sh;
      hot = !prune || !staged || fmaxf(s_rowmax[staged ? t.s0 : 0], s_rowmax[staged ? t.s1 : 0]) >= thr * 0.999f;
    }
    s_ly[i] = t;
    s_hot[i] = hot;
  }
  __syncthreads();
  const int ya = Y0 + V1C_ROWS * wave;
  const int yb = ya + V1C_ROWS < H ? ya + V1C_ROWS : H;
  if (ya >= H) return;

  // per-lane column tables: words 0..3 = columns x0 + 64 k + l, word 4 = the halo column of the end lanes
  int cs0[5], cs1[5];
  float ca0[5], ca1[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    int x = k < 4 ? x0 + 64 * k + l : (l == 63 ? x0 + V1_TILE_W : x0 - 1);
    x = x > W ? W : x;
    const LinX t = lin_x(reflect101(x, W), W, ws);
    // columns past the strip's reach (x > xb + 1 only exist beyond the image) are clamped into the staged rectangle
    cs0[k] = min(max(t.s0 - sxa, 0), sw - 1);
    cs1[k] = min(max(t.s1 - sxa, 0), sw - 1);
    ca0[k] = t.a0;
    ca1[k] = t.a1;
  }
  float hA[5], hB[5];
  int ida = -1, idb = -1;
  // row pass of the resized row r (reflected at the image border) -> t[0..3]
  auto row_t = [&](int r, float* t) {
    const LinX cy = s_ly[r - (Y0 - 2)];
    const int na = __builtin_amdgcn_readfirstlane(cy.s0), nb = __builtin_amdgcn_readfirstlane(cy.s1);
    if (na != ida || nb != idb) {   // a new source row pair: horizontal interpolation of both rows
      ida = na;
      idb = nb;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        hA[k] = s_src[ida * sw + cs0[k]] * ca0[k] + s_src[ida * sw + cs1[k]] * ca1[k];
        hB[k] = s_src[idb * sw + cs0[k]] * ca0[k] + s_src[idb * sw + cs1[k]] * ca1[k];
      }
    }
    float v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = hA[k] * cy.a0 + hB[k] * cy.a1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // left / right neighbour: the next lane; at the word seams lane 63 of the previous / lane 0 of the next word,
      // at the strip's ends the halo column (held by the end lanes themselves)
      const float lo = k == 0 ? v[4] : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k - 1]), 63));
      const float ro = k == 3 ? v[4] : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k + 1]), 0));
      const float lf = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lo), __float_as_int(v[k]), 0x138, 0xf, 0xf, false));
      const float rt = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ro), __float_as_int(v[k]), 0x130, 0xf, 0xf, false));
      t[k] = v[k] * 0.5f + (lf + rt) * 0.25f;
    }
  };
  // bit i of ns: some row of y-1 .. y+1, y = ya + i, can reach the threshold (the row needs its s); a row needs its
  // t when one of its three consumers needs its s
  unsigned long long ns;
  {
    const int i = ya + l - (Y0 - 2);
    ns = __ballot(l < V1C_ROWS && ya + l < yb && (s_hot[l < V1C_ROWS ? i - 1 : 0] | s_hot[l < V1C_ROWS ? i : 0] | s_hot[l < V1C_ROWS ? i + 1 : 0]) != 0);
  }
  const unsigned long long nt = ns | (ns << 1) | (ns >> 1);
  auto need_s = [&](int y) { return ((ns >> (y - ya)) & 1ull) != 0; };
  auto need_t = [&](int r) {
    return r < ya ? (ns & 1ull) != 0 : (((nt >> (r - ya)) & 1ull) != 0);   // (r - ya <= V1C_ROWS < 64)
  };

  int* cnt = n_runs + (size_t)b * n_parts + part;
  V1Run* out = runs + ((size_t)b * n_parts + part) * run_cap;
  V1Run* list = s_list[wave];
  int n_list = 0;
  auto flush = [&]() {
    if (n_list == 0) return;
    __builtin_amdgcn_wave_barrier();   // (the list was written by other lanes of this wave)
    int base = 0;
    if (l == 0) base = atomicAdd(cnt, n_list);
    base = __shfl(base, 0);
    for (int i = l; i < n_list; i += 64) {
      if (base + i < run_cap) out[base + i] = list[i];
      else *overflow = 1;
    }
    __builtin_amdgcn_wave_barrier();
    n_list = 0;
  };

  float tp[4] = {0.f, 0.f, 0.f, 0.f}, tc[4] = {0.f, 0.f, 0.f, 0.f}, tn[4] = {0.f, 0.f, 0.f, 0.f};
  if (need_t(ya - 1)) row_t(ya - 1, tp);
  if (need_t(ya)) row_t(ya, tc);
  for (int y = ya; y < yb; y++) {
    if (need_t(y + 1)) row_t(y + 1, tn);
    if (need_s(y)) {
      float v[4];
      unsigned long long mw[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        v[k] = tc[k] * 0.5f + (tp[k] + tn[k]) * 0.25f;
        mw[k] = __ballot(v[k] > thr && x0 + 64 * k + l < W);
      }
#ifdef V1C_NO_RUNS
      n_list += (int)__popcll(mw[0] ^ mw[1] ^ mw[2] ^ mw[3]);
      if (n_list == 0x7fffffff) *overflow = 1;
      if (0) {
#else
      if (mw[0] | mw[1] | mw[2] | mw[3]) {
#endif
        // ---- runs of this row, one after the other (uniform loop: a row holds a few).  Mask values are positive
        // floats, so their bit patterns order like the values: the run's maximum is an integer wave reduction
        // (4 DPP steps inside the rows of 16 lanes + 4 readlanes), its first position the lowest set bit of the
        // ballots of "equals the maximum" -- what a left-to-right walk with `>` finds.
        int n_row = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned long long nxt = (mw[k] >> 1) | (k < 3 ? (mw[k + 1] & 1ull) << 63 : 0ull);
          n_row += __popcll(mw[k] & ~nxt);
        }
        if (n_list + n_row > V1C_LIST) flush();
        int pos = 0;
        while (true) {
          const int start = v1c_next_bit(mw, pos, false);
          if (start >= 256) break;
          const int stop = v1c_next_bit(mw, start, true);   // first pixel past the run
          unsigned key = 0u;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int px = 64 * k + l;
            const unsigned kk = (px >= start && px < stop) ? __float_as_uint(v[k]) : 0u;
            key = kk > key ? kk : key;
          }
          key = v1c_max16(key);
          const unsigned m01 = max((unsigned)__builtin_amdgcn_readlane((int)key, 0), (unsigned)__builtin_amdgcn_readlane((int)key, 16));
          const unsigned m23 = max((unsigned)__builtin_amdgcn_readlane((int)key, 32), (unsigned)__builtin_amdgcn_readlane((int)key, 48));
          const unsigned best = max(m01, m23);
          int bx = 256;
#pragma unroll
          for (int k = 3; k >= 0; k--) {
            const int px = 64 * k + l;
            const unsigned long long eq = __ballot(px >= start && px < stop && __float_as_uint(v[k]) == best);
            if (eq) bx = 64 * k + __ffsll((long long)eq) - 1;
          }
          if (l == 0) {
            V1Run r;
            r.y = y;
            r.xs = x0 + start;
            r.xe = x0 + stop - 1;
            r.bx = x0 + bx;
            r.best = __uint_as_float(best);
            list[n_list] = r;
          }
          n_list++;
          pos = stop;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      tp[k] = tc[k];
      tc[k] = tn[k];
    }
  }
#ifndef V1C_NO_RUNS
  flush();
#endif
}

// ---------------------------------------------------------------------------
// Literal getKeypoints for blobs that are not "simple" (utils.py:14-22 = cv2.findContours(RETR_TREE,
// CHAIN_APPROX_SIMPLE) + cv2.fillConvexPoly + minMaxLoc; oracle: decode_v1.py blob_peaks / find_contours_tree /
// fill_convex_poly, which state the OpenCV sources).
//
// A blob with exactly ONE run of mask pixels in each of its rows has no hole, and fillConvexPoly paints exactly
// its pixels: both edge walkers follow a y-monotone boundary chain whose vertical and diagonal pieces they hit
// exactly, and the horizontal pieces are outline pixels.  For such blobs -- every isolated bump -- the connected
// component's first raster maximum IS the reference's keypoint and nothing below runs.  Every other blob (several
// runs in a row: U shapes, rings, people merged side by side) goes through the literal algorithm on the run-length
// representation, ONE THREAD PER BLOB (blobs are independent: a border only touches pieces of its own blob):
// Suzuki-Abe border following as OpenCV codes it (pixel tests = binary searches in the row's sorted pieces), hole
// borders found at run ends that no earlier trace has "right-marked", the polygon's outline + two-walker scan
// fill, the maximum of the smoothed map over what was painted (a piece whose stored maximum lies under the paint
// contributes it directly; a piece whose maximum cannot beat the running one is skipped; only what is left is
// re-evaluated pixel by pixel with K-A's arithmetic).
// ---------------------------------------------------------------------------
#define V1_TRACE_CAP 262144 // border pixels of one contour (runaway guard)

struct V1Grid {
  const short2* srun;  // LDS: mask pieces {xs, xe}; rows ascending, xs ascending within a row (a run that crosses
  const int* rowoff;   //      a K-A tile seam is two touching pieces); row y = [rowoff[y], rowoff[y+1])
  int W, H;
};

#ifdef V1B_STAMP
// Diagnostic build only (tools/v1_blobs_stamp.py): where v1_blobs_kernel spends its time.
__device__ unsigned long long v1b_dbg[40];
extern "C" int ph_debug_v1_blobs(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[40] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(v1b_dbg), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(v1b_dbg), 40 * sizeof(unsigned long long));
}
#define V1B_ADD(i, v) atomicAdd(&v1b_dbg[i], (unsigned long long)(v))
#define V1B_NOW() __builtin_readcyclecounter()
#else
#define V1B_ADD(i, v)
#define V1B_NOW() 0ll
#endif

// sorted position of the piece that holds pixel (x, y), -1 for a background / outside pixel
__device__ __forceinline__ int v1_piece_at(const V1Grid& g, int x, int y) {
  if (x < 0 || y < 0 || x >= g.W || y >= g.H) return -1;
  const int r0 = g.rowoff[y];
  int lo = r0, hi = g.rowoff[y + 1];
  while (lo < hi) {  // first piece with xs > x
    const int mid = (lo + hi) >> 1;
    if (g.srun[mid].x <= x) lo = mid + 1;
    else hi = mid;
  }
  return (lo > r0 && g.srun[lo - 1].y >= x) ? lo - 1 : -1;
}

// The same for a caller that probes neighbouring pixels one after the other (border following, outline chains): `rel`
// = position inside its row of the first piece that starts right of the last probe.  Pieces of adjacent rows of a
// blob line up, so the guess is almost always right, and checking it costs two dependent LDS reads (the row's offsets,
// then the two pieces around the guess) instead of the binary search's five to seven.
__device__ __forceinline__ int v1_piece_near(const V1Grid& g, int x, int y, int& rel) {
  if (x < 0 || y < 0 || x >= g.W || y >= g.H) return -1;
  const int r0 = g.rowoff[y], r1 = g.rowoff[y + 1];
  int q = r0 + rel;
  q = q < r0 ? r0 : (q > r1 ? r1 : q);
  const short2 a = q > r0 ? g.srun[q - 1] : make_short2(-32768, -32768);
  const short2 b = q < r1 ? g.srun[q] : make_short2(32767, 32767);
  if (a.x <= x && x < b.x) {   // q is the first piece with xs > x
    rel = q - r0;
    return a.y >= x ? q - 1 : -1;
  }
  int lo = r0, hi = r1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (g.srun[mid].x <= x) lo = mid + 1;
    else hi = mid;
  }
  rel = lo - r0;
  return (lo > r0 && g.srun[lo - 1].y >= x) ? lo - 1 : -1;
}

// the smoothed map at one pixel, with the operations of v1_runs_kernel (resize, row pass, column pass)
// (not inlined: the pixel-by-pixel re-evaluation is rare -- under-filled shapes only -- and its 36 loads at every place
// a span is painted made the blobs kernel outgrow the instruction cache)
__device__ __attribute__((noinline)) float v1_smooth_at(const float* src, int hs, int ws, int W, int H, int x, int y) {
  float t[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const LinX cy = lin_y(reflect101(y - 1 + r, H), H, hs);
    float h[3];
#pragma unroll
    for (int c = 0; c < 3; c++) h[c] = resize_tab(src, ws, lin_x(reflect101(x - 1 + c, W), W, ws), cy);
    t[r] = h[1] * 0.5f + (h[0] + h[2]) * 0.25f;
  }
  return t[1] * 0.5f + (t[0] + t[2]) * 0.25f;
}

__device__ __forceinline__ unsigned long long v1_key(float v, int x, int y, int W) {
  return ((unsigned long long)__float_as_uint(v) << 32) | (0xffffffffu - (unsigned)(y * W + x));
}

// the 8 neighbour offsets, E NE N NW W SW S SE (y grows downwards), as 2-bit fields: no table in memory
__device__ __forceinline__ int v1_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int v1_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

// One thread's contour vertices: a block of the (frame, part)'s pool that grows by doubling -- in place while it is
// the pool's last block, else by moving -- and is kept for the thread's next contour.
struct V1Verts {
  short2* pool;
  int* used;      // LDS: entries of the pool handed out
  int pool_cap;
  int base, cap;
  __device__ bool push(int n, short2 v) {
    if (n == cap) {
      const int want = cap ? 2 * cap : 64;
      if (cap && atomicCAS(used, base + cap, base + want) == base + cap) {
        if (base + want > pool_cap) return false;
      } else {
        const int nb = atomicAdd(used, want);
        if (nb + want > pool_cap) return false;
        for (int i = 0; i < n; i++) pool[nb + i] = pool[base + i];
        base = nb;
      }
      cap = want;
    }
    pool[base + n] = v;
    return true;
  }
};

// A blob is walked by a GROUP of 8 adjacent lanes (control flow uniform within the group): the 8 neighbour probes
// of a border step, the rows of the fill, the edges of the outline and the rows of the hole search are spread over
// the lanes; lane 0 of the group owns the stores.
__device__ __forceinline__ unsigned v1_gballot(bool p, int lane) {
  return (unsigned)((__ballot(p) >> (lane & 56)) & 0xffull);
}

// icvFetchContour on the pieces: follows the border that starts at (x0, y0), writes the CHAIN_APPROX_SIMPLE
// vertices and records, for every border pixel whose right neighbour the search examined and found empty, the
// contour id on that pixel's piece (the "-nbd" mark of the library: only run ends can carry it).  Returns the
// vertex count, -1 if the pool is exhausted or the walk does not close.  One group; `verts` is lane 0's.
__device__ __forceinline__ int v1_trace(const V1Grid& g, int x0, int y0, bool is_hole, unsigned short cid, V1Verts& verts,
                                        unsigned short* mark, int lane) {
  const int r = lane & 7;
  int n = 0, ok = 1;
  int s_end = is_hole ? 0 : 4;
  // the library looks clockwise from s_end - 1 for the first set neighbour; reaching s_end again = single pixel
  int s = (s_end - 1 - r) & 7;
  int rel = 0;                       // this lane's guess for v1_piece_near
  unsigned hit = v1_gballot(r < 7 && v1_piece_near(g, x0 + v1_dx(s), y0 + v1_dy(s), rel) >= 0, lane);
  int j3 = v1_piece_at(g, x0, y0);   // piece of the current border pixel
  if (!hit) {
    if (r == 0) {
      mark[j3] = cid;
      ok = verts.push(0, make_short2((short)x0, (short)y0));
    }
    verts.base = __shfl(verts.base, 0, 8);
    __threadfence_block();
    return __shfl(ok, 0, 8) ? 1 : -1;
  }
  s = (s_end - __ffs(hit)) & 7;
  const int i1x = x0 + v1_dx(s), i1y = y0 + v1_dy(s);
  int i3x = x0, i3y = y0, prev_s = s ^ 4;
  int guard = 0;
  for (; guard < V1_TRACE_CAP; guard++) {
    s_end = s;
    const int sk = (s_end + 1 + r) & 7;   // lane r: the (r+1)-th direction of the counter-clockwise search
    const int jk = v1_piece_near(g, i3x + v1_dx(sk), i3y + v1_dy(sk), rel);
    hit = v1_gballot(jk >= 0, lane);
    if (!hit) return -1;                  // (cannot happen: the pixel the walk came from is set)
    const int k0 = __ffs(hit) - 1;
    s = (s_end + 1 + k0) & 7;
    const int i4x = i3x + v1_dx(s), i4y = i3y + v1_dy(s), j4 = __shfl(jk, k0, 8);   // lane k0's probe
    if ((unsigned)(s - 1) < (unsigned)s_end && r == 0) mark[j3] = cid;
    if (s != prev_s) {
      if (r == 0 && ok) ok = verts.push(n, make_short2((short)i3x, (short)i3y));
      n++;
      prev_s = s;
    }
    if (i4x == x0 && i4y == y0 && i3x == i1x && i3y == i1y) break;
    i3x = i4x;
    i3y = i4y;
    j3 = j4;
    s = (s + 4) & 7;
  }
  verts.base = __shfl(verts.base, 0, 8);
  __threadfence_block();   // lane 0's vertices, read by the whole group from here on
  if (r == 0) V1B_ADD(13, guard);
  return (guard < V1_TRACE_CAP && __shfl(ok, 0, 8)) ? n : -1;
}

// running first-raster maximum of the smoothed map over painted mask pixels (one thread's state)
struct V1Paint {
  V1Grid g;
  const float* bval;            // LDS, per piece: its maximum of the smoothed map
  const unsigned short* byrow;  // LDS, per piece: its index in R
  const V1Run* R;
  const float* plane;
  int hs, ws;
  unsigned long long cur;
};

// pixels [lo, hi] of piece j (row y) are painted
__device__ __forceinline__ void v1_consider(V1Paint& p, int j, int y, int lo, int hi) {
  if (__float_as_uint(p.bval[j]) < (unsigned)(p.cur >> 32)) return;  // nothing in the piece can win
  const V1Run r = p.R[p.byrow[j]];
  if (r.bx >= lo && r.bx <= hi) {
    const unsigned long long k = v1_key(r.best, r.bx, y, p.g.W);
    p.cur = k > p.cur ? k : p.cur;
    return;
  }
  V1B_ADD(3, hi - lo + 1);
  V1B_ADD(4, 1);
  for (int x = lo; x <= hi; x++) {
    const unsigned long long k = v1_key(v1_smooth_at(p.plane, p.hs, p.ws, p.g.W, p.g.H, x, y), x, y, p.g.W);
    p.cur = k > p.cur ? k : p.cur;
  }
}

// columns [x1, x2] of row y are painted: pieces of ANY blob under them count, as in cv2
__device__ __forceinline__ void v1_paint_span(V1Paint& p, int y, int x1, int x2) {
  if (y < 0 || y >= p.g.H) return;
  int lo = p.g.rowoff[y];
  const int r1 = p.g.rowoff[y + 1];
  for (int hi = r1; hi - lo > 4;) {   // a crowded row: skip the pieces that end left of x1
    const int mid = (lo + hi) >> 1;
    if (p.g.srun[mid].y < x1) lo = mid + 1;
    else hi = mid;
  }
  for (int j = lo; j < r1; j++) {
    const short2 q = p.g.srun[j];
    if (q.x > x2) break;
    if (q.y < x1) continue;
    v1_consider(p, j, y, x1 > q.x ? x1 : q.x, x2 < q.y ? x2 : q.y);
  }
}

template <int GW>
__device__ __forceinline__ unsigned long long v1_gmax(unsigned long long k) {
  for (int o = 1; o < GW; o <<= 1) {
    const unsigned long long t = __shfl_xor(k, o, GW);
    k = t > k ? t : k;
  }
  return k;
}

// FillConvexPoly (lineType 8, shift 0) of the contour v[0 .. n): the two edge walkers' span in every row from the
// polygon's top row on, then the outline (the polygon's edges are straight 8-connected pixel chains).  One group of
// GW lanes (8: the one-group walk; 64: the wave of v1_fast_blob): every lane runs the walkers (a few additions per
// row), lane k paints the k-th row of each GW; outline edges are dealt round-robin.  On return p.cur is the group's
// maximum in every lane.
// (forced inline, like the other helpers that take the paint state: one real call is enough to put that state -- the
// running maximum included -- into scratch memory for the whole kernel)
template <int GW, bool FILL = true>
__device__ __forceinline__ void v1_fill_paint(V1Paint& p, const short2* v, int n, int lane) {
  const int W = p.g.W, H = p.g.H, r = lane & (GW - 1), r8 = lane & 7;
  const long long XY_ONE = 1 << 16;
  int xmin = 0x7fffffff, xmax = -0x7fffffff, ymax = -0x7fffffff, ykey = 0x7fffffff;   // ykey = (y << 16) | index
  for (int i = r; i < n; i += GW) {
    const short2 q = v[i];
    const int k = ((int)q.y << 16) | i;
    ykey = k < ykey ? k : ykey;
    ymax = ymax > q.y ? ymax : q.y;
    xmax = xmax > q.x ? xmax : q.x;
    xmin = xmin < q.x ? xmin : q.x;
  }
  for (int o = 1; o < GW; o <<= 1) {
    const int a = __shfl_xor(ykey, o, GW), b = __shfl_xor(ymax, o, GW), c = __shfl_xor(xmax, o, GW),
              d = __shfl_xor(xmin, o, GW);
    ykey = a < ykey ? a : ykey;
    ymax = b > ymax ? b : ymax;
    xmax = c > xmax ? c : xmax;
    xmin = d < xmin ? d : xmin;
  }
  const int ymin = ykey >> 16, imin = ykey & 0xffff;   // the first vertex of the top row
  if (FILL && !(n < 3 || xmax < 0 || ymax < 0 || xmin >= W || ymin >= H)) {
    ymax = ymax < H - 1 ? ymax : H - 1;
    int edges = n;
    int e_idx[2] = {imin, imin}, e_ye[2] = {ymin, ymin}, e_di[2] = {1, n - 1};
    long long e_x[2] = {-XY_ONE, -XY_ONE}, e_dx[2] = {0, 0};
    // The walkers read the vertices in order, one forwards, one backwards, and the pool is in global memory: lane k
    // of every 8 fetches the k-th vertex ahead, the others take it by shuffle -- one memory latency per eight
    // vertices instead of one each.
    const short2 vtop = v[imin];
    short2 e_v[2] = {vtop, vtop};   // the vertex at e_idx
    int wbase[2] = {-1, -1};
    int wbuf[2] = {0, 0};
    auto vert = [&](int i, int idx) -> short2 {
      int off = i == 0 ? idx - wbase[0] : wbase[1] - idx;
      off = off < 0 ? off + n : off;
      if (wbase[i] < 0 || off >= 8) {   // (uniform within the group)
        wbase[i] = idx;
        int j = i == 0 ? idx + r8 : idx - r8 + 8 * n;
        j %= n;
        const short2 q = v[j];
        wbuf[i] = ((int)(unsigned short)q.y << 16) | (int)(unsigned short)q.x;
        off = 0;
      }
      const int raw = __shfl(wbuf[i], off, 8);
      return make_short2((short)(raw & 0xffff), (short)(raw >> 16));
    };
    const bool narrow = W < 16000;      // (xe - xs) * 2 then fits 32 bits: same quotient from the cheaper division
    int y = ymin, cnt = 0, my_y = -1, my_x1 = 1, my_x2 = 0;
    do {
      for (int i = 0; i < 2; i++) {
        if (y >= e_ye[i]) {
          short2 q0 = e_v[i];
          const int di = e_di[i];
          int idx = e_idx[i] + di;
          if (idx >= n) idx -= n;
          for (; edges-- > 0;) {
            const short2 q = vert(i, idx);
            const int ty = q.y;
            if (ty > y) {
              const long long xs = (long long)q0.x << 16, xe = (long long)q.x << 16;
              e_ye[i] = ty;
              // C division: towards zero, as the library's
              if (narrow) e_dx[i] = (long long)(((int)(xe - xs) * 2 + (ty - y)) / (2 * (ty - y)));
              else e_dx[i] = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
              e_x[i] = xs;
              e_idx[i] = idx;
              e_v[i] = q;
              break;
            }
            q0 = q;
            idx += di;
            if (idx >= n) idx -= n;
          }
        }
      }
      if (edges < 0) break;
      if (y >= 0) {
        const int left = e_x[0] > e_x[1] ? 1 : 0, right = 1 - left;
        int xx1 = (int)((e_x[left] + (XY_ONE >> 1)) >> 16), xx2 = (int)((e_x[right] + (XY_ONE >> 1)) >> 16);
        if (xx2 >= 0 && xx1 < W && (cnt & (GW - 1)) == r) {
          my_y = y;
          my_x1 = xx1 < 0 ? 0 : xx1;
          my_x2 = xx2 >= W ? W - 1 : xx2;
        }
        if ((++cnt & (GW - 1)) == 0) {
          if (my_y >= 0) v1_paint_span(p, my_y, my_x1, my_x2);
          my_y = -1;
        }
      }
      e_x[0] += e_dx[0];
      e_x[1] += e_dx[1];
    } while (++y <= ymax);
    if (my_y >= 0) v1_paint_span(p, my_y, my_x1, my_x2);
    p.cur = v1_gmax<GW>(p.cur);   // the outline's skip test starts from the whole fill's maximum
    if (r == 0) V1B_ADD(14, ymax - ymin + 1);
  }
  [[maybe_unused]] const long long to_ = V1B_NOW();
  // outline
  if (n == 1) {
    if (r == 0) v1_paint_span(p, v[0].y, v[0].x, v[0].x);
  } else {
    int cj = -1, cy = -1;  // piece of the previous outline pixel
    int rel = 0;
    for (int e = r; e < n; e += GW) {
      const short2 a = v[e == 0 ? n - 1 : e - 1], c = v[e];
      const int dx = c.x - a.x, dy = c.y - a.y;
      if (dy == 0) {
        v1_paint_span(p, a.y, dx > 0 ? a.x : c.x, dx > 0 ? c.x : a.x);
      } else {
        const int steps = abs(dy);  // vertical or diagonal chain
        const int sx = (dx > 0) - (dx < 0), sy = dy > 0 ? 1 : -1;
        for (int t = 1; t <= steps; t++) {
          const int x = a.x + t * sx, y = a.y + t * sy;
          if (!(cj >= 0 && y == cy && p.g.srun[cj].x <= x && x <= p.g.srun[cj].y)) {
            cj = v1_piece_near(p.g, x, y, rel);
            cy = y;
          }
          if (cj >= 0) v1_consider(p, cj, y, x, x);
        }
      }
    }
  }
  p.cur = v1_gmax<GW>(p.cur);
  if (r == 0) V1B_ADD(15, V1B_NOW() - to_);
}

// ---------------------------------------------------------------------------
// All border loops of one blob at once, by one WAVE (the walk above follows one border pixel after the other with one
// group of 8 lanes: ~2 000 cycles per pixel, and the largest blob of a plane was the kernel's critical path).
//
// A step of the library's border following is a function of its state alone -- the border pixel and the direction
// `back` of the pixel the walk came from: it sweeps the neighbours counter-clockwise from back + 1 to the first set
// one, emits the pixel if the move's direction differs from the last move's (= back ^ 4), right-marks the pixel if
// the sweep passed an empty east neighbour, and moves.  Every border loop of the blob (outer and holes) is a cycle
// of such states.  The states whose sweep passes an empty WEST or EAST neighbour are known without walking: the pixel
// is a run's first / last pixel and `back` is the first set neighbour clockwise from that side (exactly the
// library's search for a border's second pixel).  Between two of them the walk runs along the top or bottom of a
// run: one or two steps on the staircases that blob borders are.  So:
//   - every run end (a "crack": piece, side) is a node; one LANE per node walks its segment to the next node and
//     records that node, the vertices the segment emits (into a scratch part of the vertex pool) and whether the
//     node's own sweep passes the east side (a crack whose sweep starts at the other side of the same pixel is part
//     of that node's step);
//   - the cycles of the node graph are the border loops: the one through the blob's first pixel is the outer border,
//     every other one a hole border, which the library starts at its first east crack in raster order (the first
//     run end no earlier border has right-marked) -- a rotation of the cycle;
//   - vertices are copied to their places in loop order, every piece gets the contour that swept its east crack as
//     its mark, and the loops are filled by the wave's groups side by side.
// The tables live in what the block's LDS has left over; a blob that does not fit (or trips one of the guards
// below) is left untouched and goes through the one-group walk.
// ---------------------------------------------------------------------------
#define V1F_K 6          // vertices one segment may emit
#define V1F_STEPS 16384  // steps of one segment (runaway guard)
#define V1F_LOOPS 32     // border loops of one blob
#define V1F_UNITS (10 + 2 * V1F_K)   // pool entries (4 B) the tables of one piece take at most

struct V1FastCtx {
  const int* parent;
  unsigned short* mark;
  char* lds;            // the block's dynamic LDS
  int* s_free;          // first free byte of it
  int lds_limit;
  short2* vpool;
  int* s_vused;
  int vert_cap;         // entries of the pool
  int fast_cap;         // ... the fast walk may fill it up to (the rest is the one-group walk's, as before)
  int* ltab;            // LDS, this wave's: V1F_LOOPS x {vertices, first east piece, base, -}
  int W, kp_cap;
  int* s_ncont;
  int* c_disc;
  short* c_par;
  short* c_link;
  unsigned char* c_hole;
  unsigned long long* c_key;
  int* overflow;
};

__device__ __forceinline__ void v1_wave_sync() {
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int v1_wave_incl_scan(int v, int lane) {
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// compare-and-swap bump allocation (nothing is taken when it does not fit)
__device__ __forceinline__ int v1_bump(int* ctr, int need, int limit) {
  int old = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (true) {
    if (old + need > limit) return -1;
    const int prev = atomicCAS(ctr, old, old + need);
    if (prev == old) return old;
    old = prev;
  }
}

// tables in global memory that one lane writes and another reads: loads that do not stop at the CU's cache
template <typename T>
__device__ __forceinline__ T v1_gl(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The 8 neighbours of (x, y) as a bit per direction (E NE N NW W SW S SE) and the piece of each, from the rows
// y - 1, y, y + 1 -- all three looked up at once: the rows' offsets, then the three pieces around each row's guess
// (rel[r] = position inside the row of the first piece that starts right of x + 1, the caller's last answer), so a
// neighbourhood costs two LDS round trips when the guesses hold; a wrong guess is repaired by a binary search.  Only
// the last two pieces that start at or before x + 1 can reach columns x - 1 .. x + 1 (touching pieces exist at tile
// seams only, 256 columns apart).  top_xe: the end of the last piece of row y - 1 that starts at or before x + 1;
// bot_xs: the start of the first piece of row y + 1 that ends at or after x - 1 (for the straight stretches).
__device__ __forceinline__ int v1_nbhd(const V1Grid& g, int x, int y, int* rel, int* pc, int& top_xe, int& bot_xs) {
  const short2 lo_ = make_short2(-32768, -32768), hi_ = make_short2(32767, 32767);
  int r0[3], r1[3], q[3];
  short2 a[3], b[3], c[3];
  bool in[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int yy = y - 1 + r;
    in[r] = yy >= 0 && yy < g.H;
    r0[r] = in[r] ? g.rowoff[yy] : 0;
    r1[r] = in[r] ? g.rowoff[yy + 1] : 0;
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    int t = r0[r] + rel[r];
    t = t < r0[r] ? r0[r] : (t > r1[r] ? r1[r] : t);
    q[r] = t;
    a[r] = t > r0[r] ? g.srun[t - 1] : lo_;
    c[r] = t - 1 > r0[r] ? g.srun[t - 2] : lo_;
    b[r] = t < r1[r] ? g.srun[t] : hi_;
  }
  int bits[3];
  int p[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    if (!(a[r].x <= x + 1 && x + 1 < b[r].x)) {   // the guess was wrong
      int lo = r0[r], hi = r1[r];
      while (lo < hi) {   // first piece with xs > x + 1
        const int mid = (lo + hi) >> 1;
        if (g.srun[mid].x <= x + 1) lo = mid + 1;
        else hi = mid;
      }
      q[r] = lo;
      a[r] = lo > r0[r] ? g.srun[lo - 1] : lo_;
      c[r] = lo - 1 > r0[r] ? g.srun[lo - 2] : lo_;
      b[r] = lo < r1[r] ? g.srun[lo] : hi_;
    }
    rel[r] = q[r] - r0[r];
    int m = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int cx = x - 1 + k;
      p[r][k] = -1;
      if (a[r].x <= cx && cx <= a[r].y) {
        m |= 1 << k;
        p[r][k] = q[r] - 1;
      } else if (c[r].x <= cx && cx <= c[r].y) {
        m |= 1 << k;
        p[r][k] = q[r] - 2;
      }
    }
    bits[r] = m;
  }
  top_xe = a[0].y;
  bot_xs = c[2].y >= x - 1 ? c[2].x : (a[2].y >= x - 1 ? a[2].x : b[2].x);
  pc[0] = p[1][2]; pc[1] = p[0][2]; pc[2] = p[0][1]; pc[3] = p[0][0]; pc[4] = p[1][0];
  pc[5] = p[2][0]; pc[6] = p[2][1]; pc[7] = p[2][2];
  const int top = bits[0], mid = bits[1], bot = bits[2];
  return ((mid >> 2) & 1) | ((top >> 2) & 1) << 1 | ((top >> 1) & 1) << 2 | (top & 1) << 3 | (mid & 1) << 4 |
         (bot & 1) << 5 | ((bot >> 1) & 1) << 6 | ((bot >> 2) & 1) << 7;
}

// FillConvexPoly's scan fill by a whole wave.  Each of the two edge walkers steps through the vertices in its
// direction from the top vertex and changes edge whenever its row reaches the current edge's end: its edges are the
// vertices that lie lower than every vertex before them (a running maximum), each starting at its predecessor's
// column in the row of the previous such vertex, with a slope fixed there -- so a row's two edge positions follow
// from a prefix maximum instead of from the rows before it.  The library's budget of n vertex visits for both
// walkers together ends the fill at the first row where the visits so far exceed n; visits only grow, so a row is
// painted iff the two edges that cover it were reached within n visits.  `tab`: 6 ints per row (global scratch).
// Returns false (nothing painted) when the scratch does not fit.
__device__ __forceinline__ bool v1_fill_wave(V1Paint& p, const short2* v, int n, int lane, const V1FastCtx& cx) {
  const int W = p.g.W, H = p.g.H;
  int xmin = 0x7fffffff, xmax = -0x7fffffff, ymax = -0x7fffffff, ykey = 0x7fffffff;   // ykey = (y << 16) | index
  for (int i = lane; i < n; i += 64) {
    const short2 q = v[i];
    const int k = ((int)q.y << 16) | i;
    ykey = k < ykey ? k : ykey;
    ymax = ymax > q.y ? ymax : q.y;
    xmax = xmax > q.x ? xmax : q.x;
    xmin = xmin < q.x ? xmin : q.x;
  }
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_xor(ykey, o), b = __shfl_xor(ymax, o), c = __shfl_xor(xmax, o), d = __shfl_xor(xmin, o);
    ykey = a < ykey ? a : ykey;
    ymax = b > ymax ? b : ymax;
    xmax = c > xmax ? c : xmax;
    xmin = d < xmin ? d : xmin;
  }
  const int ymin = ykey >> 16, imin = ykey & 0xffff;
  if (n < 3 || xmax < 0 || ymax < 0 || xmin >= W || ymin >= H || ymax <= ymin) return true;   // nothing to fill
  const int rows = ymax - ymin;   // rows ymin .. ymax - 1 can be painted
  int tb = 0;
  if (lane == 0) tb = v1_bump(cx.s_vused, rows * 6, cx.fast_cap);
  tb = __shfl(tb, 0);
  if (tb < 0) return false;
  int* tab = (int*)(cx.vpool + tb);
  for (int k = lane; k < rows * 2; k += 64) tab[k * 3 + 2] = 0x7fffffff;   // visits: no edge covers the row
  v1_wave_sync();
  const int chunk = (n + 63) / 64;
  for (int wk = 0; wk < 2; wk++) {
    const int t0 = 1 + lane * chunk, t1 = t0 + chunk < n + 1 ? t0 + chunk : n + 1;
    auto idx = [&](int t) {
      int k = wk == 0 ? imin + t : imin - t + n;
      return k >= n ? k - n : k;
    };
    int mx = -0x7fffffff;
    for (int t = t0; t < t1; t++) {
      const int y = v[idx(t)].y;
      mx = mx > y ? mx : y;
    }
    int pm = mx;   // inclusive prefix maximum over the lanes, then the maximum of everything before this lane
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(pm, o);
      if (lane >= o) pm = pm > u ? pm : u;
    }
    pm = __shfl_up(pm, 1);
    if (lane == 0 || pm < ymin) pm = ymin;
    if (t0 < t1) {
      short2 q0 = v[idx(t0 - 1)];
      for (int t = t0; t < t1; t++) {
        const short2 q = v[idx(t)];
        if (q.y > pm) {   // the walker's edge from row pm to row q.y
          const int dy = q.y - pm;
          const int dx = (((int)q.x - (int)q0.x) * 131072 + dy) / (2 * dy);   // C division, as the library's
          const int w0 = (int)(unsigned short)q0.x | pm << 16;
          for (int y = pm; y < q.y; y++) {
            int* e = tab + ((y - ymin) * 2 + wk) * 3;
            e[0] = w0;
            e[1] = dx;
            e[2] = t;
          }
          pm = q.y;
        }
        q0 = q;
      }
    }
  }
  v1_wave_sync();
  // lane k takes rows k * per, k * per + 1, ...: the first round already samples the whole height of the polygon, and
  // with the wave's maximum after every round the later ones skip most pieces without looking at them
  const int per = (rows + 63) / 64;
  for (int k = 0; k < per; k++) {
    const int y = ymin + lane * per + k;
    if (y < ymax) {
      const int* e = tab + (y - ymin) * 6;
      const int a0 = v1_gl(e), a1 = v1_gl(e + 1), a2 = v1_gl(e + 2), b0 = v1_gl(e + 3), b1 = v1_gl(e + 4), b2 = v1_gl(e + 5);
      if (a2 != 0x7fffffff && b2 != 0x7fffffff && a2 + b2 <= n) {
        const long long ex0 = ((long long)(a0 & 0xffff) << 16) + (long long)(y - (a0 >> 16)) * a1;
        const long long ex1 = ((long long)(b0 & 0xffff) << 16) + (long long)(y - (b0 >> 16)) * b1;
        const long long xl = ex0 > ex1 ? ex1 : ex0, xr = ex0 > ex1 ? ex0 : ex1;
        const int xx1 = (int)((xl + 32768) >> 16), xx2 = (int)((xr + 32768) >> 16);
        if (xx2 >= 0 && xx1 < W) v1_paint_span(p, y, xx1 < 0 ? 0 : xx1, xx2 >= W ? W - 1 : xx2);
      }
    }
    p.cur = v1_gmax<64>(p.cur);
  }
  p.cur = v1_gmax<64>(p.cur);
  return true;
}

// a table of the wave walk: in LDS or, when the block had none left, in the vertex pool
template <typename T>
struct V1Tab {
  T* l;
  T* g;
  bool in_lds;
  __device__ __forceinline__ T get(int k) const { return in_lds ? l[k] : v1_gl(g + k); }
  __device__ __forceinline__ void set(int k, T v) const {
    if (in_lds) l[k] = v;
    else g[k] = v;
  }
};

// One wave (all 64 lanes, uniform arguments).  `i` = the blob's number = its outer border's contour, `me` = its root
// piece, rows y_first .. y_last.  True: every border loop of the blob is in the contour tables, filled, with its
// arg-max key; false: nothing was changed.
__device__ __forceinline__ bool v1_fast_blob(const V1Grid& g, V1Paint& pt, const V1FastCtx& cx, int i, int me,
                                             int y_first, int y_last, int lane) {
  const int* rowoff = g.rowoff;
  const short2* srun = g.srun;
  [[maybe_unused]] const long long t0_ = V1B_NOW();
  // (1) the blob's pieces, a lane per row
  int np = 0, xmin = 0x7fff, xmax = -1;
  for (int yb = y_first; yb <= y_last; yb += 64) {
    const int y = yb + lane;
    int c = 0;
    if (y <= y_last)
      for (int j = rowoff[y]; j < rowoff[y + 1]; j++)
        if (cx.parent[j] == me) {
          const short2 q = srun[j];
          xmin = q.x < xmin ? q.x : xmin;
          xmax = q.y > xmax ? q.y : xmax;
          c++;
        }
    np += __shfl(v1_wave_incl_scan(c, lane), 63);
  }
  for (int o = 32; o; o >>= 1) {
    const int t0 = __shfl_xor(xmin, o), t1 = __shfl_xor(xmax, o);
    xmin = t0 < xmin ? t0 : xmin;
    xmax = t1 > xmax ? t1 : xmax;
  }
  if (np < 1 || np > 5000) return false;   // (16-bit vertex sums: at most V1F_K per node)
  if (lane == 0) V1B_ADD(23, np);
  const int nn = 2 * np;
  int* ltab = cx.ltab;
  // (2) tables (32 B per piece) in LDS when the block has that much left, else in the vertex pool -- generic
  // pointers, and every read of what another lane wrote goes through v1_gl; the segments' scratch vertices
  // (2 V1F_K entries per piece) always in the pool
  // and, when the block has that left as well, a bitmap of the blob's bounding box (one empty pixel around it) for the
  // segment walk: a pixel's neighbourhood is then six word reads instead of three searches in the rows' pieces
  const int bw = (xmax - xmin + 3 + 31) / 32 + 1, bh = y_last - y_first + 3;   // words per row (one to spare), rows
  int a = 0, tb = 0, ab = -1;
  if (lane == 0) {
    a = v1_bump(cx.s_free, 32 * np, cx.lds_limit);
    // (only while plenty is left: the tables of the blobs still to come matter more -- from the pool they cost a wave
    // six times the set-up time per round of segments)
    if (a >= 0) ab = v1_bump(cx.s_free, bw * bh * 4, cx.lds_limit - 24576);
    tb = v1_bump(cx.s_vused, (a < 0 ? V1F_UNITS : V1F_UNITS - 8) * np, cx.fast_cap);
  }
  a = __shfl(a, 0);
  ab = __shfl(ab, 0);
  tb = __shfl(tb, 0);
  if (tb < 0) {
    if (lane == 0) V1B_ADD(17, 1);
    return false;
  }
  short2* tmp = cx.vpool + tb;                                      // nn x V1F_K: a segment's vertices
  // (8-byte aligned in the pool: the scratch vertices take 2 K entries per piece, the rounding at most 1)
  // (the tables: LDS pointers and global pointers kept apart -- through ONE generic pointer every access became a flat
  // instruction, and a flat load waits for the wave's outstanding global stores, the scratch vertices, as well)
  const bool tl = a >= 0;
  char* const tl_base = cx.lds + (tl ? a : 0);
  char* const tg_base = (char*)(cx.vpool + ((tb + 2 * V1F_K * np + 1) & ~1));
  V1Tab<unsigned long long> jw = {(unsigned long long*)tl_base, (unsigned long long*)tg_base, tl};   // nn: jumping words
  V1Tab<unsigned short> succ = {(unsigned short*)(tl_base + 8 * nn), (unsigned short*)(tg_base + 8 * nn), tl};     // nn: next node
  V1Tab<unsigned short> st = {(unsigned short*)(tl_base + 10 * nn), (unsigned short*)(tg_base + 10 * nn), tl};     // nn: predecessor, then the loop's first node
  V1Tab<unsigned short> plist = {(unsigned short*)(tl_base + 12 * nn), (unsigned short*)(tg_base + 12 * nn), tl};  // np: sorted position of the piece
  V1Tab<unsigned short> prow = {(unsigned short*)(tl_base + 13 * nn), (unsigned short*)(tg_base + 13 * nn), tl};   // np: its row
  V1Tab<unsigned char> cntfl = {(unsigned char*)(tl_base + 14 * nn), (unsigned char*)(tg_base + 14 * nn), tl};     // nn: vertices | 0x40 sweeps east | 0x80 node
  V1Tab<unsigned char> loop = {(unsigned char*)(tl_base + 15 * nn), (unsigned char*)(tg_base + 15 * nn), tl};      // nn: (first nodes) number of the loop
  auto jw_ld = [&](int k) -> unsigned long long { return jw.get(k); };
  auto jw_st = [&](int k, unsigned long long w) { jw.set(k, w); };
  const bool bm = ab >= 0;
  if (lane == 0) {
    V1B_ADD(32, bm ? 1 : 0);
    V1B_ADD(33, a >= 0 ? 1 : 0);
  }
  unsigned* bmp = (unsigned*)(cx.lds + (bm ? ab : 0));   // bit (x - xmin + 1) of row (y - y_first + 1)
  const int bx0 = xmin - 1, by0 = y_first - 1;
  if (bm) {
    for (int k = lane; k < bw * bh; k += 64) bmp[k] = 0u;
    v1_wave_sync();
  }
  {
    int base = 0;
    for (int yb = y_first; yb <= y_last; yb += 64) {
      const int y = yb + lane;
      int c = 0;
      if (y <= y_last)
        for (int j = rowoff[y]; j < rowoff[y + 1]; j++) c += cx.parent[j] == me;
      const int incl = v1_wave_incl_scan(c, lane);
      int pos = base + incl - c;
      if (y <= y_last)
        for (int j = rowoff[y]; j < rowoff[y + 1]; j++)
          if (cx.parent[j] == me) {
            plist.set(pos, (unsigned short)j);
            prow.set(pos, (unsigned short)y);
            cx.mark[j] = (unsigned short)(pos + 1);   // (for the moment: the piece's place in the list)
            pos++;
            if (bm) {   // (a row is one lane's: plain read-modify-write)
              const short2 q = srun[j];
              unsigned* row = bmp + (y - by0) * bw;
              const int b0 = q.x - bx0, b1 = q.y - bx0;
              for (int w = b0 >> 5; w <= (b1 >> 5); w++) {
                const int lo = w == (b0 >> 5) ? (b0 & 31) : 0, hi = w == (b1 >> 5) ? (b1 & 31) : 31;
                row[w] |= (hi == 31 ? 0xffffffffu : ((2u << hi) - 1u)) & ~((1u << lo) - 1u);
              }
            }
          }
      base += __shfl(incl, 63);
    }
  }
  v1_wave_sync();
  [[maybe_unused]] const long long t1_ = V1B_NOW();
  // (3) a lane per node: its segment
  int bad = 0;
#ifdef V1B_STAMP
  int dbg_steps = 0;
#endif
  for (int m = lane; m < nn; m += 64) {
    [[maybe_unused]] const long long ts0_ = V1B_NOW();
    [[maybe_unused]] long long ts1_ = ts0_, ts2_ = ts0_;
    const int j = plist.get((m >> 1)), side = m & 1, y = prow.get((m >> 1));
    const short2 q = srun[j];
    int cf = 0, nxt = 0xffff;
    const bool crack = !(side == 0 ? (j > rowoff[y] && srun[j - 1].y + 1 == q.x)
                                   : (j + 1 < rowoff[y + 1] && srun[j + 1].x == q.y + 1));   // not cut at a tile seam
    if (crack) {
      int px = side ? q.y : q.x, py = y;
      const int rg = j - rowoff[y] + 1;   // (pieces of adjacent rows of a blob line up: a first guess for all three rows)
      int rel[3] = {rg, rg, rg}, pc[8], top_xe = 0, bot_xs = 0;
      for (int k = 0; k < 8; k++) pc[k] = -1;
      auto hood = [&]() -> int {
        if (!bm) return v1_nbhd(g, px, py, rel, pc, top_xe, bot_xs);
        const int bx = px - bx0;
        const unsigned* r1 = bmp + (py - by0) * bw;
        const int w = (bx - 1) >> 5, sft = (bx - 1) & 31;
        const unsigned long long vt = r1[w - bw] | (unsigned long long)r1[w - bw + 1] << 32;
        const unsigned long long vm = r1[w] | (unsigned long long)r1[w + 1] << 32;
        const unsigned long long vb = r1[w + bw] | (unsigned long long)r1[w + bw + 1] << 32;
        const int top = (int)(vt >> sft) & 7, mid = (int)(vm >> sft) & 7, bot = (int)(vb >> sft) & 7;
        return ((mid >> 2) & 1) | ((top >> 2) & 1) << 1 | ((top >> 1) & 1) << 2 | (top & 1) << 3 | (mid & 1) << 4 |
               (bot & 1) << 5 | ((bot >> 1) & 1) << 6 | ((bot >> 2) & 1) << 7;
      };
      // bitmap rows: the first bit of the wanted kind at or left of / at or right of position p (the margins are empty,
      // so an empty bit is always found; -1 / 1 << 30 when no set bit is)
      auto left_of = [&](const unsigned* row, int p, bool set) -> int {
        int w = p >> 5;
        unsigned mk = (set ? row[w] : ~row[w]) & ((p & 31) == 31 ? 0xffffffffu : ((2u << (p & 31)) - 1u));
        while (!mk) {
          if (--w < 0) return -1;
          mk = set ? row[w] : ~row[w];
        }
        return (w << 5) + 31 - __clz((int)mk);
      };
      auto right_of = [&](const unsigned* row, int p, bool set) -> int {
        int w = p >> 5;
        unsigned mk = (set ? row[w] : ~row[w]) & ~((1u << (p & 31)) - 1u);
        while (!mk) {
          if (++w >= bw) return 1 << 30;
          mk = set ? row[w] : ~row[w];
        }
        return (w << 5) + __ffs((int)mk) - 1;
      };
      int nb = hood();
      const int d0 = side ? 0 : 4;
      // clockwise from the crack: the pixel the walk comes from; a crack whose search passes the pixel's other side
      // (empty too) is swept by the step that starts there
      int tcw = 8;
      for (int t = 7; t >= 1; t--)
        if (nb >> ((d0 - t) & 7) & 1) tcw = t;
      if (tcw == 8) bad = 1;   // a single pixel
      else if (tcw <= 4) {
        int back = (d0 - tcw) & 7, cnt = 0, j3 = j;
        cf = 0x80;
        nxt = -1;
        bool straight = false;   // the last move repeated the one before it
        ts1_ = V1B_NOW();
        for (int step = 0; step < V1F_STEPS; step++) {
#ifdef V1B_STAMP
          dbg_steps++;
#endif
          if (step) {
            // (ONE neighbourhood per pass of this loop, whatever a lane does with it: the lanes of a wave go through
            // every path any of them takes)
            nb = hood();
            // A straight stretch along the top of a run (westwards: NE, N, NW empty, W set) or along its bottom
            // (eastwards: SW, S, SE empty, E set): every step up to the run's end, or to the pixel diagonally below /
            // above the nearest mask pixel of the neighbouring row, is the same -- the same move, nothing emitted,
            // no side swept -- and is taken at once: the pass ends with the lane at the stretch's last pixel
            int xt = px;
            if (straight && back == 0 && (nb & 0x1e) == 0x10) {
              if (bm) {
                const unsigned* r1 = bmp + (py - by0) * bw;
                const int bx = px - bx0, za = left_of(r1, bx - 1, false), sa = left_of(r1 - bw, bx + 1, true);
                xt = (za > sa ? za : sa) + 1 + bx0;
              } else {
                const int xs3 = g.srun[j3].x;
                xt = top_xe + 1 > xs3 ? top_xe + 1 : xs3;
              }
            } else if (straight && back == 4 && (nb & 0xe1) == 0x01) {
              if (bm) {
                const unsigned* r1 = bmp + (py - by0) * bw;
                const int bx = px - bx0, za = right_of(r1, bx + 1, false), sa = right_of(r1 + bw, bx - 1, true);
                xt = (za < sa ? za : sa) - 1 + bx0;
              } else {
                const int xe3 = g.srun[j3].y;
                xt = bot_xs - 1 < xe3 ? bot_xs - 1 : xe3;
              }
            }
            if (xt != px) {
              px = xt;
              straight = false;
              continue;
            }
          }
          const int sh = (back + 1) & 7;
          const int rot = ((nb >> sh) | (nb << (8 - sh))) & 0xff;
          const int z = __ffs(rot) - 1;              // empty neighbours swept before the first set one
          if (z < 0) break;                          // (cannot happen: the way back is set)
          const int sp = (sh + z) & 7;
          const int dw = (4 - sh) & 7, de = (0 - sh) & 7;
          const bool sw = dw < z, se = de < z;
          if (step > 0 && (sw || se)) {
            if (bm) j3 = v1_piece_at(g, px, py);   // (the bitmap walk does not keep track of the pieces)
            nxt = j3 < 0 ? -1 : 2 * ((int)cx.mark[j3] - 1) + ((sw && (!se || dw < de)) ? 0 : 1);
            break;
          }
          if (step == 0 && se) cf |= 0x40;
          const bool turn = sp != (back ^ 4);
          if (turn) {
            if (cnt == V1F_K) break;
            tmp[m * V1F_K + cnt] = make_short2((short)px, (short)py);
            cnt++;
          }
          px += v1_dx(sp);
          py += v1_dy(sp);
          // (selects, not pc[sp]: an array indexed at run time lives in scratch memory)
          j3 = sp == 0 ? pc[0] : sp == 1 ? pc[1] : sp == 2 ? pc[2] : sp == 3 ? pc[3] : sp == 4 ? pc[4] : sp == 5 ? pc[5]
                                                                                                  : sp == 6 ? pc[6] : pc[7];
          back = sp ^ 4;
          straight = !turn;
          const int dyv = v1_dy(sp);
          if (dyv > 0) { rel[0] = rel[1]; rel[1] = rel[2]; } else if (dyv < 0) { rel[2] = rel[1]; rel[1] = rel[0]; }
        }
        ts2_ = V1B_NOW();
        if (nxt < 0) {
          bad = 1;
          nxt = 0xffff;
        }
        cf |= cnt;
      }
    }
    succ.set(m, (unsigned short)nxt);
    cntfl.set(m, (unsigned char)cf);
    st.set(m, 0xffff);
    loop.set(m, 0xff);
#ifdef V1B_STAMP
    {
      long long d1 = ts1_ - ts0_, d2 = ts2_ - ts1_;
      for (int o = 32; o; o >>= 1) { const long long u1 = __shfl_xor(d1, o), u2 = __shfl_xor(d2, o); d1 = d1 > u1 ? d1 : u1; d2 = d2 > u2 ? d2 : u2; }
      if (lane == 0) { V1B_ADD(34, d1); V1B_ADD(35, d2); V1B_ADD(36, V1B_NOW() - ts0_); V1B_ADD(tl ? (bm ? 37 : 38) : 39, d1); V1B_ADD(tl ? (bm ? 3 : 4) : 6, 1); }
      int mx = dbg_steps;
      for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(mx, o); mx = mx > t ? mx : t; }
      V1B_ADD(31, dbg_steps);
      if (lane == 0) V1B_ADD(13, mx);
      dbg_steps = 0;
    }
#endif
  }
  v1_wave_sync();
  [[maybe_unused]] const long long t2_ = V1B_NOW();
  // (4) the loops = the cycles of succ[].  First: succ[] must be a permutation of the nodes (every successor a node,
  // no two nodes with the same one)
  int nvert = 0;
  for (int m = lane; m < nn; m += 64) {
    const int cf = cntfl.get(m);
    if (!(cf & 0x80)) continue;
    const int nx = succ.get(m);
    if (nx >= nn || !(cntfl.get(nx) & 0x80)) bad = 1;
    else st.set(nx, (unsigned short)m);
    nvert += cf & 0x3f;
  }
  v1_wave_sync();
  for (int m = lane; m < nn; m += 64)
    if ((cntfl.get(m) & 0x80) && !bad && st.get(succ.get(m)) != m) bad = 1;
  for (int o = 32; o; o >>= 1) nvert += __shfl_xor(nvert, o);
  const int ms = (cntfl.get(0) & 0x80) ? 0 : 1;   // the outer border starts at the first pixel's west crack
  if (nvert > 65535 || !(cntfl.get(ms) & 0x80)) bad = 1;
  bad = __any(bad);
  int nl = 1, total = 0, fbase = 0, cbase = 0;
  if (!bad) {
    // pointer jumping on {next node, key, vertices to the key's node, vertices of the window}: key = 0 for the outer
    // border's first node, 1 + piece for a node that sweeps an east crack (pieces are in raster order): after
    // log2(nodes) doublings every node knows its loop's first node -- the library starts a hole border at its first
    // east crack in raster order -- and the vertices emitted from itself up to there.  Updated in place: any mix of
    // old and new words is a valid pair of windows.
    for (int m = lane; m < nn; m += 64) {
      const int cf = cntfl.get(m);
      unsigned long long w = (unsigned long long)m | 0xffffull << 16;
      if (cf & 0x80) {
        const unsigned key = m == ms ? 0u : ((cf & 0x40) ? (unsigned)(m >> 1) + 1u : 0xffffu);
        w = (unsigned long long)succ.get(m) | (unsigned long long)key << 16 | (unsigned long long)(cf & 0x3f) << 48;
      }
      jw_st(m, w);
    }
    v1_wave_sync();
    int rounds = 1;
    while ((1 << rounds) < nn) rounds++;
    for (int k = 0; k < rounds; k++) {
      for (int mb = lane; mb < nn; mb += 256) {
        unsigned long long w[4], w2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = mb + 64 * u < nn ? jw_ld(mb + 64 * u) : 0ull;
#pragma unroll
        for (int u = 0; u < 4; u++) w2[u] = jw_ld((int)(w[u] & 0xffff));
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (mb + 64 * u >= nn) continue;
          const unsigned ka = (unsigned)(w[u] >> 16) & 0xffff, kb = (unsigned)(w2[u] >> 16) & 0xffff;
          const unsigned sa = (unsigned)(w[u] >> 48), sb = (unsigned)(w2[u] >> 48);
          const unsigned da = (unsigned)(w[u] >> 32) & 0xffff, db = (unsigned)(w2[u] >> 32) & 0xffff;
          const unsigned key = kb < ka ? kb : ka, dist = kb < ka ? (sa + db) & 0xffff : da;
          jw_st(mb + 64 * u, (w2[u] & 0xffff) | (unsigned long long)key << 16 | (unsigned long long)dist << 32 |
                                 (unsigned long long)((sa + sb) & 0xffff) << 48);
        }
      }
      v1_wave_sync();
    }
    // the loop's first node
    for (int m = lane; m < nn; m += 64) {
      if (!(cntfl.get(m) & 0x80)) continue;
      const unsigned key = (unsigned)(jw_ld(m) >> 16) & 0xffff;
      int s0 = ms;
      if (key == 0xffff) bad = 1;
      else if (key) s0 = (cntfl.get(2 * (key - 1) + 1) & 0x80) ? 2 * (key - 1) + 1 : 2 * (key - 1);
      st.set(m, (unsigned short)s0);
    }
    v1_wave_sync();
    // number the loops (0 = outer, holes in node order), their vertex counts
    for (int mb = 0; mb < nn; mb += 64) {
      const int m = mb + lane;
      const bool first = m < nn && (cntfl.get(m) & 0x80) && st.get(m) == m;
      const bool hole = first && m != ms;
      const unsigned long long bal = __ballot(hole);
      if (first) {
        const int no = hole ? nl + __popcll(bal & ((1ull << lane) - 1ull)) : 0;
        const unsigned long long w = jw_ld(m);
        const int nx = succ.get(m);
        if ((w >> 32) & 0xffff) bad = 1;   // (the first node is the key's node: no vertices before it)
        if (no < V1F_LOOPS) {
          loop.set(m, (unsigned char)no);
          ltab[no * 4] = (cntfl.get(m) & 0x3f) + (nx == m ? 0 : (int)((jw_ld(nx) >> 32) & 0xffff));
          ltab[no * 4 + 1] = (int)((w >> 16) & 0xffff) - 1;
        }
      }
      nl += __popcll(bal);
    }
    if (nl > V1F_LOOPS) bad = 1;
    bad = __any(bad);
  }
  if (!bad) {
    v1_wave_sync();
    const int T = lane < nl ? ltab[lane * 4] : 0;
    const int incl = v1_wave_incl_scan(T, lane);
    if (lane < nl) ltab[lane * 4 + 2] = incl - T;
    total = __shfl(incl, 63);
    if ((lane < nl && T < 1) || total != nvert) bad = 1;
    bad = __any(bad);
  }
  if (!bad) {
    if (lane == 0) {
      fbase = v1_bump(cx.s_vused, total, cx.fast_cap);
      if (fbase >= 0 && nl > 1) cbase = atomicAdd(cx.s_ncont, nl - 1);
    }
    fbase = __shfl(fbase, 0);
    cbase = __shfl(cbase, 0);
    if (fbase < 0) bad = 1;
    else if (nl > 1 && cbase + nl - 1 > cx.kp_cap) {
      // the contour table is full: the result is flagged, the entries taken below the cap only have to be harmless
      for (int c = cbase + lane; c < cx.kp_cap; c += 64) {
        cx.c_disc[c] = 0;
        cx.c_par[c] = (short)i;
        cx.c_link[c] = -1;
        cx.c_hole[c] = 1;
        cx.c_key[c] = 0ull;
      }
      if (lane == 0) *cx.overflow = 1;
      for (int k = lane; k < np; k += 64) cx.mark[plist.get(k)] = 0;
      v1_wave_sync();
      return true;
    }
  }
  if (bad) {
    for (int k = lane; k < np; k += 64) cx.mark[plist.get(k)] = 0;
    v1_wave_sync();
    if (lane == 0) V1B_ADD(18, 1);
    return false;
  }
  v1_wave_sync();
  [[maybe_unused]] const long long t3_ = V1B_NOW();
  // (5) vertices to their places (a lane reads back what it wrote itself), marks, contour entries
  for (int m = lane; m < nn; m += 64) {
    const int cf = cntfl.get(m);
    if (!(cf & 0x80)) continue;
    const int s0 = st.get(m), l = loop.get(s0);
    const int first = m == s0 ? 0 : ltab[l * 4] - (int)((jw_ld(m) >> 32) & 0xffff);
    short2* dst = cx.vpool + fbase + ltab[l * 4 + 2] + first;
    for (int k = 0; k < (cf & 0x3f); k++) dst[k] = tmp[m * V1F_K + k];
  }
  for (int k = lane; k < np; k += 64) {
    const int e = cntfl.get(2 * k + 1), w = cntfl.get(2 * k);
    int l = -1;
    if (e & 0x80) l = loop.get(st.get(2 * k + 1));
    else if ((w & 0xc0) == 0xc0) l = loop.get(st.get(2 * k));
    cx.mark[plist.get(k)] = (unsigned short)(l < 0 ? 0 : (l == 0 ? i : cbase + l - 1) + 1);
  }
  for (int l = 1 + lane; l < nl; l += 64) {
    const int c = cbase + l - 1, k = ltab[l * 4 + 1];
    cx.c_disc[c] = (int)prow.get(k) * cx.W + srun[plist.get(k)].y;
    cx.c_par[c] = (short)i;
    cx.c_link[c] = -1;
    cx.c_hole[c] = 1;
    cx.c_key[c] = 0ull;
  }
  v1_wave_sync();
  [[maybe_unused]] const long long t4_ = V1B_NOW();
  // (6) the fills, the whole wave on one loop after the other
  for (int l = 0; l < nl; l++) {
    const short2* v = cx.vpool + fbase + ltab[l * 4 + 2];
    const int n = ltab[l * 4];
    pt.cur = 0ull;
    if (v1_fill_wave(pt, v, n, lane, cx)) v1_fill_paint<64, false>(pt, v, n, lane);
    else v1_fill_paint<64, true>(pt, v, n, lane);
    if (lane == 0) cx.c_key[l == 0 ? i : cbase + l - 1] = pt.cur;
  }
  if (lane == 0) {
    V1B_ADD(16, 1);
    V1B_ADD(19, t1_ - t0_);
    V1B_ADD(20, t2_ - t1_);
    V1B_ADD(21, t3_ - t2_);
    V1B_ADD(22, t4_ - t3_);
    V1B_ADD(24, V1B_NOW() - t4_);
    V1B_ADD(25, nl);
    V1B_ADD(26, total);
  }
  return true;
}

// ---------------------------------------------------------------------------
// K-B: blobs from runs.  grid (n_parts, B), block V1B_THREADS (8 waves: a (frame, part) plane of a crowded 1080p
// scene has 5-6 blobs that need the literal walk, and walks that share a wave run one after the other)
// ---------------------------------------------------------------------------
#define V1B_THREADS 512
__device__ __forceinline__ int uf_root(const int* parent, int i) {
  // (relaxed atomic loads, not volatile ones: other threads hook roots meanwhile, and a volatile access keeps the
  // pointer generic -- flat loads -- where the table is known to be in LDS)
  int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (p != i) {
    i = p;
    p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  return i;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_root(parent, a);
    b = uf_root(parent, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }  // a > b: hook a under b
    const int old = atomicCAS(&parent[a], a, b);
    if (old == a) return;
  }
}

__global__ __launch_bounds__(V1B_THREADS) void v1_blobs_kernel(const float* __restrict__ net_out, int C, int hs, int ws, int W,
                                                       int H, int n_parts, int run_cap, int kp_cap, int vert_cap,
                                                       const int* __restrict__ n_runs, const V1Run* __restrict__ runs,
                                                       short2* __restrict__ ws_verts,
                                                       float* __restrict__ kps, int* __restrict__ n_kps,
                                                       int* __restrict__ overflow, int lds_bytes, int fast) {
  extern __shared__ __attribute__((aligned(16))) int ismem[];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const long long tb0_ = V1B_NOW();
  (void)tb0_;
  int n = n_runs[(size_t)b * n_parts + part];
  n = n > run_cap ? run_cap : n;
  float* out = kps + ((size_t)b * n_parts + part) * kp_cap * 3;
  int* out_n = n_kps + (size_t)b * n_parts + part;
  if (n == 0) {
    if (tid == 0) *out_n = 0;
    return;
  }
  const V1Run* R = runs + ((size_t)b * n_parts + part) * run_cap;
  short2* vpool = ws_verts + ((size_t)b * n_parts + part) * vert_cap;
  // LDS: the mask as pieces sorted by (row, xs); every table below is indexed by sorted position.  The tables are
  // laid out for the n pieces this plane has, not for run_cap: what is left of the block's LDS (sized for run_cap by
  // the launcher) is the fast walk's table space
  const int nn = (n + 3) & ~3;
  int* parent = ismem;                                          // n: component links, then each piece's blob root
  int* rowoff = parent + nn;                                    // H + 2
  int* roots = rowoff + H + 2;                                  // kp_cap: the blobs' root pieces
  float* bval = (float*)(roots + kp_cap);                       // n: the piece's maximum
  short2* srun = (short2*)(bval + nn);                          // n: {xs, xe}
  unsigned short* byrow = (unsigned short*)(srun + nn);         // n: index in R
  unsigned short* mark = byrow + nn;                            // n: contour id + 1 of the trace that
                                                                //    right-marked the piece's last pixel
  // (offsets from the LDS base, not integer casts of pointers: the tables then stay LDS pointers for the compiler --
  // through a cast they became generic ones, read with flat loads at about twice the latency)
  auto up = [&](const void* p_, size_t a) {
    const size_t off = (size_t)((const char*)p_ - (const char*)ismem);
    return (char*)ismem + ((off + a - 1) & ~(a - 1));
  };
  unsigned long long* c_key = (unsigned long long*)up(mark + nn, 8);  // per contour: arg-max key
  int* c_disc = (int*)(c_key + kp_cap);          // discovery position (raster index of the border's start)
  short* c_par = (short*)(c_disc + kp_cap);      // parent contour, -1 = the frame
  short* c_link = c_par + kp_cap;                // outer borders: the border met last before their first pixel
  short* c_depth = c_link + kp_cap;
  unsigned short* irr = (unsigned short*)(c_depth + kp_cap);   // blobs that need the literal walk
  unsigned char* c_hole = (unsigned char*)(irr + kp_cap);
  // per blob (position in roots[]): first-raster maximum, first pixel, last row, number of (seam-merged) runs
  unsigned long long* b_best = (unsigned long long*)up(c_hole + kp_cap, 8);
  int* b_first = (int*)(b_best + kp_cap);
  int* b_ymax = b_first + kp_cap;
  int* b_nrun = b_ymax + kp_cap;
  __shared__ int s_nroot, s_nirr, s_ncont, s_vused, s_free;
  __shared__ int s_wsum[V1B_THREADS / 64];
  __shared__ int s_ltab[V1B_THREADS / 64][V1F_LOOPS * 4];

  for (int i = tid; i < H + 2; i += V1B_THREADS) rowoff[i] = 0;
  if (tid == 0) {
    s_nroot = 0;
    s_nirr = 0;
    s_vused = 0;
    s_free = (int)((const char*)up(b_nrun + kp_cap, 16) - (const char*)ismem);
  }
  __syncthreads();
  for (int i = tid; i < n; i += V1B_THREADS) {
    parent[i] = i;
    mark[i] = 0;
    atomicAdd(&rowoff[R[i].y + 1], 1);
  }
  __syncthreads();
  // exclusive scan of the per-row counts (rowoff[y+1] = count of row y)
  {
    const int per = (H + 1 + V1B_THREADS - 1) / V1B_THREADS;
    const int lo = tid * per, hi = lo + per < H + 1 ? lo + per : H + 1;
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += rowoff[i + 1];
    int v = sum;   // inclusive scan over the block: inside the wave by shuffles, wave totals through LDS
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o);
      if ((tid & 63) >= o) v += t;
    }
    if ((tid & 63) == 63) s_wsum[tid >> 6] = v;
    __syncthreads();
    int acc = v - sum;
    for (int w = 0; w < (tid >> 6); w++) acc += s_wsum[w];
    for (int i = lo; i < hi; i++) {
      const int t = rowoff[i + 1];
      rowoff[i + 1] = acc;   // start offset of row i
      acc += t;
    }
  }
  __syncthreads();
  // rowoff[y+1] now = start of row y; scatter with a moving cursor: afterwards rowoff[y+1] = end of row y = start
  // of row y+1, so row y spans [rowoff[y], rowoff[y+1]) with rowoff[0] = 0 (row -1 is empty)
  for (int i = tid; i < n; i += V1B_THREADS) {
    const V1Run r = R[i];
    const int pos = atomicAdd(&rowoff[r.y + 1], 1);
    byrow[pos] = (unsigned short)i;
    srun[pos] = make_short2((short)r.xs, (short)r.xe);
    bval[pos] = r.best;
    mark[pos] = (unsigned short)r.y;   // (the row, for the component search; the sort below moves pieces inside a row only)
  }
  __syncthreads();
  if (tid == 0) V1B_ADD(27, V1B_NOW() - tb0_);
  for (int y = tid; y < H; y += V1B_THREADS) {   // order each row by xs (a handful of pieces: insertion sort)
    const int lo = rowoff[y], hi = rowoff[y + 1];
    for (int a = lo + 1; a < hi; a++) {
      const short2 sv = srun[a];
      const unsigned short bv = byrow[a];
      const float fv = bval[a];
      int q = a - 1;
      while (q >= lo && srun[q].x > sv.x) {
        srun[q + 1] = srun[q];
        byrow[q + 1] = byrow[q];
        bval[q + 1] = bval[q];
        q--;
      }
      srun[q + 1] = sv;
      byrow[q + 1] = bv;
      bval[q + 1] = fv;
    }
  }
  __syncthreads();
  if (tid == 0) V1B_ADD(28, V1B_NOW() - tb0_);
  // Connected components of the pieces (8-connectivity: a piece touches the pieces of the previous row that reach
  // into [xs - 1, xe + 1], and its left neighbour across a tile seam).  Pieces are numbered in raster order, so every
  // piece first hangs under the first piece it touches above or to its left; pointer jumping then makes every piece
  // point at its tree's root (a blob's height halves per round); a union-find pass over the touching pairs then
  // merges the trees that meet further down (a bump is one tree from the start; a U or an M is not), and a second
  // jumping pass flattens what that built.  A union-find from single pieces, with finds from every piece, took 300 k
  // cycles here on a crowded 1080p plane: it climbs a piece per row.
  auto above = [&](int y, const short2 r, int& k1) {   // first piece of row y - 1 that ends at or after xs - 1
    int k = rowoff[y - 1];
    k1 = rowoff[y];
    for (int hi = k1; hi - k > 4;) {
      const int mid = (k + hi) >> 1;
      if (srun[mid].y < r.x - 1) k = mid + 1;
      else hi = mid;
    }
    while (k < k1 && srun[k].y < r.x - 1) k++;
    return k;
  };
  for (int j = tid; j < n; j += V1B_THREADS) {
    const int y = mark[j];   // (the row, left there by the scatter; the marks are cleared below)
    const short2 r = srun[j];
    int link = j;
    if (j > rowoff[y] && srun[j - 1].y + 1 == r.x) link = j - 1;
    if (y > 0) {
      int k1;
      const int k = above(y, r, k1);
      if (k < k1 && srun[k].x <= r.y + 1) link = k;
    }
    parent[j] = link;
  }
  __syncthreads();
  auto jump = [&]() {   // pointer jumping to the roots, four pieces of a thread at a time
    for (;;) {
      int ch = 0;
      for (int j0 = tid; j0 < n; j0 += 4 * V1B_THREADS) {
        int p[4], gp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = j0 + u * V1B_THREADS < n ? parent[j0 + u * V1B_THREADS] : 0;
#pragma unroll
        for (int u = 0; u < 4; u++) gp[u] = parent[p[u]];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (j0 + u * V1B_THREADS < n && gp[u] != p[u]) {
            parent[j0 + u * V1B_THREADS] = gp[u];
            ch = 1;
          }
      }
      if (!__syncthreads_or(ch)) break;
    }
  };
  jump();
  // the trees are flat now: a union per touching pair is two short finds and, when the trees differ, one hook
  for (int j = tid; j < n; j += V1B_THREADS) {
    const int y = mark[j];
    const short2 r = srun[j];
    if (j > rowoff[y] && srun[j - 1].y + 1 == r.x) uf_union(parent, j, j - 1);
    if (y > 0) {
      int k1;
      for (int k = above(y, r, k1); k < k1 && srun[k].x <= r.y + 1; k++) uf_union(parent, j, k);
    }
  }
  __syncthreads();
  jump();
  for (int j = tid; j < n; j += V1B_THREADS) mark[j] = 0;
  if (tid == 0) V1B_ADD(29, V1B_NOW() - tb0_);
  // the blobs = the roots, in any order; a root's piece carries its blob's number (complemented) for a moment so
  // that every piece finds it in one step
  for (int j = tid; j < n; j += V1B_THREADS)
    if (parent[j] == j) {
      const int s = atomicAdd(&s_nroot, 1);
      if (s < kp_cap) roots[s] = j; else *overflow = 1;
    }
  __syncthreads();
  if (tid == 0) V1B_ADD(30, V1B_NOW() - tb0_);
  const int nr = s_nroot < kp_cap ? s_nroot : kp_cap;
  for (int i = tid; i < nr; i += V1B_THREADS) {
    b_best[i] = 0ull;
    b_first[i] = 0x7fffffff;
    b_ymax[i] = -1;
    b_nrun[i] = 0;
    parent[roots[i]] = ~i;
  }
  __syncthreads();
  // per-blob first-raster maximum, first pixel, last row and number of (seam-merged) runs: LDS atomics
  for (int j = tid; j < n; j += V1B_THREADS) {
    int id = parent[j];
    if (id >= 0) {
      id = parent[id];
      if (id >= 0) continue;   // a blob beyond kp_cap (overflow is set)
    }
    id = ~id;
    const V1Run r = R[byrow[j]];
    const unsigned pos = (unsigned)(r.y * W + r.bx);
    atomicMax(&b_best[id], ((unsigned long long)__float_as_uint(r.best) << 32) | (0xffffffffu - pos));
    atomicMin(&b_first[id], r.y * W + r.xs);
    atomicMax(&b_ymax[id], r.y);
    if (!(j > rowoff[r.y] && srun[j - 1].y + 1 == r.xs)) atomicAdd(&b_nrun[id], 1);
  }
  __syncthreads();
  for (int i = tid; i < nr; i += V1B_THREADS) parent[roots[i]] = roots[i];
  const float* plane = net_out + ((size_t)b * C + part) * hs * ws;
  // the contour table starts with the blobs' outer borders; blobs with more than one run in some row need the
  // literal contour + fill (see the helpers' header), every other blob's keypoint is its component's arg-max
  if (tid == 0) s_ncont = nr;
  for (int i = tid; i < nr; i += V1B_THREADS) {
    const int fp = b_first[i];
    c_disc[i] = fp;
    c_par[i] = -1;
    c_link[i] = -1;
    c_hole[i] = 0;
    c_key[i] = b_best[i];
    if (b_nrun[i] != b_ymax[i] - fp / W + 1) irr[atomicAdd(&s_nirr, 1)] = (unsigned short)i;
  }
  __syncthreads();
  const int nirr = s_nirr;
  if (nirr > 1) {
    // the largest blobs first: they are the kernel's critical path, and the first to come get the fast walk's tables
    for (int t = tid; t < nirr; t += V1B_THREADS) {
      const int it = irr[t], kt = b_nrun[it];
      int rank = 0;
      for (int u = 0; u < nirr; u++) {
        const int ku = b_nrun[irr[u]];
        rank += ku > kt || (ku == kt && u < t);
      }
      c_depth[rank] = (short)it;
    }
    __syncthreads();
    for (int t = tid; t < nirr; t += V1B_THREADS) irr[t] = (unsigned short)c_depth[t];
    __syncthreads();
  }
  [[maybe_unused]] const long long tw0_ = V1B_NOW();
  if (tid == 0) {
    V1B_ADD(7, tw0_ - tb0_);
    V1B_ADD(5, nirr);
    V1B_ADD(9, 1);
    V1B_ADD(10, n);
  }
  if (nirr) {
    const V1Grid g = {srun, rowoff, W, H};
    V1Paint pt = {g, bval, byrow, R, plane, hs, ws, 0ull};
    V1Verts vv = {vpool, &s_vused, vert_cap, 0, 0};
    const int lane = tid & 63, r = tid & 7;
    // one group of 8 lanes per blob.  Groups of one wave that walk different blobs run one after the other (their
    // control flow differs), so consecutive blobs go to different WAVES first: blob k -> wave k % NW, group k / NW % 8
    constexpr int NW = V1B_THREADS / 64;
    // first a WAVE per blob: all its border loops at once (v1_fast_blob); the blobs that leaves (bit 15 of their
    // entry) go through the one-group walk below
    {
      const V1FastCtx cx = {parent, mark, (char*)ismem, &s_free, lds_bytes, vpool, &s_vused, vert_cap,
                            vert_cap - 8 * run_cap, s_ltab[tid >> 6], W, kp_cap,
                            &s_ncont, c_disc, c_par, c_link, c_hole, c_key, overflow};
      for (int k = tid >> 6; k < nirr; k += NW) {
        const int i = irr[k];
        const int fp = c_disc[i];
        [[maybe_unused]] const long long ta_ = V1B_NOW();
        const bool done = fast && v1_fast_blob(g, pt, cx, i, roots[i], fp / W, b_ymax[i], lane);
        if (lane == 0) {
          if (!done) irr[k] = (unsigned short)(i | 0x8000);
          atomicAdd(overflow + (done ? 1 : 2), 1);
          V1B_ADD(2, V1B_NOW() - ta_);
        }
      }
    }
    __syncthreads();
    for (int k = ((tid >> 3) & 7) * NW + (tid >> 6); k < nirr; k += 8 * NW) {
      if (!(irr[k] & 0x8000)) continue;
      const int i = irr[k] & 0x7fff, me = roots[i];
      const int fp = c_disc[i], y_first = fp / W, y_last = b_ymax[i];
      int c = i, sx = fp - y_first * W, sy = y_first;   // the outer border starts at the blob's first raster pixel
      bool hole = false;
      for (;;) {
        [[maybe_unused]] const long long ta_ = V1B_NOW();
        const int nv = v1_trace(g, sx, sy, hole, (unsigned short)(c + 1), vv, mark, lane);
        if (nv <= 0) {
          *overflow = 1;
          break;
        }
        [[maybe_unused]] const long long tb_ = V1B_NOW();
        pt.cur = 0ull;
        v1_fill_paint<8>(pt, vpool + vv.base, nv, lane);
        if (r == 0) c_key[c] = pt.cur;
        [[maybe_unused]] const long long tc_ = V1B_NOW();
        if (r == 0) {
          V1B_ADD(0, tb_ - ta_);
          V1B_ADD(1, tc_ - tb_);
          V1B_ADD(6, 1);
          V1B_ADD(11, nv);
        }
        // next hole border: the blob's first run end in raster order, from here on, that no trace has right-marked
        // (lane k looks at the k-th row of each 8)
        int hx = -1, hy = -1;
        for (int yb = sy; yb <= y_last && hx < 0; yb += 8) {
          const int y = yb + r;
          int fx = -1;
          if (y <= y_last)
            for (int j = rowoff[y]; j < rowoff[y + 1]; j++) {
              const short2 q = srun[j];
              if ((hole && y == sy && q.y <= sx) || mark[j] != 0 || parent[j] != me) continue;
              if (j + 1 < rowoff[y + 1] && srun[j + 1].x == q.y + 1) continue;   // cut at a tile seam: not the run's end
              fx = q.y;
              break;
            }
          const unsigned hit = v1_gballot(fx >= 0, lane);
          if (hit) {
            const int k0 = __ffs(hit) - 1;
            hx = __shfl(fx, k0, 8);
            hy = yb + k0;
          }
        }
        if (hx < 0) break;
        if (r == 0) c = atomicAdd(&s_ncont, 1);
        c = __shfl(c, 0, 8);
        if (c >= kp_cap) {
          *overflow = 1;
          break;
        }
        if (r == 0) {
          c_disc[c] = hy * W + hx;   // (discovered one pixel further right; same order among holes)
          c_par[c] = (short)i;
          c_link[c] = -1;
          c_hole[c] = 1;
          c_key[c] = 0ull;
        }
        sx = hx;
        sy = hy;
        hole = true;
      }
    }
    __syncthreads();
  }
  if (tid == 0) V1B_ADD(8, V1B_NOW() - tw0_);
  const int nc = s_ncont < kp_cap ? s_ncont : kp_cap;
  if (nc > nr) {
    // parents of the outer borders (RETR_TREE): the border met last on the first pixel's row = the one that
    // right-marked the nearest run end to its left; a hole border is the parent itself, an outer border hands on
    // its own parent (its first pixel comes earlier in raster order, so the chain ends)
    for (int X = tid; X < nr; X += V1B_THREADS) {
      const int fp = c_disc[X], y0 = fp / W, x0 = fp - y0 * W;
      const int j0 = v1_piece_at({srun, rowoff, W, H}, x0, y0);
      int link = -1;
      if (j0 > rowoff[y0]) {
        link = (int)mark[j0 - 1] - 1;
        if (link < 0) {   // a simple blob: its only border is its outer one
          const int rt = parent[j0 - 1];
          for (int q = 0; q < nr; q++)
            if (roots[q] == rt) link = q;
        }
      }
      c_link[X] = (short)link;
    }
    __syncthreads();
    for (int X = tid; X < nr; X += V1B_THREADS) {
      int p = c_link[X];
      while (p >= 0 && !c_hole[p]) p = c_link[p];
      c_par[X] = (short)p;
    }
    __syncthreads();
    for (int X = tid; X < nc; X += V1B_THREADS) {
      int d = 0;
      for (int p = c_par[X]; p >= 0; p = c_par[p]) d++;
      c_depth[X] = (short)d;
    }
    __syncthreads();
  }
  // list order of cvTreeToNodeSeq: a node, its children, its next sibling; siblings latest-discovered first.  A
  // contour's place = the number of contours that come before it.
  for (int a = tid; a < nc; a += V1B_THREADS) {
    int rank = 0;
    if (nc == nr) {
      const int da = c_disc[a];
      for (int q = 0; q < nc; q++) rank += c_disc[q] > da;
    } else {
      for (int q = 0; q < nc; q++) {
        if (q == a) continue;
        int x = a, y = q, dx = c_depth[a], dy = c_depth[q];
        while (dx > dy) {
          x = c_par[x];
          dx--;
        }
        while (dy > dx) {
          y = c_par[y];
          dy--;
        }
        if (x == y) {   // one is the other's ancestor: the ancestor comes first
          rank += c_depth[q] < c_depth[a];
          continue;
        }
        while (c_par[x] != c_par[y]) {
          x = c_par[x];
          y = c_par[y];
        }
        rank += c_disc[y] > c_disc[x];
      }
    }
    const unsigned long long key = c_key[a];
    const unsigned pos = key ? 0xffffffffu - (unsigned)(key & 0xffffffffu) : 0u;   // (key 0: a walk that overflowed)
    const int y = pos / W, x = pos - y * W;
    out[rank * 3 + 0] = (float)x;
    out[rank * 3 + 1] = (float)y;
    out[rank * 3 + 2] = resize_at(plane, hs, ws, W, H, x, y);
  }
  if (tid == 0) *out_n = nc;
  if (tid == 0) V1B_ADD(12, V1B_NOW() - tb0_);
}

// ---------------------------------------------------------------------------
// K-C: getValidPairs.  grid (n_limbs, B), block 256
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void v1_pairs_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                       int W, int H, V1Model mdl, int kp_cap,
                                                       const float* __restrict__ kps, const int* __restrict__ n_kps,
                                                       double* __restrict__ pairs, int* __restrict__ n_pairs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int pa = mdl.pair_parts[2 * k], pb = mdl.pair_parts[2 * k + 1];
  const int* cnt = n_kps + (size_t)b * mdl.n_parts;
  const int nA = cnt[pa], nB = cnt[pb];
  int* out_n = n_pairs + (size_t)b * mdl.n_limbs + k;
  if (nA == 0 || nB == 0) {
    if (tid == 0) *out_n = -1;  // "No Connection": invalid pair (utils.py:126-129)
    return;
  }
  int idA0 = 0, idB0 = 0;  // global keypoint ids: running count over parts (skeleton.py:130-133)
  for (int p = 0; p < pa; p++) idA0 += cnt[p];
  for (int p = 0; p < pb; p++) idB0 += cnt[p];
  float* s_pa = smem;
  float* s_pb = s_pa + hs * ws;
  const float* gA = net_out + ((size_t)b * C + mdl.mapidx[2 * k]) * hs * ws;
  const float* gB = net_out + ((size_t)b * C + mdl.mapidx[2 * k + 1]) * hs * ws;
  for (int i = tid; i < hs * ws; i += 256) {
    s_pa[i] = gA[i];
    s_pb[i] = gB[i];
  }
  __shared__ int s_wsum[4];
  __shared__ int s_base;
  if (tid == 0) s_base = 0;
  __syncthreads();
  const float* A = kps + ((size_t)b * mdl.n_parts + pa) * kp_cap * 3;
  const float* Bk = kps + ((size_t)b * mdl.n_parts + pb) * kp_cap * 3;
  double* out = pairs + ((size_t)b * mdl.n_limbs + k) * kp_cap * 3;
  for (int i0 = 0; i0 < nA; i0 += 256) {
    const int i = i0 + tid;
    int max_j = -1;
    double max_score = -1.0;
    if (i < nA) {
      const int ax = (int)A[i * 3], ay = (int)A[i * 3 + 1];
      for (int j = 0; j < nB; j++) {
        const int bx = (int)Bk[j * 3], by = (int)Bk[j * 3 + 1];
        const double dx = (double)(bx - ax), dy = (double)(by - ay);
        const double norm = sqrt(dx * dx + dy * dy);
        if (norm == 0.0) continue;
        const double d0 = dx / norm, d1 = dy / norm;
        const double stepx = dx / 9.0, stepy = dy / 9.0;
        double total = 0.0;
        int above = 0;
        for (int s = 0; s < 10; s++) {
          const double cx = s == 9 ? (double)bx : (double)s * stepx + (double)ax;
          const double cy = s == 9 ? (double)by : (double)s * stepy + (double)ay;
          const int xi = (int)rint(cx), yi = (int)rint(cy);
          const double va = (double)resize_at(s_pa, hs, ws, W, H, xi, yi);
          const double vb = (double)resize_at(s_pb, hs, ws, W, H, xi, yi);
          const double sc = va * d0 + vb * d1;
          total = total + sc;
          above += sc > 0.1;
        }
        const double avg = total / 10.0;
        if ((double)above / 10.0 > 0.5 && avg > max_score) {
          max_j = j;
          max_score = avg;
        }
      }
    }
    // ordered compaction of the A's that found a partner
    const bool found = max_j >= 0;
    const unsigned long long m = __ballot(found);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) s_wsum[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wsum[w];
    if (found) {
      const int o = off + __popcll(m & ((1ull << lane) - 1ull));
      out[o * 3 + 0] = (double)(idA0 + i);
      out[o * 3 + 1] = (double)(idB0 + max_j);
      out[o * 3 + 2] = max_score;
    }
    __syncthreads();
    if (tid == 0) s_base += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    __syncthreads();
  }
  if (tid == 0) *out_n = s_base;
}

// ---------------------------------------------------------------------------
// K-D: getPersonwiseKeypoints.  grid (B), block 64
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void v1_assemble_kernel(V1Model mdl, int kp_cap, int person_cap,
                                                         const float* __restrict__ kps,
                                                         const int* __restrict__ n_kps,
                                                         const double* __restrict__ pairs,
                                                         const int* __restrict__ n_pairs,
                                                         double* __restrict__ persons, int* __restrict__ n_persons,
                                                         int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int RW = mdl.n_parts + 1;  // len(body_parts): part slots + total score
  double* rows = dsm;              // person_cap * RW
  __shared__ int s_n;
  __shared__ int s_off[V1_MAX_PARTS + 1];
  if (lane == 0) {
    s_n = 0;
    int acc = 0;
    for (int p = 0; p < mdl.n_parts; p++) {
      s_off[p] = acc;
      acc += n_kps[(size_t)b * mdl.n_parts + p];
    }
    s_off[mdl.n_parts] = acc;
  }
  __syncthreads();
  auto kp_score = [&](int id) -> double {  // keypoints_list[id, 2]
    int p = 0;
    while (p + 1 < mdl.n_parts && id >= s_off[p + 1]) p++;
    return (double)kps[((size_t)b * mdl.n_parts + p) * kp_cap * 3 + (id - s_off[p]) * 3 + 2];
  };
  // The reference consumes the connections of a limb one after the other, but within one limb they do not depend on
  // each other: a connection looks for the FIRST row whose part-A slot holds its A keypoint, the A keypoints of a
  // limb's connections are all different (getValidPairs keeps one B per A), this limb only writes the part-B slot
  // and the score (at most once per row), and a row it appends holds an A no other connection has.  So 64
  // connections at a time: every lane scans the rows for its own A (the same row entry for all lanes: one broadcast
  // read per row), the hits update their rows, the misses append theirs in connection order (ballot prefix).
  for (int k = 0; k < mdl.n_limbs; k++) {
    const int nc = n_pairs[(size_t)b * mdl.n_limbs + k];
    if (nc < 0) continue;  // invalid pair
    const int ia = mdl.pair_parts[2 * k], ib = mdl.pair_parts[2 * k + 1];
    const double* vp = pairs + ((size_t)b * mdl.n_limbs + k) * kp_cap * 3;
    for (int i0 = 0; i0 < nc; i0 += 64) {
      const bool act = i0 + lane < nc;
      double partA = -2.0, partB = 0.0, sc = 0.0, ka = 0.0, kb = 0.0;
      if (act) {
        partA = vp[(i0 + lane) * 3];
        partB = vp[(i0 + lane) * 3 + 1];
        sc = vp[(i0 + lane) * 3 + 2];
        ka = kp_score((int)partA);
        kb = kp_score((int)partB);
      }
      const int n = s_n;
      int hit = -1;
      for (int r = 0; r < n; r++) {
        const double v = rows[r * RW + ia];
        if (hit < 0 && v == partA) hit = r;
      }
      const bool create = act && hit < 0 && k < mdl.n_parts + 1 - 2;
      const unsigned long long bal = __ballot(create);
      const int idx = n + __popcll(bal & ((1ull << lane) - 1ull)), total = __popcll(bal);
      if (act && hit >= 0) {
        rows[hit * RW + ib] = partB;
        rows[hit * RW + RW - 1] = rows[hit * RW + RW - 1] + (kb + sc);
      }
      if (create) {
        if (idx < person_cap) {
          double* row = rows + idx * RW;
          for (int q = 0; q < RW; q++) row[q] = -1.0;
          row[ia] = partA;
          row[ib] = partB;
          row[RW - 1] = (ka + kb) + sc;
        } else {
          *overflow = 1;
        }
      }
      __syncthreads();   // (one wave: orders the LDS writes above before the next chunk's reads)
      if (lane == 0) s_n = n + total < person_cap ? n + total : person_cap;
      __syncthreads();
    }
  }
  const int n = s_n;
  double* out = persons + (size_t)b * person_cap * RW;
  for (int i = lane; i < n * RW; i += 64) out[i] = rows[i];
  if (lane == 0) n_persons[b] = n;
}

// ---------------------------------------------------------------------------
// host launcher
// ---------------------------------------------------------------------------
int ph_v1_sizes(const V1Model& m, int B, int W, int H, const V1Caps& c, V1Sizes* s) {
  s->n_runs = (size_t)B * m.n_parts * sizeof(int);
  s->runs = (size_t)B * m.n_parts * c.run_cap * sizeof(V1Run);
  s->ws_verts = (size_t)B * m.n_parts * V1_VERT_POOL(c.run_cap) * sizeof(short2);   // contour vertices
  s->kps = (size_t)B * m.n_parts * c.kp_cap * 3 * sizeof(float);
  s->n_kps = (size_t)B * m.n_parts * sizeof(int);
  s->pairs = (size_t)B * m.n_limbs * c.kp_cap * 3 * sizeof(double);
  s->n_pairs = (size_t)B * m.n_limbs * sizeof(int);
  s->persons = (size_t)B * c.person_cap * (m.n_parts + 1) * sizeof(double);
  s->n_persons = (size_t)B * sizeof(int);
  return 0;
}

int ph_launch_decode_v1(const float* net_out, int C, int hs, int ws, int W, int H, float thr, const V1Model& m,
                        int B, const V1Caps& c, const V1Buffers& w, hipStream_t stream, hipEvent_t* ev) {
  if (m.n_parts < 1 || m.n_parts > V1_MAX_PARTS || m.n_limbs < 1 || m.n_limbs > V1_MAX_LIMBS || m.n_parts > C) {
    ph_set_error("decode_v1: bad model (%d parts, %d limbs, %d channels)", m.n_parts, m.n_limbs, C);
    return -1;
  }
  for (int k = 0; k < 2 * m.n_limbs; k++)
    if (m.mapidx[k] < 0 || m.mapidx[k] >= C || m.pair_parts[k] < 0 || m.pair_parts[k] >= m.n_parts) {
      ph_set_error("decode_v1: limb table entry %d out of range", k);
      return -1;
    }
  if (!(thr >= 0.f)) {   // the arg-max keys order values by their bit patterns: mask pixels must be positive
    ph_set_error("decode_v1: conf_thres %g is negative (or NaN); the blob search needs a threshold >= 0", (double)thr);
    return -1;
  }
  if (c.run_cap > 65535 || c.kp_cap > 32767) {
    ph_set_error("decode_v1: capacities beyond the 16-bit tables (%d runs, %d blobs)", c.run_cap, c.kp_cap);
    return -1;
  }
  if (W < 2 || H < 2 || W > 8192 || H > 8192 || (long long)W * H >= 0x7fffffffLL) {
    ph_set_error("decode_v1: bad image size %dx%d", W, H);
    return -1;
  }
  const size_t ldsA = ((size_t)hs * ws + (size_t)(V1_TILE_H + 2) * (V1_TILE_W + 2) +
                       (size_t)(V1_TILE_H + 2) * V1_TILE_W + (size_t)V1_TILE_H * V1_TILE_W) * 4;
  // K-B: the sorted pieces (root, maximum, extents, run index, mark), row offsets, blob and contour tables
  const size_t ldsB = (size_t)((c.run_cap + 3) & ~3) * (4 + 4 + 4 + 2 + 2) + ((size_t)H + 2 + c.kp_cap) * 4 + 8 +
                      (size_t)c.kp_cap * (8 + 4 + 2 + 2 + 2 + 2 + 1) + 16 + (size_t)c.kp_cap * (8 + 4 + 4 + 4) + 8;
  const size_t ldsC = (size_t)2 * hs * ws * 4;
  const size_t ldsD = (size_t)c.person_cap * (m.n_parts + 1) * 8;
  const size_t lim = 160 * 1024 - 8192;  // static __shared__ tables sit beside the dynamic region
  if (hs > 256) {
    ph_set_error("decode_v1: net output height %d beyond the run kernel's 256 source rows", hs);
    return -1;
  }
  if (ldsA > lim || ldsB > lim || ldsC > lim || ldsD > lim) {
    ph_set_error("decode_v1: LDS budget exceeded (%zu/%zu/%zu/%zu B)", ldsA, ldsB, ldsC, ldsD);
    return -1;
  }
  static int attr[4][PH_MAX_DEVICES];
  if (ph_ensure_lds((const void*)v1_runs_kernel, (int)lim, attr[0]) ||
      ph_ensure_lds((const void*)v1_blobs_kernel, (int)lim, attr[1]) ||
      ph_ensure_lds((const void*)v1_pairs_kernel, (int)lim, attr[2]) ||
      ph_ensure_lds((const void*)v1_assemble_kernel, (int)lim, attr[3]))
    return -1;
  V1Sizes sz;
  ph_v1_sizes(m, B, W, H, c, &sz);
  PH_CHECK(hipMemsetAsync(w.n_runs, 0, sz.n_runs, stream));
  PH_CHECK(hipMemsetAsync(w.overflow, 0, 4 * sizeof(int), stream));   // flag + the two walk counters
  PH_CHECK(hipMemsetAsync(w.kps, 0, sz.kps, stream));
  const int tiles = ((W + V1_TILE_W - 1) / V1_TILE_W) * ((H + V1_TILE_H - 1) / V1_TILE_H);
  if (ev) PH_CHECK(hipEventRecord(ev[0], stream));
  static const bool tiled_runs = [] { const char* e = getenv("POSEHIP_V1_RUNS"); return e && e[0] == '0'; }();
  const char* fe = getenv("POSEHIP_V1_FAST");   // "0": every irregular blob through the one-group walk (read per call: A/B tests)
  const bool no_fast = fe && fe[0] == '0';
  if (tiled_runs) {
    hipLaunchKernelGGL(v1_runs_kernel, dim3(tiles, m.n_parts, B), dim3(V1_RUN_THREADS), ldsA, stream, net_out, C, hs, ws, W, H,
                       thr, m.n_parts, c.run_cap, w.n_runs, w.runs, w.overflow);
  } else {
    const int blocks = ((W + V1_TILE_W - 1) / V1_TILE_W) * ((H + V1C_ROWS * V1C_WAVES - 1) / (V1C_ROWS * V1C_WAVES));
    // staged source sub-rectangle of a block: the cells that 64 + 2 rows x 256 + 2 columns can touch
    const size_t src_rows = (size_t)((V1C_ROWS * V1C_WAVES + 2) * (double)hs / H) + 4;
    const size_t src_cols = (size_t)((V1_TILE_W + 2) * (double)ws / W) + 4;
    const size_t ldsA2 = (src_rows < (size_t)hs ? src_rows : hs) * (src_cols < (size_t)ws ? src_cols : ws) * 4;
    hipLaunchKernelGGL(v1_runs_cols_kernel, dim3(blocks, m.n_parts, B), dim3(64 * V1C_WAVES), ldsA2, stream,
                       net_out, C, hs, ws, W, H, thr, m.n_parts, c.run_cap, (int)(ldsA2 / 4), w.n_runs, w.runs, w.overflow);
  }
  if (ev) PH_CHECK(hipEventRecord(ev[1], stream));
  hipLaunchKernelGGL(v1_blobs_kernel, dim3(m.n_parts, B), dim3(V1B_THREADS), ldsB, stream, net_out, C, hs, ws, W, H, m.n_parts,
                     c.run_cap, c.kp_cap, V1_VERT_POOL(c.run_cap), (const int*)w.n_runs, (const V1Run*)w.runs,
                     w.ws_verts, w.kps, w.n_kps, w.overflow, (int)ldsB, no_fast ? 0 : 1);
  if (ev) PH_CHECK(hipEventRecord(ev[2], stream));
  hipLaunchKernelGGL(v1_pairs_kernel, dim3(m.n_limbs, B), dim3(256), ldsC, stream, net_out, C, hs, ws, W, H, m,
                     c.kp_cap, (const float*)w.kps, (const int*)w.n_kps, w.pairs, w.n_pairs);
  hipLaunchKernelGGL(v1_assemble_kernel, dim3(B), dim3(64), ldsD, stream, m, c.kp_cap, c.person_cap,
                     (const float*)w.kps, (const int*)w.n_kps, (const double*)w.pairs, (const int*)w.n_pairs,
                     w.persons, w.n_persons, w.overflow);
  if (ev) PH_CHECK(hipEventRecord(ev[3], stream));
  PH_CHECK(hipGetLastError());
  return 0;
}
