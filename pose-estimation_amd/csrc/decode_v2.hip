// "V2" decode for gfx950: the post-network part of OpenPose's emplaceAndPop
// (reference call site openpose/native/python/skeleton.py:447), i.e. upstream
// resizeAndMerge (bicubic x8) -> nms (3x3 strict max, 7x7 weighted refine) ->
// bodyPartConnector (PAF line integral, sorted greedy matching, assembly,
// person filter).  Restated from the published upstream 1.7.0 algorithm; see
// oracle/decode_v2.py for the CPU statement this file is bit-exact against.
//
// MI355X design: the 75 MB/frame upsampled tensor is never written.  Kernel 1
// streams each heat-map plane through LDS in 8-row bands (horizontal cubic
// shared per source row, vertical cubic per pixel), detects peaks, orders them
// with an in-LDS rank sort and refines them; kernel 2 evaluates the PAF line
// integrals by bicubic sampling of the 46x82 source planes held in LDS and
// runs the sorted-greedy matching as iterated block-wide arg-max; kernel 3
// assembles people in LDS with wave-ballot searches.  This file is compiled
// with -ffp-contract=off: every fp32 operation is the single IEEE operation
// the oracle performs, in the same order.
#include "common.h"
#include "decode_v2.h"

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

// one pixel of the upsampled map from a source plane (LDS or global)
__device__ __forceinline__ float bicubic_at(const float* src, int hs, int ws, int Ht, int Wt, int x, int y) {
  const Taps tx = src_taps(x, Wt, ws);
  const Taps ty = src_taps(y, Ht, hs);
  const int rows[4] = {ty.i0, ty.i1, ty.i2, ty.i3};
  float t[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float* r = src + rows[i] * ws;
    t[i] = cubic(r[tx.i0], r[tx.i1], r[tx.i2], r[tx.i3], tx.d);
  }
  return cubic(t[0], t[1], t[2], t[3], ty.d);
}

// ---------------------------------------------------------------------------
// kernel 1: fused bicubic upsample + NMS + ordered compaction + refine
// grid (n_parts, B), block 256
// ---------------------------------------------------------------------------
#define NMS_BAND 8
#define NMS_TROWS 7  // source rows touched by NMS_BAND+2 output rows at scale >= 4: 3 steps + 4 taps

__global__ __launch_bounds__(256) void upsample_nms_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                           int Ht, int Wt, float threshold, int max_peaks,
                                                           float* __restrict__ peaks, int n_parts) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float* s_src = smem;                                  // hs*ws
  float* s_T = s_src + hs * ws;                         // NMS_TROWS * Wt
  float* s_U = s_T + NMS_TROWS * Wt;                    // (NMS_BAND+2) * Wt
  int* s_cand = (int*)(s_U + (NMS_BAND + 2) * Wt);      // (NMS_BAND/2) * (Wt/2+1)
  int* s_keys = s_cand + (NMS_BAND / 2) * (Wt / 2 + 1); // max_peaks
  __shared__ int s_ncand;

  const float* plane = net_out + ((size_t)b * C + part) * hs * ws;
  for (int i = tid; i < hs * ws; i += 256) s_src[i] = plane[i];
  if (tid == 0) s_ncand = 0;
  __syncthreads();

  int base = 0;  // peaks found so far (uniform across the block)
  for (int r0 = 0; r0 < Ht && base < max_peaks; r0 += NMS_BAND) {
    // output rows needed: [r0-1, r0+NMS_BAND] clamped to the image
    const int ylo = r0 - 1 < 0 ? 0 : r0 - 1;
    const int yhi = r0 + NMS_BAND > Ht - 1 ? Ht - 1 : r0 + NMS_BAND;
    const int slo = src_taps(ylo, Ht, hs).i0;
    const int shi = src_taps(yhi, Ht, hs).i3;
    const int nrows_t = shi - slo + 1;  // <= NMS_TROWS (checked on host)
    // 1. horizontal cubic of the touched source rows
    for (int i = tid; i < nrows_t * Wt; i += 256) {
      const int rr = i / Wt, x = i - rr * Wt;
      const Taps tx = src_taps(x, Wt, ws);
      const float* r = s_src + (slo + rr) * ws;
      s_T[rr * Wt + x] = cubic(r[tx.i0], r[tx.i1], r[tx.i2], r[tx.i3], tx.d);
    }
    __syncthreads();
    // 2. vertical cubic -> upsampled band with a 1-row halo
    const int nrows_u = yhi - ylo + 1;
    for (int i = tid; i < nrows_u * Wt; i += 256) {
      const int rr = i / Wt, x = i - rr * Wt;
      const Taps ty = src_taps(ylo + rr, Ht, hs);
      s_U[rr * Wt + x] = cubic(s_T[(ty.i0 - slo) * Wt + x], s_T[(ty.i1 - slo) * Wt + x],
                               s_T[(ty.i2 - slo) * Wt + x], s_T[(ty.i3 - slo) * Wt + x], ty.d);
    }
    __syncthreads();
    // 3. strict 3x3 maxima above threshold, 1-px image border excluded
    const int y_first = r0 < 1 ? 1 : r0;
    const int y_last = r0 + NMS_BAND - 1 > Ht - 2 ? Ht - 2 : r0 + NMS_BAND - 1;
    const int nrows_d = y_last - y_first + 1;
    for (int i = tid; i < nrows_d * (Wt - 2); i += 256) {
      const int rr = i / (Wt - 2), x = i - rr * (Wt - 2) + 1;
      const int y = y_first + rr;
      const float* u = s_U + (y - ylo) * Wt + x;
      const float v = u[0];
      if (v > threshold && v > u[-Wt - 1] && v > u[-Wt] && v > u[-Wt + 1] && v > u[-1] && v > u[1] &&
          v > u[Wt - 1] && v > u[Wt] && v > u[Wt + 1]) {
        const int slot = atomicAdd(&s_ncand, 1);
        s_cand[slot] = y * Wt + x;
      }
    }
    __syncthreads();
    // 4. raster order inside the band by rank; keep the first max_peaks overall
    const int n = s_ncand;
    for (int i = tid; i < n; i += 256) {
      const int key = s_cand[i];
      int rank = 0;
      for (int j = 0; j < n; j++) rank += s_cand[j] < key;
      if (base + rank < max_peaks) s_keys[base + rank] = key;
    }
    base += n;
    __syncthreads();
    if (tid == 0) s_ncand = 0;
    __syncthreads();
  }

  const int count = base < max_peaks ? base : max_peaks;
  float* out = peaks + ((size_t)b * n_parts + part) * (max_peaks + 1) * 3;
  if (tid == 0) out[0] = (float)count;
  // 5. 7x7 score-weighted centroid, sequential fp32 sums in (y, x) order
  for (int k = tid; k < count; k += 256) {
    const int key = s_keys[k];
    const int py = key / Wt, px = key - py * Wt;
    float xacc = 0.f, yacc = 0.f, sacc = 0.f;
    for (int yy = py - 3; yy <= py + 3; yy++) {
      if (yy < 0 || yy >= Ht) continue;
      for (int xx = px - 3; xx <= px + 3; xx++) {
        if (xx < 0 || xx >= Wt) continue;
        const float s = bicubic_at(s_src, hs, ws, Ht, Wt, xx, yy);
        if (s > 0.f) {
          xacc = xacc + (float)xx * s;
          yacc = yacc + (float)yy * s;
          sacc = sacc + s;
        }
      }
    }
    out[(k + 1) * 3 + 0] = ((xacc) / (sacc)) + 0.5f;
    out[(k + 1) * 3 + 1] = ((yacc) / (sacc)) + 0.5f;
    out[(k + 1) * 3 + 2] = bicubic_at(s_src, hs, ws, Ht, Wt, px, py);
  }
}

// ---------------------------------------------------------------------------
// kernel 2: PAF line-integral scoring + sorted-greedy matching per limb
// grid (n_pairs, B), block 256
// ---------------------------------------------------------------------------
__device__ __forceinline__ int pir(float x) { return (int)(x + 0.5f); }

__device__ float score_ab(const float* A, const float* Bp, const float* mapx, const float* mapy, int hs, int ws,
                          int Ht, int Wt, float inter_thr, float inter_min_above, double close_thr,
                          float close_score) {
  const float vx = Bp[0] - A[0];
  const float vy = Bp[1] - A[1];
  const float vmax = fmaxf(fabsf(vx), fabsf(vy));
  int npts = pir(sqrtf(5.0f * vmax));
  npts = npts > 25 ? 25 : npts;
  npts = npts < 5 ? 5 : npts;
  const float norm = sqrtf(vx * vx + vy * vy);
  if ((double)norm > 1e-6) {
    const float sx = A[0], sy = A[1];
    const float nx = ((vx) / (norm)), ny = ((vy) / (norm));
    float total = 0.f;
    int count = 0;
    const float stepx = vx / (float)npts, stepy = vy / (float)npts;
    for (int lm = 0; lm < npts; lm++) {
      int mx = pir(sx + (float)lm * stepx);
      int my = pir(sy + (float)lm * stepy);
      mx = mx > Wt - 1 ? Wt - 1 : mx;
      mx = mx < 0 ? 0 : mx;
      my = my > Ht - 1 ? Ht - 1 : my;
      my = my < 0 ? 0 : my;
      const float px = bicubic_at(mapx, hs, ws, Ht, Wt, mx, my);
      const float py = bicubic_at(mapy, hs, ws, Ht, Wt, mx, my);
      const float s = nx * px + ny * py;
      if (s > inter_thr) {
        total = total + s;
        count++;
      }
    }
    if ((float)count / (float)npts > inter_min_above) return total / (float)count;
    if ((double)norm < close_thr) return close_score;
  }
  return 0.f;
}

__global__ __launch_bounds__(256) void paf_match_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                        int Ht, int Wt, const float* __restrict__ peaks,
                                                        PhV2Dev mdl, float* __restrict__ pair_scores,
                                                        int* __restrict__ conn_idx, float* __restrict__ conn_score,
                                                        int* __restrict__ n_conn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pair = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int P1 = mdl.max_peaks + 1;
  const int partA = mdl.pairs[2 * pair], partB = mdl.pairs[2 * pair + 1];
  const float* candA = peaks + ((size_t)b * mdl.n_parts + partA) * P1 * 3;
  const float* candB = peaks + ((size_t)b * mdl.n_parts + partB) * P1 * 3;
  const int nA = pir(candA[0]), nB = pir(candB[0]);
  int* out_n = n_conn + (size_t)b * mdl.n_pairs + pair;
  if (nA == 0 || nB == 0) {
    if (tid == 0) *out_n = 0;
    return;
  }
  float* s_mx = smem;
  float* s_my = s_mx + hs * ws;
  float* s_A = s_my + hs * ws;  // P1*3
  float* s_B = s_A + P1 * 3;
  unsigned char* s_occA = (unsigned char*)(s_B + P1 * 3);
  unsigned char* s_occB = s_occA + P1;
  __shared__ unsigned long long s_best;
  __shared__ int s_count;

  const float* px = net_out + ((size_t)b * C + mdl.n_heat + mdl.mapidx[2 * pair]) * hs * ws;
  const float* py = net_out + ((size_t)b * C + mdl.n_heat + mdl.mapidx[2 * pair + 1]) * hs * ws;
  for (int i = tid; i < hs * ws; i += 256) {
    s_mx[i] = px[i];
    s_my[i] = py[i];
  }
  for (int i = tid; i < P1 * 3; i += 256) {
    s_A[i] = candA[i];
    s_B[i] = candB[i];
  }
  for (int i = tid; i < P1; i += 256) {
    s_occA[i] = 0;
    s_occB[i] = 0;
  }
  if (tid == 0) {
    s_best = 0ull;
    s_count = 0;
  }
  __syncthreads();

  float* sc = pair_scores + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks * mdl.max_peaks;
  const int npairs = nA * nB;
  for (int p = tid; p < npairs; p += 256) {
    const int i = p / nB, j = p - i * nB;  // 0-based
    sc[p] = score_ab(s_A + (i + 1) * 3, s_B + (j + 1) * 3, s_mx, s_my, hs, ws, Ht, Wt, mdl.inter_threshold,
                     mdl.inter_min_above, mdl.close_thr, mdl.close_score);
  }
  __syncthreads();  // this block's own global writes are visible to it after the barrier's vmcnt(0)

  // sorted-greedy == repeatedly take the max (score, i, j) with both ends free
  const int minAB = nA < nB ? nA : nB;
  int* oi = conn_idx + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks * 2;
  float* os = conn_score + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks;
  const int off = 3 * P1;
  for (int it = 0; it < minAB; it++) {
    unsigned long long best = 0ull;
    for (int p = tid; p < npairs; p += 256) {
      const float s = sc[p];
      if ((double)s > 1e-6) {
        const int i = p / nB, j = p - i * nB;
        if (!s_occA[i] && !s_occB[j]) {
          const unsigned long long key =
              ((unsigned long long)__float_as_uint(s) << 32) | ((unsigned)(i + 1) << 16) | (unsigned)(j + 1);
          best = key > best ? key : best;
        }
      }
    }
    if (best) atomicMax(&s_best, best);
    __syncthreads();
    const unsigned long long win = s_best;
    __syncthreads();
    if (win == 0ull) break;
    if (tid == 0) {
      const int i = (int)((win >> 16) & 0xffff), j = (int)(win & 0xffff);
      const int c = s_count;
      oi[2 * c] = partA * off + i * 3 + 2;
      oi[2 * c + 1] = partB * off + j * 3 + 2;
      os[c] = __uint_as_float((unsigned)(win >> 32));
      s_count = c + 1;
      s_occA[i - 1] = 1;
      s_occB[j - 1] = 1;
      s_best = 0ull;
    }
    __syncthreads();
  }
  if (tid == 0) *out_n = s_count;
}

// ---------------------------------------------------------------------------
// kernel 3: person assembly + filter + pack.  grid (B), block 64 (one wave)
// ---------------------------------------------------------------------------
#define ASM_MAX_PARTS 32

__global__ __launch_bounds__(64) void assemble_kernel(const float* __restrict__ peaks, PhV2Dev mdl,
                                                      const int* __restrict__ conn_idx,
                                                      const float* __restrict__ conn_score,
                                                      const int* __restrict__ n_conn, float scale, int max_people,
                                                      int row_cap, int* __restrict__ n_people,
                                                      float* __restrict__ keypoints, float* __restrict__ scores,
                                                      int* __restrict__ subsets, int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int NP = mdl.n_parts, P1 = mdl.max_peaks + 1, off = 3 * P1;
  const int RW = NP + 1;  // uint16 words per row (last = part count)
  float* s_score = smem;                                    // row_cap
  unsigned short* s_rows = (unsigned short*)(s_score + row_cap);  // row_cap * RW
  __shared__ int s_nrows;
  const float* pk = peaks + (size_t)b * NP * off;
  if (lane == 0) s_nrows = 0;
  __syncthreads();

  auto add_row = [&](int pa, int ia, int pb, int ib, int cnt, float sc) {  // lane 0 only
    const int r = s_nrows;
    if (r >= row_cap) {
      *overflow = 1;
      return;
    }
    unsigned short* row = s_rows + r * RW;
    for (int k = 0; k < NP; k++) row[k] = 0;
    row[pa] = (unsigned short)ia;
    if (pb >= 0) row[pb] = (unsigned short)ib;
    row[NP] = (unsigned short)cnt;
    s_score[r] = sc;
    s_nrows = r + 1;
  };
  // first row whose slot `part` holds idx, or -1 (wave-parallel search)
  auto find_row = [&](int part, int idx) -> int {
    const int n = s_nrows;
    for (int r0 = 0; r0 < n; r0 += 64) {
      const int r = r0 + lane;
      const bool hit = r < n && s_rows[r * RW + part] == (unsigned short)idx;
      const unsigned long long m = __ballot(hit);
      if (m) return r0 + __ffsll((long long)m) - 1;
    }
    return -1;
  };

  for (int pair = 0; pair < mdl.n_pairs; pair++) {
    const int partA = mdl.pairs[2 * pair], partB = mdl.pairs[2 * pair + 1];
    const int nA = pir(pk[partA * off]), nB = pir(pk[partB * off]);
    if (nA == 0 || nB == 0) {
      const int part = nA == 0 ? partB : partA;
      const int n = nA == 0 ? nB : nA;
      for (int i = 1; i <= n; i++) {
        const int idx = part * off + i * 3 + 2;
        const int f = find_row(part, idx);
        if (f < 0 && lane == 0) add_row(part, idx, -1, 0, 1, pk[idx]);
        __syncthreads();
      }
      continue;
    }
    const int nc = n_conn[(size_t)b * mdl.n_pairs + pair];
    const int* ci = conn_idx + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks * 2;
    const float* cs = conn_score + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks;
    if (nc == 0) continue;
    if (pair == 0) {
      if (lane == 0)
        for (int c = 0; c < nc; c++) {
          const int ia = ci[2 * c], ib = ci[2 * c + 1];
          add_row(partA, ia, partB, ib, 2, (pk[ia] + pk[ib]) + cs[c]);
        }
      __syncthreads();
    } else if (pair == mdl.ear_pair0 || pair == mdl.ear_pair1) {
      for (int c = 0; c < nc; c++) {
        const int ia = ci[2 * c], ib = ci[2 * c + 1];
        const int n = s_nrows;
        for (int r = lane; r < n; r += 64) {
          unsigned short* row = s_rows + r * RW;
          if (row[partA] == ia && row[partB] == 0)
            row[partB] = (unsigned short)ib;
          else if (row[partB] == ib && row[partA] == 0)
            row[partA] = (unsigned short)ia;
        }
        __syncthreads();
      }
    } else {
      for (int c = 0; c < nc; c++) {
        const int ia = ci[2 * c], ib = ci[2 * c + 1];
        const float s = cs[c];
        const int f = find_row(partA, ia);
        if (lane == 0) {
          if (f >= 0) {
            unsigned short* row = s_rows + f * RW;
            row[partB] = (unsigned short)ib;
            row[NP] = row[NP] + 1;
            s_score[f] = s_score[f] + (pk[ib] + s);
          } else {
            add_row(partA, ia, partB, ib, 2, (pk[ia] + pk[ib]) + s);
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();

  // ---- removePeopleBelowThresholds + peopleVectorToPeopleArray -------------
  const int n = s_nrows;
  const float denom = (float)(NP + mdl.n_pairs);
  int total = 0;
  for (int pass = 0; pass < 2; pass++) {  // pass 1 = maximizePositives retry
    total = 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
      const int r = r0 + lane;
      bool ok = false;
      if (r < n) {
        const unsigned short* row = s_rows + r * RW;
        int counter = row[NP];
        if (pass == 0 && NP == 25) {
          int feet = 0;
          for (int k = 19; k < 25; k++) feet += row[k] > 0;
          counter -= feet;
        }
        ok = counter >= mdl.min_subset_cnt && s_score[r] / (float)counter >= mdl.min_subset_score;
      }
      const unsigned long long m = __ballot(ok);
      if (ok) {
        const int o = total + __popcll(m & ((1ull << lane) - 1ull));
        if (o < max_people) {
          const unsigned short* row = s_rows + r * RW;
          float* kp = keypoints + ((size_t)b * max_people + o) * NP * 3;
          for (int k = 0; k < NP; k++) {
            const int idx = row[k];
            if (idx > 0) {
              kp[k * 3 + 0] = pk[idx - 2] * scale;
              kp[k * 3 + 1] = pk[idx - 1] * scale;
              kp[k * 3 + 2] = pk[idx];
            } else {
              kp[k * 3 + 0] = 0.f;
              kp[k * 3 + 1] = 0.f;
              kp[k * 3 + 2] = 0.f;
            }
          }
          scores[(size_t)b * max_people + o] = ((s_score[r]) / (denom));
          if (subsets) {
            int* sb = subsets + ((size_t)b * max_people + o) * RW;
            for (int k = 0; k <= NP; k++) sb[k] = row[k];
          }
        }
      }
      total += __popcll(m);
    }
    if (total > 0) break;
  }
  if (lane == 0) n_people[b] = total;
}

// ---------------------------------------------------------------------------
// optional: materialise upsampled maps (datum.poseHeatMaps export only -- the
// decode itself never needs them).  grid-stride over [nc][Ht][Wt] of one frame.
// ---------------------------------------------------------------------------
__global__ void upsample_maps_kernel(const float* __restrict__ planes, int hs, int ws, int Ht, int Wt, int nc,
                                     float* __restrict__ out) {
  const size_t total = (size_t)nc * Ht * Wt;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % Wt;
    const size_t t = i / Wt;
    const int y = t % Ht;
    const int c = t / Ht;
    out[i] = bicubic_at(planes + (size_t)c * hs * ws, hs, ws, Ht, Wt, x, y);
  }
}

int ph_launch_upsample_maps(const float* planes, int hs, int ws, int Ht, int Wt, int nc, float* out,
                            hipStream_t stream) {
  const size_t total = (size_t)nc * Ht * Wt;
  int blocks = (int)((total + 255) / 256);
  blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(upsample_maps_kernel, dim3(blocks), dim3(256), 0, stream, planes, hs, ws, Ht, Wt, nc, out);
  PH_CHECK(hipGetLastError());
  return 0;
}

// per-map global maximum and its first (raster) location: the reference's
// single-pose decode, cv2.minMaxLoc per heat map (openpose/opencv/skeleton.py:95)
__global__ __launch_bounds__(256) void argmax_maps_kernel(const float* __restrict__ maps, int n, float* __restrict__ vals,
                                                          int* __restrict__ locs) {
  __shared__ float s_v[256];
  __shared__ int s_i[256];
  const float* m = maps + (size_t)blockIdx.x * n;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = m[i];
    if (v > bv) {  // strided scan: the lowest index among equal maxima wins below
      bv = v;
      bi = i;
    }
  }
  s_v[threadIdx.x] = bv;
  s_i[threadIdx.x] = bi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      const float ov = s_v[threadIdx.x + st];
      const int oi = s_i[threadIdx.x + st];
      if (ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) {
        s_v[threadIdx.x] = ov;
        s_i[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    vals[blockIdx.x] = s_v[0];
    locs[blockIdx.x] = s_i[0] == 0x7fffffff ? 0 : s_i[0];
  }
}

int ph_launch_argmax_maps(const float* maps, int n_maps, int n, float* vals, int* locs, hipStream_t stream) {
  hipLaunchKernelGGL(argmax_maps_kernel, dim3(n_maps), dim3(256), 0, stream, maps, n, vals, locs);
  PH_CHECK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// host launcher
// ---------------------------------------------------------------------------
int ph_v2_workspace_sizes(const PhV2Dev& m, int B, int max_people, PhV2Sizes* s) {
  s->peaks = (size_t)B * m.n_parts * (m.max_peaks + 1) * 3 * sizeof(float);
  s->pair_scores = (size_t)B * m.n_pairs * m.max_peaks * m.max_peaks * sizeof(float);
  s->conn_idx = (size_t)B * m.n_pairs * m.max_peaks * 2 * sizeof(int);
  s->conn_score = (size_t)B * m.n_pairs * m.max_peaks * sizeof(float);
  s->n_conn = (size_t)B * m.n_pairs * sizeof(int);
  s->n_people = (size_t)B * sizeof(int);
  s->keypoints = (size_t)B * max_people * m.n_parts * 3 * sizeof(float);
  s->scores = (size_t)B * max_people * sizeof(float);
  s->subsets = (size_t)B * max_people * (m.n_parts + 1) * sizeof(int);
  return 0;
}

int ph_launch_decode_v2(const float* net_out, int C, int hs, int ws, int Ht, int Wt, const PhV2Dev& m, int B,
                        float scale, int max_people, const PhV2Buffers& w, hipStream_t stream, hipEvent_t* ev) {
  if (m.n_parts > ASM_MAX_PARTS || m.n_pairs > 64) {
    ph_set_error("decode_v2: model too large (%d parts, %d pairs)", m.n_parts, m.n_pairs);
    return -1;
  }
  if (Ht < 4 * hs || Wt < 4 * ws) {
    ph_set_error("decode_v2: target %dx%d must be >= 4x the net output %dx%d", Ht, Wt, hs, ws);
    return -1;
  }
  if (Ht < 3 || Wt < 3 || Wt > 65535 || Ht > 32767) {
    ph_set_error("decode_v2: bad target size %dx%d", Ht, Wt);
    return -1;
  }
  if (m.n_heat + 2 * m.n_pairs > C) {
    ph_set_error("decode_v2: net output has %d channels, model needs %d", C, m.n_heat + 2 * m.n_pairs);
    return -1;
  }
  const size_t lds1 = ((size_t)hs * ws + (size_t)NMS_TROWS * Wt + (size_t)(NMS_BAND + 2) * Wt +
                       (size_t)(NMS_BAND / 2) * (Wt / 2 + 1) + m.max_peaks) * 4;
  const size_t lds2 = ((size_t)2 * hs * ws + (size_t)2 * (m.max_peaks + 1) * 3) * 4 + 2 * (m.max_peaks + 1) + 16;
  const int row_cap = 2048;
  const size_t lds3 = (size_t)row_cap * 4 + (size_t)row_cap * (m.n_parts + 1) * 2;
  if (lds1 > 160 * 1024 - 64 || lds2 > 160 * 1024 - 64 || lds3 > 160 * 1024 - 64) {
    ph_set_error("decode_v2: LDS budget exceeded (%zu/%zu/%zu B) for net output %dx%d target %dx%d", lds1, lds2,
                 lds3, hs, ws, Ht, Wt);
    return -1;
  }
  static bool attr_done = false;
  if (!attr_done) {
    PH_CHECK(hipFuncSetAttribute((const void*)upsample_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - 64));
    PH_CHECK(hipFuncSetAttribute((const void*)paf_match_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - 64));
    PH_CHECK(hipFuncSetAttribute((const void*)assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - 64));
    attr_done = true;
  }
  PhV2Sizes sz;
  ph_v2_workspace_sizes(m, B, max_people, &sz);
  PH_CHECK(hipMemsetAsync(w.peaks, 0, sz.peaks, stream));
  PH_CHECK(hipMemsetAsync(w.overflow, 0, sizeof(int), stream));
  if (ev) PH_CHECK(hipEventRecord(ev[0], stream));
  hipLaunchKernelGGL(upsample_nms_kernel, dim3(m.n_parts, B), dim3(256), lds1, stream, net_out, C, hs, ws, Ht, Wt,
                     m.nms_threshold, m.max_peaks, w.peaks, m.n_parts);
  if (ev) PH_CHECK(hipEventRecord(ev[1], stream));
  hipLaunchKernelGGL(paf_match_kernel, dim3(m.n_pairs, B), dim3(256), lds2, stream, net_out, C, hs, ws, Ht, Wt,
                     (const float*)w.peaks, m, w.pair_scores, w.conn_idx, w.conn_score, w.n_conn);
  if (ev) PH_CHECK(hipEventRecord(ev[2], stream));
  hipLaunchKernelGGL(assemble_kernel, dim3(B), dim3(64), lds3, stream, (const float*)w.peaks, m,
                     (const int*)w.conn_idx, (const float*)w.conn_score, (const int*)w.n_conn, scale, max_people,
                     row_cap, w.n_people, w.keypoints, w.scores, w.subsets, w.overflow);
  if (ev) PH_CHECK(hipEventRecord(ev[3], stream));
  PH_CHECK(hipGetLastError());
  return 0;
}
