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

#include "bicubic.h"

// ---------------------------------------------------------------------------
// kernel 1a: fused bicubic upsample + strict 3x3 NMS + ordered compaction.
// grid (n_parts, B, n_split): each block owns a horizontal stripe of the
// upsampled map and emits that stripe's peak pixels in raster order.
// ---------------------------------------------------------------------------
#define NMS_BAND 8

// ---------------------------------------------------------------------------
// kernel 1a, column form: no upsampled pixel ever goes through LDS (the first form staged upsampled 8-row bands
// in LDS and took 0.50 ms per 32 frames against 0.27 ms; removed in round 2).  A lane owns one output column and
// walks down the stripe: the four horizontally interpolated source rows
// of the current vertical taps and the cubic's coefficients live in registers (they change once per source row,
// i.e. every `scale` output rows), an output pixel is the 8 fp32 operations of the cubic in d -- the very
// operations of cubic(), so results are bit-identical -- and the 3x3 test takes the column maxima of the two
// neighbouring lanes by shuffle.  A wave covers 64 consecutive columns of which lanes 1..62 emit (lanes 0 and 63
// only feed their neighbours); a wave owns up to G such column groups.  Block-wide work is only the ordering of a
// band's peaks (as above).
// ---------------------------------------------------------------------------
#define NMS_COLS 62
// value of lane-1 / lane+1 across the whole wave (DPP wave shifts: no LDS round trip); the end lanes keep `v`
__device__ __forceinline__ float lane_left(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_right(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// FAST: the instantiation for planes whose values are all finite and below 1e30 -- nothing in the two cubic passes
// can then be an infinity or a NaN (|result| <= 10 |input| per pass), so the loop body needs no NaN bookkeeping, the
// eight strict tests collapse into one comparison with a maximum, and hits are collected as a bit per (row, column
// group) and emitted once per band: no branch per pixel.  A block that finds another kind of plane marks its stripe
// (key_cnt = -1) and leaves it to the !FAST instantiation, launched right after on the same stream, whose blocks
// return at once unless their stripe carries that mark.  (One kernel with both bodies: 88 instead of 60 registers
// at G = 2 and 0.34 instead of 0.29 ms per 32 frames with the refine kernel; as two kernels, the fast one held to 64
// registers: 0.27 ms.  The row body is now 29 VALU instructions for two pixels, 9 of them packed; SQ_INSTS_VALU said
// the kernel issued 60 per row body -- the rest was load_taps (the source rows of the next `scale` output rows and
// the cubic's a, b, c per column group, behind chains of uniform branches) and the copies at the loop's joins; with
// the window of source rows moved on by one in the usual case instead of searched: 50, 0.25 ms; with the bands cut
// where the source rows change, so that the taps are loaded once per band outside the unrolled rows (a third of the
// scalar instructions and more than half of the code gone): 0.24 ms.)
template <int G, bool FAST>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(G == 4 ? 4 : 8))) void nms_detect_cols_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                               int Ht, int Wt, float threshold, int max_peaks,
                                                               int n_parts, int n_split, int rows_per_split,
                                                               int n_groups, int* __restrict__ keys,
                                                               int* __restrict__ key_cnt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int part = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
  const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = tid >> 6, n_waves = NT >> 6;
  float* s_src = smem;                                        // hs*ws
  TapsP* s_yt = (TapsP*)(s_src + hs * ws);                    // rows_per_split + 2: vertical taps of every row
  int* s_seg = (int*)(s_yt + rows_per_split + 2);             // rows_per_split + 3: last row of the run of rows
                                                              // with the same four source rows that starts at i
  int* s_cand = s_seg + rows_per_split + 3;                   // (NMS_BAND/2) * (Wt/2+1)
  int* s_keys = s_cand + (NMS_BAND / 2) * (Wt / 2 + 1);       // max_peaks
  __shared__ int s_ncand;

  const int y_begin = sp * rows_per_split;
  const int y_end = y_begin + rows_per_split < Ht ? y_begin + rows_per_split : Ht;
  const int yo_first = y_begin < 1 ? 1 : y_begin;             // output rows: 1-px image border excluded
  const int yo_last = y_end - 1 > Ht - 2 ? Ht - 2 : y_end - 1;
  const int ylo = yo_first - 1;                               // upsampled rows [ylo, yo_last + 1] are evaluated
  const int n_u = yo_last + 2 - ylo;
  int* out = keys + (((size_t)b * n_parts + part) * n_split + sp) * max_peaks;
  int* const my_cnt = key_cnt + ((size_t)b * n_parts + part) * n_split + sp;
  if (!FAST && *my_cnt != -1) return;                         // (block-uniform) the fast instantiation did this stripe
  if (yo_first > yo_last) {                                   // (block-uniform)
    if (tid == 0) *my_cnt = 0;
    return;
  }
  const float* plane = net_out + ((size_t)b * C + part) * hs * ws;
  bool fin = true;
  for (int i = tid; i < hs * ws; i += NT) {
    const float v = plane[i];
    s_src[i] = v;
    fin = fin && fabsf(v) < 1e30f;
  }
  for (int i = tid; i < n_u; i += NT) s_yt[i] = pack_taps(src_taps(ylo + i, Ht, hs));
  if (tid == 0) s_ncand = 0;

  int xcol[G];
  bool emit[G];
  TapsP tx[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int gi = wave + g * n_waves;
    const int x = gi * NMS_COLS + lane - 1;
    xcol[g] = x;
    emit[g] = gi < n_groups && lane >= 1 && lane <= NMS_COLS && x >= 1 && x <= Wt - 2;
    const int xc = x < 0 ? 0 : (x > Wt - 1 ? Wt - 1 : x);
    tx[g] = pack_taps(src_taps(xc, Wt, ws));
  }
  if (FAST) {
    if (!__syncthreads_and(fin)) {                            // (block-uniform)
      if (tid == 0) *my_cnt = -1;
      return;
    }
  } else {
    __syncthreads();
  }
  constexpr bool fast = FAST;
  unsigned emit_bits = 0;                                     // bit r * G + g: column group g emits in row r of a band
#pragma unroll
  for (int r = 0; r < NMS_BAND; r++)
#pragma unroll
    for (int g = 0; g < G; g++) emit_bits |= emit[g] ? 1u << (r * G + g) : 0u;
  for (int i = tid; i < n_u; i += NT) {                       // run ends (runs are <= ceil(scale) rows long)
    const TapsP t = s_yt[i];
    int e = i;
    while (e + 1 < n_u) {
      const TapsP u = s_yt[e + 1];
      if (u.i0 != t.i0 || u.i1 != t.i1 || u.i2 != t.i2 || u.i3 != t.i3) break;
      e++;
    }
    s_seg[i] = e;
  }
  __syncthreads();

  int ti0 = -1, ti1 = -1, ti2 = -1, ti3 = -1;                 // source rows held in tv[][0..3] (block-uniform)
  float tv[G][4], ca[G], cb[G], cc[G];
#pragma unroll
  for (int g = 0; g < G; g++) tv[g][0] = tv[g][1] = tv[g][2] = tv[g][3] = ca[g] = cb[g] = cc[g] = 0.f;
  // the four source rows of upsampled row index i (relative to ylo) -> registers, cubic()'s a, b, c of them
  auto load_taps = [&](int i) {
    const TapsP ty = s_yt[i];
    const int n0 = __builtin_amdgcn_readfirstlane((int)ty.i0), n1 = __builtin_amdgcn_readfirstlane((int)ty.i1);
    const int n2 = __builtin_amdgcn_readfirstlane((int)ty.i2), n3 = __builtin_amdgcn_readfirstlane((int)ty.i3);
    // the usual step down the image: the window of source rows moves on by one (its last row may repeat at the
    // bottom edge) -- one new interpolated row per column group, no search through the rows held
    const bool shift = n0 == ti1 && n1 == ti2 && n2 == ti3;
#pragma unroll
    for (int g = 0; g < G; g++) {
      float v0, v1, v2, v3;
      if (shift) {
        v0 = tv[g][1]; v1 = tv[g][2]; v2 = tv[g][3];
        v3 = n3 == ti3 ? tv[g][3] : cubic_row(s_src + n3 * ws, tx[g]);
      } else {
        auto pick = [&](int nk) -> float {
          if (nk == ti0) return tv[g][0];
          if (nk == ti1) return tv[g][1];
          if (nk == ti2) return tv[g][2];
          if (nk == ti3) return tv[g][3];
          return cubic_row(s_src + nk * ws, tx[g]);
        };
        v0 = pick(n0); v1 = pick(n1); v2 = pick(n2); v3 = pick(n3);
      }
      tv[g][0] = v0; tv[g][1] = v1; tv[g][2] = v2; tv[g][3] = v3;
      ca[g] = (((-0.5f * v0 + 1.5f * v1) - 1.5f * v2) + 0.5f * v3);   // cubic()'s a, b, c
      cb[g] = (((v0 - 2.5f * v1) + 2.0f * v2) - 0.5f * v3);
      cc[g] = 0.5f * (v0 - v2);
    }
    ti0 = n0; ti1 = n1; ti2 = n2; ti3 = n3;
    return __builtin_amdgcn_readfirstlane(s_seg[i]);
  };
  auto u_val = [&](int g, float d) { return ((((ca[g] * d) * d) * d + (cb[g] * d) * d) - cc[g] * d) + tv[g][1]; };
  // the same operations for two column groups at once, as 2-vectors (packed fp32 multiplies and adds)
  typedef float nms_f2 __attribute__((ext_vector_type(2)));
  auto u_val2 = [&](int g, float d, float& u0, float& u1) {
    const nms_f2 a = {ca[g], ca[g + 1]}, bq = {cb[g], cb[g + 1]}, c = {cc[g], cc[g + 1]}, v1 = {tv[g][1], tv[g + 1][1]};
    const nms_f2 dv = {d, d};
    const nms_f2 u = ((((a * dv) * dv) * dv + (bq * dv) * dv) - c * dv) + v1;
    u0 = u.x;
    u1 = u.y;
  };

  float up[G], uc[G];
  int seg_last = load_taps(0);                                // rows are indexed relative to ylo from here on
  {
    const float d0 = s_yt[0].d;
#pragma unroll
    for (int g = 0; g < G; g++) up[g] = u_val(g, d0);
  }
  if (seg_last < 1) seg_last = load_taps(1);
  {
    const float d1 = s_yt[1].d;
#pragma unroll
    for (int g = 0; g < G; g++) uc[g] = u_val(g, d1);
  }
  const int i_last = yo_last - ylo;                           // output rows are i = 1 .. i_last
  int key_row = yo_first * Wt;
  int base = 0;      // peaks found so far in this stripe (uniform across the block)
  for (int i = 1; i <= i_last && base < max_peaks;) {
    int band_last = i + NMS_BAND - 1 < i_last ? i + NMS_BAND - 1 : i_last;
    if constexpr (FAST) {
      // a band ends where the source rows of the NEXT output row change (row i evaluates row i + 1): the taps are
      // loaded here, once per band, and the unrolled rows below carry neither the test nor eight copies of that code
      if (i + 1 > seg_last) seg_last = load_taps(i + 1);
      band_last = band_last < seg_last - 1 ? band_last : seg_last - 1;
    }
    float dd[NMS_BAND];                                       // the band's vertical phases: one LDS wait per band
#pragma unroll
    for (int r = 0; r < NMS_BAND; r++) dd[r] = s_yt[i + 1 + r < n_u ? i + 1 + r : n_u - 1].d;
    // Finite planes: a pixel is a peak iff it exceeds the maximum of the threshold, its two vertical neighbours and
    // the two neighbouring columns' 3-row maxima (no NaN anywhere, so that is the conjunction of the eight strict
    // tests); hits are collected as a bit per (row, column group) and emitted once per band -- no branch per pixel.
    // Column groups beyond the last one run on clamped taps and never emit.
    unsigned hit = 0;
    const int key_band = key_row;
    if constexpr (FAST) {
#pragma unroll
      for (int r = 0; r < NMS_BAND; r++) {
        if (i > band_last) break;
        const float d = dd[r];
        float un[G];
        if constexpr (G == 1) {
          un[0] = u_val(0, d);
        } else {
#pragma unroll
          for (int g = 0; g < G; g += 2) u_val2(g, d, un[g], un[g + 1]);
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
          const float m3 = fmaxf(fmaxf(up[g], uc[g]), un[g]);
          const float ml = lane_left(m3), mr = lane_right(m3);
          const float v = uc[g];
          const float t = fmaxf(fmaxf(fmaxf(fmaxf(up[g], un[g]), threshold), ml), mr);
          hit |= v > t ? 1u << (r * G + g) : 0u;
          up[g] = v;
          uc[g] = un[g];
        }
        key_row += Wt;
        i++;
      }
      hit &= emit_bits;
      while (hit) {
        const int bit = __builtin_ctz(hit);
        hit &= hit - 1;
        const int g = bit % G, r = bit / G;
        int xc = xcol[0];
#pragma unroll
        for (int q = 1; q < G; q++) xc = g == q ? xcol[q] : xc;
        const int slot = atomicAdd(&s_ncand, 1);
        s_cand[slot] = key_band + r * Wt + xc;
      }
    }
#pragma unroll
    for (int r = 0; r < NMS_BAND; r++) {
      if (fast || i > band_last) break;
      if (i + 1 > seg_last) seg_last = load_taps(i + 1);
      const float d = dd[r];
#pragma unroll
      for (int g = 0; g < G; g++) {
        if (G > 1 && wave + g * n_waves >= n_groups) continue;
        const float un = u_val(g, d);
        // column maximum of the 3 rows; NaN if any is NaN (then `v > m` is false, as `v > NaN` is in the 8 tests)
        float m3 = fmaxf(fmaxf(up[g], uc[g]), un);
        if (up[g] != up[g] || uc[g] != uc[g] || un != un) m3 = __builtin_nanf("");
        const float ml = lane_left(m3), mr = lane_right(m3);
        const float v = uc[g];
        if (emit[g] && v > threshold && v > up[g] && v > un && v > ml && v > mr) {
          const int slot = atomicAdd(&s_ncand, 1);
          s_cand[slot] = key_row + xcol[g];
        }
        up[g] = v;
        uc[g] = un;
      }
      key_row += Wt;
      i++;
    }
    __syncthreads();
    // raster order inside the band by rank; keep the first max_peaks of the stripe
    const int n = s_ncand;
    for (int k = tid; k < n; k += NT) {
      const int key = s_cand[k];
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
  for (int k = tid; k < count; k += NT) out[k] = s_keys[k];
  if (tid == 0) *my_cnt = count;
}

// ---------------------------------------------------------------------------
// kernel 1b: concatenate the stripes' peak lists (raster order, first
// max_peaks), 7x7 score-weighted centroid.  grid (n_parts, B), block 128
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(128) void nms_refine_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                         int Ht, int Wt, int max_peaks, int n_parts, int n_split,
                                                         float offset, const int* __restrict__ keys,
                                                         const int* __restrict__ key_cnt, float* __restrict__ peaks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float* s_src = smem;
  const float* plane = net_out + ((size_t)b * C + part) * hs * ws;
  const int* cnt = key_cnt + ((size_t)b * n_parts + part) * n_split;
  int total = 0;
  for (int s = 0; s < n_split; s++) total += cnt[s];
  const int count = total < max_peaks ? total : max_peaks;
  float* out = peaks + ((size_t)b * n_parts + part) * (max_peaks + 1) * 3;
  if (tid == 0) out[0] = (float)count;
  if (count == 0) return;
  for (int i = tid; i < hs * ws; i += 128) s_src[i] = plane[i];
  __syncthreads();
  for (int k = tid; k < count; k += 128) {
    int s = 0, kk = k;
    while (kk >= cnt[s]) {
      kk -= cnt[s];
      s++;
    }
    const int key = keys[(((size_t)b * n_parts + part) * n_split + s) * max_peaks + kk];
    const int py = key / Wt, px = key - py * Wt;
    // sequential fp32 sums in (y, x) order over the in-image part of the 7x7 window
    TapsP tx[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
      int xx = px - 3 + i;
      xx = xx < 0 ? 0 : (xx > Wt - 1 ? Wt - 1 : xx);
      tx[i] = pack_taps(src_taps(xx, Wt, ws));
    }
    float xacc = 0.f, yacc = 0.f, sacc = 0.f;
    for (int yy = py - 3; yy <= py + 3; yy++) {
      if (yy < 0 || yy >= Ht) continue;
      const TapsP ty = pack_taps(src_taps(yy, Ht, hs));
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const int xx = px - 3 + i;
        if (xx < 0 || xx >= Wt) continue;
        const float sv = bicubic_tp(s_src, ws, tx[i], ty);
        if (sv > 0.f) {
          xacc = xacc + (float)xx * sv;
          yacc = yacc + (float)yy * sv;
          sacc = sacc + sv;
        }
      }
    }
    out[(k + 1) * 3 + 0] = xacc / sacc + offset;
    out[(k + 1) * 3 + 1] = yacc / sacc + offset;
    out[(k + 1) * 3 + 2] = bicubic_tp(s_src, ws, tx[3], pack_taps(src_taps(py, Ht, hs)));
  }
}

// ---------------------------------------------------------------------------
// kernel 2: PAF line-integral scoring (2a) + sorted-greedy matching per limb (2b)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int pir(float x) { return (int)(x + 0.5f); }

#define PAF_LIST_CAP 8192
#define PAF_THREADS 1024

// One cubic_row() from a table of its coefficients: cubic()'s a, b, c and v1 depend on the source row and the
// centre tap only (the other taps follow from it by clamping), so they are computed once per source pixel and a
// sample costs one 16-byte LDS read + the 8 operations in d per row instead of 4 reads + 24 operations.  Same
// operations on the same values as cubic_row(): bit-identical.
__device__ __forceinline__ float4 cubic_coef(const float* __restrict__ row, int c, int ws) {
  const int i0 = c - 1 < 0 ? 0 : c - 1, i2 = c + 1 > ws - 1 ? ws - 1 : c + 1, i3 = i2 + 1 > ws - 1 ? ws - 1 : i2 + 1;
  const float v0 = row[i0], v1 = row[c], v2 = row[i2], v3 = row[i3];
  float4 q;
  q.x = (((-0.5f * v0 + 1.5f * v1) - 1.5f * v2) + 0.5f * v3);
  q.y = (((v0 - 2.5f * v1) + 2.0f * v2) - 0.5f * v3);
  q.z = 0.5f * (v0 - v2);
  q.w = v1;
  return q;
}
// The two PAF planes of a limb are sampled at the same point, so their tables are kept INTERLEAVED -- 32 B per source
// pixel: {a_x, a_y, b_x, b_y}, {c_x, c_y, v1_x, v1_y} -- and both planes go through cubic()'s operations as 2-vectors
// (packed fp32 multiplies and adds: each element sees the same IEEE operations in the same order as cubic_row() / cubic() on one plane).
typedef float paf_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ paf_f2 cubic_row_cf2(const float4* row, int i1, paf_f2 d) {
  const float4 q0 = row[2 * i1], q1 = row[2 * i1 + 1];
  const paf_f2 a = {q0.x, q0.y}, b = {q0.z, q0.w}, c = {q1.x, q1.y}, v1 = {q1.z, q1.w};
  return ((((a * d) * d) * d + (b * d) * d) - c * d) + v1;
}
__device__ __forceinline__ paf_f2 cubic2(paf_f2 v0, paf_f2 v1, paf_f2 v2, paf_f2 v3, paf_f2 d) {
  const paf_f2 a = (((-0.5f * v0 + 1.5f * v1) - 1.5f * v2) + 0.5f * v3);
  const paf_f2 b = (((v0 - 2.5f * v1) + 2.0f * v2) - 0.5f * v3);
  const paf_f2 c = 0.5f * (v0 - v2);
  return ((((a * d) * d) * d + (b * d) * d) - c * d) + v1;
}
__device__ __forceinline__ paf_f2 bicubic_cf2(const float4* tab, int ws, const TapsP& tx, const TapsP& ty) {
  const paf_f2 dx = {tx.d, tx.d}, dy = {ty.d, ty.d};
  const paf_f2 t0 = cubic_row_cf2(tab + 2 * (ty.i0 * ws), tx.i1, dx), t1 = cubic_row_cf2(tab + 2 * (ty.i1 * ws), tx.i1, dx);
  const paf_f2 t2 = cubic_row_cf2(tab + 2 * (ty.i2 * ws), tx.i1, dx), t3 = cubic_row_cf2(tab + 2 * (ty.i3 * ws), tx.i1, dx);
  return cubic2(t0, t1, t2, t3, dy);
}

// COEF: mapx / mapy are float4 coefficient tables (above), else the raw planes.
// kmin[n] = the smallest count k with (float)k / (float)n > inter_min_above (n + 1 if none): upstream's acceptance
// test and the "can no longer pass" test below are that comparison, which is monotone in k.
template <bool COEF>
__device__ float score_ab(const float* A, const float* Bp, const void* mapx, const void* mapy, int ws, int Ht,
                          int Wt, const TapsP* xt, const TapsP* yt, float inter_thr, const unsigned char* kmin,
                          double close_thr, float close_score) {
  const float vx = Bp[0] - A[0];
  const float vy = Bp[1] - A[1];
  const float vmax = fmaxf(fabsf(vx), fabsf(vy));
  int npts = pir(sqrtf(5.0f * vmax));
  npts = npts > 25 ? 25 : npts;
  npts = npts < 5 ? 5 : npts;
  const float norm = sqrtf(vx * vx + vy * vy);
  if ((double)norm > 1e-6) {
    const float sx = A[0], sy = A[1];
    const float nx = vx / norm, ny = vy / norm;
    float total = 0.f;
    int count = 0;
    const float fn = (float)npts;
    const float stepx = vx / fn, stepy = vy / fn;
    const int need = kmin[npts];
    bool alive = true;
    for (int lm = 0; lm < npts; lm++) {
      // the acceptance test below can no longer pass: the sum is not needed (same result, less work)
      if (count + npts - lm < need) {
        alive = false;
        break;
      }
      int mx = pir(sx + (float)lm * stepx);
      int my = pir(sy + (float)lm * stepy);
      mx = mx > Wt - 1 ? Wt - 1 : mx;
      mx = mx < 0 ? 0 : mx;
      my = my > Ht - 1 ? Ht - 1 : my;
      my = my < 0 ? 0 : my;
      const TapsP tx = xt[mx], ty = yt[my];
      float px, py;
      if constexpr (COEF) {   // mapx = the interleaved table of both planes
        const paf_f2 pxy = bicubic_cf2((const float4*)mapx, ws, tx, ty);
        px = pxy.x;
        py = pxy.y;
      } else {
        px = bicubic_tp((const float*)mapx, ws, tx, ty);
        py = bicubic_tp((const float*)mapy, ws, tx, ty);
      }
      const float s = nx * px + ny * py;
      if (s > inter_thr) {
        total = total + s;
        count++;
      }
    }
    if (alive && count >= need) return total / (float)count;
    if ((double)norm < close_thr) return close_score;
  }
  return 0.f;
}

// kernel 2a: scores of the candidate pairs of one limb of one frame; grid (n_pairs, B, n_sub): the pairs of a limb
// are cut into n_sub ranges (a limb with many accepted pairs costs 10x the average one, and one block per limb
// left the kernel waiting for its slowest blocks).  Writes the nA x nB score matrix.
#define PAF_MIN_CHUNK 1024
template <bool COEF>
__global__ __launch_bounds__(PAF_THREADS) void paf_score_kernel(const float* __restrict__ net_out, int C, int hs, int ws,
                                                                int Ht, int Wt, const float* __restrict__ peaks,
                                                                PhV2Dev mdl, float* __restrict__ pair_scores,
                                                                int n_sub, int* __restrict__ n_accept) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pair = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int P1 = mdl.max_peaks + 1;
  const int partA = mdl.pairs[2 * pair], partB = mdl.pairs[2 * pair + 1];
  const float* candA = peaks + ((size_t)b * mdl.n_parts + partA) * P1 * 3;
  const float* candB = peaks + ((size_t)b * mdl.n_parts + partB) * P1 * 3;
  const int nA = pir(candA[0]), nB = pir(candB[0]);
  const int npairs = nA * nB;
  int chunk = (npairs + n_sub - 1) / n_sub;
  chunk = chunk < PAF_MIN_CHUNK ? PAF_MIN_CHUNK : chunk;
  const int p_begin = blockIdx.z * chunk;
  if (p_begin >= npairs) return;  // (also nA == 0 or nB == 0)
  const int p_end = p_begin + chunk < npairs ? p_begin + chunk : npairs;

  constexpr int PLANE_WORDS = COEF ? 4 : 1;  // floats per source pixel and channel
  float* s_mx = smem;
  float* s_my = s_mx + hs * ws * PLANE_WORDS;
  float* s_A = s_my + hs * ws * PLANE_WORDS;  // P1*3
  float* s_B = s_A + P1 * 3;
  TapsP* s_xt = (TapsP*)(s_B + P1 * 3);  // Wt
  TapsP* s_yt = s_xt + Wt;               // Ht
  __shared__ unsigned char s_kmin[32];

  const float* px = net_out + ((size_t)b * C + mdl.n_heat + mdl.mapidx[2 * pair]) * hs * ws;
  const float* py = net_out + ((size_t)b * C + mdl.n_heat + mdl.mapidx[2 * pair + 1]) * hs * ws;
  for (int i = tid; i < hs * ws; i += PAF_THREADS) {
    if (COEF) {
      const int r = i / ws, c = i - r * ws;
      const float4 qx = cubic_coef(px + r * ws, c, ws), qy = cubic_coef(py + r * ws, c, ws);
      ((float4*)s_mx)[2 * i] = make_float4(qx.x, qy.x, qx.y, qy.y);       // (s_my follows s_mx: one 32-B entry per pixel)
      ((float4*)s_mx)[2 * i + 1] = make_float4(qx.z, qy.z, qx.w, qy.w);
    } else {
      s_mx[i] = px[i];
      s_my[i] = py[i];
    }
  }
  for (int i = tid; i < P1 * 3; i += PAF_THREADS) {
    s_A[i] = candA[i];
    s_B[i] = candB[i];
  }
  for (int x = tid; x < Wt; x += PAF_THREADS) s_xt[x] = pack_taps(src_taps(x, Wt, ws));
  for (int y = tid; y < Ht; y += PAF_THREADS) s_yt[y] = pack_taps(src_taps(y, Ht, hs));
  if (tid < 32) {  // sample counts are 5..25
    int k = 0;
    while (k <= tid && !((float)k / (float)tid > mdl.inter_min_above)) k++;
    s_kmin[tid] = (unsigned char)k;
  }
  __syncthreads();

  float* sc = pair_scores + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks * mdl.max_peaks;
  int accepted = 0;
  for (int p = p_begin + tid; p < p_end; p += PAF_THREADS) {
    const int i = p / nB, j = p - i * nB;  // 0-based
    const float v = score_ab<COEF>(s_A + (i + 1) * 3, s_B + (j + 1) * 3, s_mx, s_my, ws, Ht, Wt, s_xt, s_yt,
                                   mdl.inter_threshold, s_kmin, mdl.close_thr, mdl.close_score);
    sc[p] = v;
    accepted += (double)v > 1e-6;
  }
  if (n_accept) {  // connector 1 sizes its connection list with this
    for (int o = 32; o > 0; o >>= 1) accepted += __shfl_down(accepted, o);
    if ((tid & 63) == 0 && accepted) atomicAdd(&n_accept[b], accepted);
  }
}

// kernel 2b: sorted-greedy matching of one limb of one frame from its score matrix; grid (n_pairs, B)
__global__ __launch_bounds__(PAF_THREADS) void paf_greedy_kernel(const float* __restrict__ peaks, PhV2Dev mdl,
                                                                 const float* __restrict__ pair_scores,
                                                                 int* __restrict__ conn_idx,
                                                                 float* __restrict__ conn_score,
                                                                 int* __restrict__ n_conn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pair = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int P1 = mdl.max_peaks + 1;
  const int partA = mdl.pairs[2 * pair], partB = mdl.pairs[2 * pair + 1];
  const int nA = pir(peaks[((size_t)b * mdl.n_parts + partA) * P1 * 3]);
  const int nB = pir(peaks[((size_t)b * mdl.n_parts + partB) * P1 * 3]);
  int* out_n = n_conn + (size_t)b * mdl.n_pairs + pair;
  if (nA == 0 || nB == 0) {
    if (tid == 0) *out_n = 0;
    return;
  }
  unsigned long long* s_list = (unsigned long long*)smem;  // PAF_LIST_CAP accepted (score, i, j) keys
  unsigned long long* s_bestA = s_list + PAF_LIST_CAP;
  unsigned long long* s_bestB = s_bestA + P1;
  unsigned long long* s_sel = s_bestB + P1;  // selected keys, <= min(nA, nB) <= max_peaks
  unsigned char* s_occA = (unsigned char*)(s_sel + P1);
  unsigned char* s_occB = s_occA + P1;
  __shared__ unsigned long long s_best;
  __shared__ int s_count, s_nlist;
  for (int i = tid; i < P1; i += PAF_THREADS) {
    s_occA[i] = 0;
    s_occB[i] = 0;
  }
  if (tid == 0) {
    s_best = 0ull;
    s_count = 0;
    s_nlist = 0;
  }
  __syncthreads();
  const float* sc = pair_scores + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks * mdl.max_peaks;
  const int npairs = nA * nB;
  for (int p = tid; p < npairs; p += PAF_THREADS) {
    const float s = sc[p];
    if ((double)s > 1e-6) {
      const int i = p / nB, j = p - i * nB;
      const int slot = atomicAdd(&s_nlist, 1);
      if (slot < PAF_LIST_CAP)
        s_list[slot] = ((unsigned long long)__float_as_uint(s) << 32) | ((unsigned)(i + 1) << 16) | (unsigned)(j + 1);
    }
  }
  __syncthreads();

  // Sorted-greedy matching, in parallel.  Upstream sorts all accepted (score, i, j)
  // descending and takes a pair when both ends are unused.  A pair that is the largest
  // remaining key of BOTH its endpoints is always taken by that scan, so each round
  // selects every such pair at once (keys are unique); rounds repeat until nothing is
  // free.  The selected set equals the sequential result; it is emitted in descending
  // key order = the order upstream appends connections.
  const int nlist = s_nlist;
  const bool use_list = nlist <= PAF_LIST_CAP;  // else the rounds re-read the score matrix
  int* oi = conn_idx + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks * 2;
  float* os = conn_score + ((size_t)b * mdl.n_pairs + pair) * mdl.max_peaks;
  const int off = 3 * P1;
  auto for_each_free = [&](auto&& fn) {
    if (use_list) {
      for (int q = tid; q < nlist; q += PAF_THREADS) {
        const unsigned long long key = s_list[q];
        const int i = (int)((key >> 16) & 0xffff) - 1, j = (int)(key & 0xffff) - 1;
        if (!s_occA[i] && !s_occB[j]) fn(key, i, j);
      }
    } else {
      for (int p = tid; p < npairs; p += PAF_THREADS) {
        const float s = sc[p];
        if ((double)s > 1e-6) {
          const int i = p / nB, j = p - i * nB;
          if (!s_occA[i] && !s_occB[j])
            fn(((unsigned long long)__float_as_uint(s) << 32) | ((unsigned)(i + 1) << 16) | (unsigned)(j + 1), i, j);
        }
      }
    }
  };
  while (nlist > 0) {
    for (int i = tid; i < P1; i += PAF_THREADS) {
      s_bestA[i] = 0ull;
      s_bestB[i] = 0ull;
    }
    if (tid == 0) s_best = 0ull;  // "something was selected this round"
    __syncthreads();
    for_each_free([&](unsigned long long key, int i, int j) {
      atomicMax(&s_bestA[i], key);
      atomicMax(&s_bestB[j], key);
    });
    __syncthreads();
    for_each_free([&](unsigned long long key, int i, int j) {
      if (s_bestA[i] == key && s_bestB[j] == key) {
        s_sel[atomicAdd(&s_count, 1)] = key;
        s_best = 1ull;
      }
    });
    __syncthreads();
    // mark after the round so that the scans above saw one consistent state
    const int nsel = s_count;
    for (int q = tid; q < nsel; q += PAF_THREADS) {
      const unsigned long long key = s_sel[q];
      s_occA[(int)((key >> 16) & 0xffff) - 1] = 1;
      s_occB[(int)(key & 0xffff) - 1] = 1;
    }
    const bool progressed = s_best != 0ull;
    __syncthreads();
    if (!progressed) break;
  }
  const int nsel = s_count;
  for (int q = tid; q < nsel; q += PAF_THREADS) {
    const unsigned long long key = s_sel[q];
    int rank = 0;
    for (int r = 0; r < nsel; r++) rank += s_sel[r] > key;
    const int i = (int)((key >> 16) & 0xffff), j = (int)(key & 0xffff);
    oi[2 * rank] = partA * off + i * 3 + 2;
    oi[2 * rank + 1] = partB * off + j * 3 + 2;
    os[rank] = __uint_as_float((unsigned)(key >> 32));
  }
  if (tid == 0) *out_n = s_count;
}

// ---------------------------------------------------------------------------
// kernel 3: person assembly + filter + pack.  grid (B), block 64 (one wave)
//
// Upstream walks the limbs in order and, per limb, its connections in order.
// Within ONE limb the connections have pairwise distinct A peaks and distinct B
// peaks (greedy matching), look rows up by the A slot only and write the B slot,
// the count and the score, so they touch disjoint rows: they are processed in
// parallel, one lane per connection, and rows that must be appended get their
// positions from a prefix sum in connection order.  `s_owner[part][peak]` is a
// hint = lowest row that was ever given that peak in that slot; it is verified
// on use and a miss falls back to the exact linear "first row" search.
// Everything the serial dependency chain touches lives in LDS.
// ---------------------------------------------------------------------------
#define ASM_MAX_PARTS 32
#define ASM_NO_ROW 0xffff

__global__ __launch_bounds__(64) void assemble_kernel(const float* __restrict__ peaks, PhV2Dev mdl,
                                                      const int* __restrict__ conn_idx,
                                                      const float* __restrict__ conn_score,
                                                      const int* __restrict__ n_conn, float scale, int max_people,
                                                      int row_cap, int* __restrict__ n_people,
                                                      float* __restrict__ keypoints, float* __restrict__ scores,
                                                      int* __restrict__ subsets, int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int NP = mdl.n_parts, MP = mdl.max_peaks, P1 = MP + 1, off = 3 * P1;
  const int RW = NP + 1;  // uint16 words per row (last = part count)
  float* s_score = smem;                            // row_cap
  float* s_pk = s_score + row_cap;                  // NP * P1 peak scores ([part][0] = count)
  float* s_cs = s_pk + NP * P1;                     // MP connection scores
  int* s_ci = (int*)(s_cs + MP);                    // 2 * MP connection peak indices
  int* s_owner = s_ci + 2 * MP;                     // NP * P1 row hints
  unsigned short* s_mapA = (unsigned short*)(s_owner + NP * P1);  // P1: ear limb, peak i of A -> conn
  unsigned short* s_mapB = s_mapA + P1 + (P1 & 1);                // P1
  unsigned short* s_rows = s_mapB + P1 + (P1 & 1);                // row_cap * RW
  __shared__ int s_nrows;
  __shared__ int s_nc[64];
  const float* pk = peaks + (size_t)b * NP * off;
  for (int i = lane; i < NP * P1; i += 64) {
    const int part = i / P1, k = i - part * P1;
    s_pk[i] = k == 0 ? pk[part * off] : pk[part * off + k * 3 + 2];
    s_owner[i] = 0x7fffffff;
  }
  for (int i = lane; i < mdl.n_pairs; i += 64) s_nc[i] = n_conn[(size_t)b * mdl.n_pairs + i];
  if (lane == 0) s_nrows = 0;
  __syncthreads();
  // upstream peak index idx = part*off + i*3 + 2  ->  slot of (part, i) in the [NP][P1] tables
  auto slot_of = [&](int idx) -> int { return (idx / off) * P1 + (idx % off) / 3; };

  // exact "first row whose slot `part` holds idx" (verified hint, else linear search)
  auto first_row = [&](int part, int idx, int n) -> int {
    const int h = s_owner[slot_of(idx)];
    if (h == 0x7fffffff) return -1;  // never assigned anywhere
    if (h < n && s_rows[h * RW + part] == (unsigned short)idx) return h;
    for (int r = 0; r < n; r++)
      if (s_rows[r * RW + part] == (unsigned short)idx) return r;
    return -1;
  };
  auto write_row = [&](int r, int pa, int ia, int pb, int ib, int cnt, float sc) {
    unsigned short* row = s_rows + r * RW;
    for (int k = 0; k < NP; k++) row[k] = 0;
    row[pa] = (unsigned short)ia;
    atomicMin(&s_owner[slot_of(ia)], r);
    if (pb >= 0) {
      row[pb] = (unsigned short)ib;
      atomicMin(&s_owner[slot_of(ib)], r);
    }
    row[NP] = (unsigned short)cnt;
    s_score[r] = sc;
  };
  // ordered append: lanes with want=true get consecutive rows after `base`, in lane order
  auto alloc_rows = [&](bool want, int base, int* total) -> int {
    const unsigned long long m = __ballot(want);
    *total = __popcll(m);
    return base + __popcll(m & ((1ull << lane) - 1ull));
  };

  for (int pair = 0; pair < mdl.n_pairs; pair++) {
    const int partA = mdl.pairs[2 * pair], partB = mdl.pairs[2 * pair + 1];
    const int nA = pir(s_pk[partA * P1]), nB = pir(s_pk[partB * P1]);
    if (nA == 0 || nB == 0) {
      // add the non-empty side's peaks nobody holds yet as single-part people (in peak order)
      const int part = nA == 0 ? partB : partA;
      const int n = nA == 0 ? nB : nA;
      for (int i0 = 1; i0 <= n; i0 += 64) {
        const int i = i0 + lane;
        const int nrows = s_nrows;
        const int idx = part * off + i * 3 + 2;
        const bool want = i <= n && first_row(part, idx, nrows) < 0;
        int tot;
        const int r = alloc_rows(want, nrows, &tot);
        if (want) {
          if (r < row_cap) write_row(r, part, idx, -1, 0, 1, s_pk[part * P1 + i]);
          else *overflow = 1;
        }
        __syncthreads();
        if (lane == 0) s_nrows = nrows + tot < row_cap ? nrows + tot : row_cap;
        __syncthreads();
      }
      continue;
    }
    const int nc = s_nc[pair];
    if (nc == 0) continue;
    const int* gci = conn_idx + ((size_t)b * mdl.n_pairs + pair) * MP * 2;
    const float* gcs = conn_score + ((size_t)b * mdl.n_pairs + pair) * MP;
    for (int i = lane; i < 2 * nc; i += 64) s_ci[i] = gci[i];
    for (int i = lane; i < nc; i += 64) s_cs[i] = gcs[i];
    const bool ear = pair == mdl.ear_pair0 || pair == mdl.ear_pair1;
    if (ear && pair != 0)
      for (int i = lane; i < P1; i += 64) {
        s_mapA[i] = ASM_NO_ROW;
        s_mapB[i] = ASM_NO_ROW;
      }
    __syncthreads();
    if (ear && pair != 0) {
      // rows holding A (resp. B) of a connection and nothing in the other slot get the partner;
      // connections have distinct A's and B's, so each row matches at most one of them
      for (int c = lane; c < nc; c += 64) {
        s_mapA[(s_ci[2 * c] % off) / 3] = (unsigned short)c;
        s_mapB[(s_ci[2 * c + 1] % off) / 3] = (unsigned short)c;
      }
      __syncthreads();
      const int n = s_nrows;
      for (int r = lane; r < n; r += 64) {
        unsigned short* row = s_rows + r * RW;
        const int a = row[partA], bb = row[partB];
        if (a != 0 && bb == 0) {
          const int c = s_mapA[(a % off) / 3];
          if (c != ASM_NO_ROW) {
            row[partB] = (unsigned short)s_ci[2 * c + 1];
            atomicMin(&s_owner[slot_of(s_ci[2 * c + 1])], r);
          }
        } else if (bb != 0 && a == 0) {
          const int c = s_mapB[(bb % off) / 3];
          if (c != ASM_NO_ROW) {
            row[partA] = (unsigned short)s_ci[2 * c];
            atomicMin(&s_owner[slot_of(s_ci[2 * c])], r);
          }
        }
      }
      __syncthreads();
      continue;
    }
    for (int c0 = 0; c0 < nc; c0 += 64) {
      const int c = c0 + lane;
      const int nrows = s_nrows;
      int ia = 0, ib = 0, f = -1;
      float s = 0.f;
      if (c < nc) {
        ia = s_ci[2 * c];
        ib = s_ci[2 * c + 1];
        s = s_cs[c];
        if (pair != 0) f = first_row(partA, ia, nrows);
      }
      const bool want = c < nc && f < 0;
      int tot;
      const int r = alloc_rows(want, nrows, &tot);
      __syncthreads();  // all lookups done before any row changes
      if (c < nc) {
        if (f >= 0) {
          unsigned short* row = s_rows + f * RW;
          row[partB] = (unsigned short)ib;
          atomicMin(&s_owner[slot_of(ib)], f);
          row[NP] = row[NP] + 1;
          s_score[f] = s_score[f] + (s_pk[slot_of(ib)] + s);
        } else if (r < row_cap) {
          write_row(r, partA, ia, partB, ib, 2, (s_pk[slot_of(ia)] + s_pk[slot_of(ib)]) + s);
        } else {
          *overflow = 1;
        }
      }
      __syncthreads();
      if (lane == 0) s_nrows = nrows + tot < row_cap ? nrows + tot : row_cap;
      __syncthreads();
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
          scores[(size_t)b * max_people + o] = s_score[r] / denom;
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
  // (upstream's removePeopleBelowThresholds stops at POSE_MAX_PEOPLE: the count never exceeds the rows stored)
  if (lane == 0) n_people[b] = total < max_people ? total : max_people;
}

// ---------------------------------------------------------------------------
// kernel 3': the CUDA build's assembly (upstream pafPtrIntoVector + pafVectorIntoPeopleVector; oracle:
// paf_ptr_into_vector / paf_vector_into_people_vector).  grid (B), block MRG_THREADS.
//
// Upstream sorts EVERY accepted (limb, A_i, B_j) of the frame by (total, paf, limb, i, j) descending and walks the
// list once: create / extend / add-score / merge-if-disjoint, or ignore.  Two facts make that walk parallel:
//  * "ignore" is permanent.  People only grow (parts are never removed, merged people keep all parts), so a
//    connection that is ignored under the current state (its person already holds that part, or the two people
//    overlap) would be ignored at any later time too: it can be dropped as soon as it is seen to be dead.
//  * a live connection that is the largest live key touching BOTH of its clusters (a cluster = a person, or a
//    still unassigned peak) is reached by the sequential walk before anything else changes those clusters, so it
//    can be applied now; all such connections of a round touch pairwise disjoint clusters and commute.
// Rounds of (drop dead, find each cluster's best live key, apply the mutual bests) therefore reproduce the walk,
// including the fp32 order of every person's score additions (all operations on one cluster stay in key order).
// The walk's person order (= creation order, survivors of merges keep the lower index) is the order of the
// founding connections' keys: people carry that key and are ranked by it at the end.
// Live connections are kept in a compacted list in global memory once they fit it; until then the rounds scan
// the score matrix itself and kill dead entries in place (noise maps: ~10^5 accepted pairs per frame).
// ---------------------------------------------------------------------------
#define MRG_THREADS 1024
#define MRG_NONE 0xffffu

struct MrgEntry {
  unsigned total, paf, idx, cl;  // monotone keys of the two scores; idx = limb << 16 | i << 8 | j; clusters a | b << 16
};

__device__ __forceinline__ unsigned mrg_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving float -> uint
}

__global__ __launch_bounds__(MRG_THREADS) void assemble_merge_kernel(
    const float* __restrict__ peaks, PhV2Dev mdl, float* __restrict__ pair_scores, const int* __restrict__ n_accept,
    MrgEntry* __restrict__ lists, int list_cap, float scale, int max_people, int* __restrict__ n_people,
    float* __restrict__ keypoints, float* __restrict__ scores, int* __restrict__ subsets) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NP = mdl.n_parts, MP = mdl.max_peaks, P1 = MP + 1, NPK = NP * MP, CAP = NPK / 2 + 1;
  // LDS carve-up (8-byte items first)
  unsigned long long* s_best64 = (unsigned long long*)smem_raw;        // NPK: best live key of a cluster
  unsigned long long* s_ck64 = s_best64 + NPK;                         // CAP: founding key of a person
  unsigned* s_bestidx = (unsigned*)(s_ck64 + CAP);                     // NPK
  unsigned* s_ckidx = s_bestidx + NPK;                                 // CAP
  unsigned* s_mask = s_ckidx + CAP;                                    // CAP: parts a person holds
  float* s_score = (float*)(s_mask + CAP);                             // CAP
  float* s_pk = s_score + CAP;                                         // NP * P1 peak scores ([part][0] = count)
  int* s_sel = (int*)(s_pk + NP * P1);                                 // CAP: entries selected this round
  unsigned short* s_assigned = (unsigned short*)(s_sel + CAP);         // NPK: person of a peak
  unsigned short* s_root = s_assigned + NPK + (NPK & 1);               // CAP: founding A peak = cluster id
  unsigned char* s_rows = (unsigned char*)(s_root + CAP + (CAP & 1));  // CAP * NP: peak number per part, 0 = none
  unsigned char* s_cnt = s_rows + (size_t)CAP * NP;                    // CAP: parts counted (0 = merged away)
  __shared__ int s_nrows, s_nalive, s_nsel, s_moff[65], s_wcnt[MRG_THREADS / 64];

  const float* pk = peaks + (size_t)b * NP * P1 * 3;
  for (int i = tid; i < NP * P1; i += MRG_THREADS) {
    const int part = i / P1, k = i - part * P1;
    s_pk[i] = k == 0 ? pk[part * P1 * 3] : pk[(part * P1 + k) * 3 + 2];
  }
  for (int i = tid; i < NPK; i += MRG_THREADS) s_assigned[i] = MRG_NONE;
  if (tid == 0) {
    s_nrows = 0;
    s_nsel = 0;
  }
  __syncthreads();
  if (tid == 0) {  // entries of limb k live at pair_scores[k][p], p < nA * nB
    int o = 0;
    for (int k = 0; k < mdl.n_pairs; k++) {
      s_moff[k] = o;
      o += pir(s_pk[mdl.pairs[2 * k] * P1]) * pir(s_pk[mdl.pairs[2 * k + 1] * P1]);
    }
    s_moff[mdl.n_pairs] = o;
  }
  __syncthreads();
  float* mat = pair_scores + (size_t)b * mdl.n_pairs * MP * MP;
  MrgEntry* list0 = lists + (size_t)b * 2 * list_cap;
  MrgEntry* list1 = list0 + list_cap;

  // what the walk would do with connection (limb, i, j) in the current state: 0 ignore (permanently), 1 create,
  // 2 add B to A's person, 3 add A to B's person, 4 same person (score only), 5 merge
  auto classify = [&](int limb, int i, int j, int* ca, int* cb) -> int {
    const int partA = mdl.pairs[2 * limb], partB = mdl.pairs[2 * limb + 1];
    const int sa = partA * MP + i - 1, sb = partB * MP + j - 1;
    const unsigned pa = s_assigned[sa], pb = s_assigned[sb];
    *ca = pa == MRG_NONE ? sa : s_root[pa];
    *cb = pb == MRG_NONE ? sb : s_root[pb];
    if (pa == MRG_NONE && pb == MRG_NONE) return 1;
    if (pb == MRG_NONE) return s_rows[pa * NP + partB] == 0 ? 2 : 0;
    if (pa == MRG_NONE) return s_rows[pb * NP + partA] == 0 ? 3 : 0;
    if (pa == pb) return 4;
    return (s_mask[pa] & s_mask[pb]) == 0 ? 5 : 0;
  };
  // matrix scan: fn(limb, i, j, paf, &mat entry) for every entry the score kernel accepted and no round killed
  auto for_each_mat = [&](auto&& fn) {
    for (int limb = 0; limb < mdl.n_pairs; limb++) {
      const int n = s_moff[limb + 1] - s_moff[limb];
      if (n == 0) continue;
      const int nB = pir(s_pk[mdl.pairs[2 * limb + 1] * P1]);
      float* m = mat + (size_t)limb * MP * MP;
      for (int p = tid; p < n; p += MRG_THREADS) {
        const float v = m[p];
        if ((double)v > 1e-6) {
          const int i = p / nB;
          fn(limb, i + 1, p - i * nB + 1, v, m + p);
        }
      }
    }
  };
  auto total_of = [&](int limb, int i, int j, float paf) -> float {
    return (paf + 0.1f * s_pk[mdl.pairs[2 * limb] * P1 + i]) + 0.1f * s_pk[mdl.pairs[2 * limb + 1] * P1 + j];
  };
  // ordered append to a list: slots from one LDS counter, one atomic per wave
  auto push = [&](MrgEntry* dst, bool want, const MrgEntry& e) {
    const unsigned long long m = __ballot(want);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&s_nalive, __popcll(m));
    base = __shfl(base, 0);
    if (want) dst[base + __popcll(m & ((1ull << lane) - 1ull))] = e;
  };

  bool in_list = n_accept[b] <= list_cap;  // (block-uniform)
  MrgEntry* cur = list0;
  MrgEntry* nxt = list1;
  int n_cur = 0;
  if (in_list) {  // the accepted pairs fit the list: build it once, the matrix is not touched again
    if (tid == 0) s_nalive = 0;
    __syncthreads();
    for (int limb = 0; limb < mdl.n_pairs; limb++) {
      const int n = s_moff[limb + 1] - s_moff[limb];
      if (n == 0) continue;
      const int nB = pir(s_pk[mdl.pairs[2 * limb + 1] * P1]);
      const float* m = mat + (size_t)limb * MP * MP;
      for (int p0 = 0; p0 < n; p0 += MRG_THREADS) {
        const int p = p0 + tid;
        const float v = p < n ? m[p] : 0.f;
        const bool want = (double)v > 1e-6;
        MrgEntry e = {0u, 0u, 0u, 0u};
        if (want) {
          const int i = p / nB + 1, j = p - (i - 1) * nB + 1;
          e.total = mrg_key(total_of(limb, i, j, v));
          e.paf = mrg_key(v);
          e.idx = ((unsigned)limb << 16) | ((unsigned)i << 8) | (unsigned)j;
        }
        push(cur, want, e);
      }
    }
    __syncthreads();
    n_cur = s_nalive;
  }

  for (;;) {
    // ---- pass 1: drop dead connections, best (total, paf) key per cluster
    for (int i = tid; i < NPK; i += MRG_THREADS) {
      s_best64[i] = 0ull;
      s_bestidx[i] = 0u;
    }
    if (tid == 0) s_nalive = 0;
    __syncthreads();
    if (in_list) {
      for (int q0 = 0; q0 < n_cur; q0 += MRG_THREADS) {
        const int q = q0 + tid;
        MrgEntry e = {0u, 0u, 0u, 0u};
        bool live = false;
        if (q < n_cur) {
          e = cur[q];
          if (e.idx != 0xffffffffu) {  // (applied connections are consumed)
            int ca, cb;
            live = classify((int)(e.idx >> 16), (int)((e.idx >> 8) & 0xff), (int)(e.idx & 0xff), &ca, &cb) != 0;
            if (live) {
              e.cl = (unsigned)ca | ((unsigned)cb << 16);
              const unsigned long long key = ((unsigned long long)e.total << 32) | e.paf;
              atomicMax(&s_best64[ca], key);
              if (cb != ca) atomicMax(&s_best64[cb], key);
            }
          }
        }
        push(nxt, live, e);
      }
    } else {
      int live_n = 0;
      for_each_mat([&](int limb, int i, int j, float paf, float* at) {
        int ca, cb;
        if (classify(limb, i, j, &ca, &cb) == 0) {
          *at = 0.f;  // dead for good
          return;
        }
        const unsigned long long key = ((unsigned long long)mrg_key(total_of(limb, i, j, paf)) << 32) | mrg_key(paf);
        atomicMax(&s_best64[ca], key);
        if (cb != ca) atomicMax(&s_best64[cb], key);
        live_n++;
      });
      for (int o = 32; o > 0; o >>= 1) live_n += __shfl_down(live_n, o);
      if (lane == 0 && live_n) atomicAdd(&s_nalive, live_n);
    }
    __syncthreads();
    const int n_alive = s_nalive;
    if (n_alive == 0) break;
    if (in_list) {
      MrgEntry* t = cur;
      cur = nxt;
      nxt = t;
      n_cur = n_alive;
    }
    // ---- pass 2: among a cluster's best keys, the largest (limb, i, j)
    if (in_list) {
      for (int q = tid; q < n_cur; q += MRG_THREADS) {
        const MrgEntry e = cur[q];
        const unsigned long long key = ((unsigned long long)e.total << 32) | e.paf;
        const int ca = e.cl & 0xffff, cb = e.cl >> 16;
        if (s_best64[ca] == key) atomicMax(&s_bestidx[ca], e.idx);
        if (cb != ca && s_best64[cb] == key) atomicMax(&s_bestidx[cb], e.idx);
      }
    } else {
      for_each_mat([&](int limb, int i, int j, float paf, float* at) {
        int ca, cb;
        classify(limb, i, j, &ca, &cb);
        const unsigned long long key = ((unsigned long long)mrg_key(total_of(limb, i, j, paf)) << 32) | mrg_key(paf);
        const unsigned idx = ((unsigned)limb << 16) | ((unsigned)i << 8) | (unsigned)j;
        if (s_best64[ca] == key) atomicMax(&s_bestidx[ca], idx);
        if (cb != ca && s_best64[cb] == key) atomicMax(&s_bestidx[cb], idx);
      });
    }
    __syncthreads();
    // ---- pass 3: connections that are the best of both their clusters
    if (in_list) {
      for (int q = tid; q < n_cur; q += MRG_THREADS) {
        const MrgEntry e = cur[q];
        const unsigned long long key = ((unsigned long long)e.total << 32) | e.paf;
        const int ca = e.cl & 0xffff, cb = e.cl >> 16;
        if (s_best64[ca] == key && s_bestidx[ca] == e.idx && s_best64[cb] == key && s_bestidx[cb] == e.idx)
          s_sel[atomicAdd(&s_nsel, 1)] = q;
      }
    } else {
      for_each_mat([&](int limb, int i, int j, float paf, float* at) {
        int ca, cb;
        classify(limb, i, j, &ca, &cb);
        const unsigned long long key = ((unsigned long long)mrg_key(total_of(limb, i, j, paf)) << 32) | mrg_key(paf);
        const unsigned idx = ((unsigned)limb << 16) | ((unsigned)i << 8) | (unsigned)j;
        if (s_best64[ca] == key && s_bestidx[ca] == idx && s_best64[cb] == key && s_bestidx[cb] == idx)
          s_sel[atomicAdd(&s_nsel, 1)] = (int)(at - mat);
      });
    }
    __syncthreads();
    // ---- apply (disjoint clusters: no two selected connections touch the same person or peak)
    const int nsel = s_nsel;
    for (int t = tid; t < nsel; t += MRG_THREADS) {
      int limb, i, j;
      float paf;
      unsigned tkey, pkey;
      if (in_list) {
        const MrgEntry e = cur[s_sel[t]];
        limb = (int)(e.idx >> 16);
        i = (int)((e.idx >> 8) & 0xff);
        j = (int)(e.idx & 0xff);
        pkey = e.paf;
        tkey = e.total;
        paf = __uint_as_float((pkey & 0x80000000u) ? (pkey & 0x7fffffffu) : ~pkey);
        cur[s_sel[t]].idx = 0xffffffffu;
      } else {
        const int off = s_sel[t];
        limb = off / (MP * MP);
        const int p = off - limb * MP * MP, nB = pir(s_pk[mdl.pairs[2 * limb + 1] * P1]);
        i = p / nB + 1;
        j = p - (i - 1) * nB + 1;
        paf = mat[off];
        pkey = mrg_key(paf);
        tkey = mrg_key(total_of(limb, i, j, paf));
        mat[off] = 0.f;
      }
      const int partA = mdl.pairs[2 * limb], partB = mdl.pairs[2 * limb + 1];
      const int sa = partA * MP + i - 1, sb = partB * MP + j - 1;
      int ca, cb;
      const int what = classify(limb, i, j, &ca, &cb);
      const float scA = s_pk[partA * P1 + i], scB = s_pk[partB * P1 + j];
      if (what == 1) {
        const int r = atomicAdd(&s_nrows, 1);  // (any free row: the final order comes from the founding keys)
        for (int k = 0; k < NP; k++) s_rows[r * NP + k] = 0;
        s_rows[r * NP + partA] = (unsigned char)i;
        s_rows[r * NP + partB] = (unsigned char)j;
        s_mask[r] = (1u << partA) | (1u << partB);
        s_cnt[r] = 2;
        s_score[r] = (scA + scB) + paf;
        s_ck64[r] = ((unsigned long long)tkey << 32) | pkey;
        s_ckidx[r] = ((unsigned)limb << 16) | ((unsigned)i << 8) | (unsigned)j;
        s_root[r] = (unsigned short)sa;
        s_assigned[sa] = (unsigned short)r;
        s_assigned[sb] = (unsigned short)r;
      } else if (what == 2 || what == 3) {
        const int r = what == 2 ? s_assigned[sa] : s_assigned[sb];
        const int part2 = what == 2 ? partB : partA, n2 = what == 2 ? j : i;
        s_rows[r * NP + part2] = (unsigned char)n2;
        s_mask[r] |= 1u << part2;
        s_cnt[r] = s_cnt[r] + 1;
        s_score[r] = s_score[r] + ((what == 2 ? scB : scA) + paf);
        s_assigned[what == 2 ? sb : sa] = (unsigned short)r;
      } else if (what == 4) {
        const int r = s_assigned[sa];
        s_score[r] = s_score[r] + paf;
      } else if (what == 5) {
        int r1 = s_assigned[sa], r2 = s_assigned[sb];
        // the walk keeps the lower index = the person founded earlier = the larger founding key
        const bool first_is_1 = s_ck64[r1] > s_ck64[r2] || (s_ck64[r1] == s_ck64[r2] && s_ckidx[r1] > s_ckidx[r2]);
        if (!first_is_1) {
          const int t2 = r1;
          r1 = r2;
          r2 = t2;
        }
        for (int k = 0; k < NP; k++) {
          const int n2 = s_rows[r2 * NP + k];
          if (n2 != 0) {
            s_rows[r1 * NP + k] = (unsigned char)n2;
            s_assigned[k * MP + n2 - 1] = (unsigned short)r1;
          }
        }
        s_mask[r1] |= s_mask[r2];
        s_cnt[r1] = s_cnt[r1] + s_cnt[r2];
        s_score[r1] = s_score[r1] + (s_score[r2] + paf);
        s_cnt[r2] = 0;
        s_mask[r2] = 0;
      }
    }
    __syncthreads();
    if (tid == 0) s_nsel = 0;
    // the live connections fit the list now: compact them out of the matrix once and go on from the list
    if (!in_list && n_alive <= list_cap) {
      if (tid == 0) s_nalive = 0;
      __syncthreads();
      for (int limb = 0; limb < mdl.n_pairs; limb++) {
        const int n = s_moff[limb + 1] - s_moff[limb];
        if (n == 0) continue;
        const int nB = pir(s_pk[mdl.pairs[2 * limb + 1] * P1]);
        const float* m = mat + (size_t)limb * MP * MP;
        for (int p0 = 0; p0 < n; p0 += MRG_THREADS) {
          const int p = p0 + tid;
          const float v = p < n ? m[p] : 0.f;
          const bool want = (double)v > 1e-6;
          MrgEntry e = {0u, 0u, 0u, 0u};
          if (want) {
            const int i = p / nB + 1, j = p - (i - 1) * nB + 1;
            e.total = mrg_key(total_of(limb, i, j, v));
            e.paf = mrg_key(v);
            e.idx = ((unsigned)limb << 16) | ((unsigned)i << 8) | (unsigned)j;
          }
          push(cur, want, e);
        }
      }
      __syncthreads();
      n_cur = s_nalive;
      in_list = true;
    }
    __syncthreads();
  }

  // ---- person order of the walk: descending founding key among the people that were not merged away
  const int n = s_nrows;
  unsigned short* s_order = (unsigned short*)s_best64;  // (the round tables are free now)
  int* s_total = &s_nalive;
  __syncthreads();
  for (int r = tid; r < n; r += MRG_THREADS) {
    if (s_cnt[r] == 0) continue;
    const unsigned long long k64 = s_ck64[r];
    const unsigned kid = s_ckidx[r];
    int rank = 0;
    for (int q = 0; q < n; q++)
      rank += s_cnt[q] != 0 && (s_ck64[q] > k64 || (s_ck64[q] == k64 && s_ckidx[q] > kid));
    s_order[rank] = (unsigned short)r;
  }
  if (tid == 0) {
    int live = 0;
    for (int r = 0; r < n; r++) live += s_cnt[r] != 0;
    s_nsel = live;
  }
  __syncthreads();
  const int n_live = s_nsel;
  // ---- removePeopleBelowThresholds + peopleVectorToPeopleArray (as in assemble_kernel)
  const float denom = (float)(NP + mdl.n_pairs);
  const int off = 3 * P1;
  int total = 0;
  for (int pass = 0; pass < 2; pass++) {  // pass 1 = maximizePositives retry
    total = 0;
    for (int q0 = 0; q0 < n_live; q0 += MRG_THREADS) {
      const int q = q0 + tid;
      bool ok = false;
      int r = 0;
      if (q < n_live) {
        r = s_order[q];
        int counter = s_cnt[r];
        if (pass == 0 && NP == 25) {
          int feet = 0;
          for (int k = 19; k < 25; k++) feet += s_rows[r * NP + k] > 0;
          counter -= feet;
        }
        ok = counter >= mdl.min_subset_cnt && s_score[r] / (float)counter >= mdl.min_subset_score;
      }
      const unsigned long long m = __ballot(ok);
      if (lane == 0) s_wcnt[wave] = __popcll(m);
      __syncthreads();
      int before = 0, all = 0;
      for (int w = 0; w < MRG_THREADS / 64; w++) {
        before += w < wave ? s_wcnt[w] : 0;
        all += s_wcnt[w];
      }
      if (ok) {
        const int o = total + before + __popcll(m & ((1ull << lane) - 1ull));
        if (o < max_people) {
          float* kp = keypoints + ((size_t)b * max_people + o) * NP * 3;
          for (int k = 0; k < NP; k++) {
            const int pn = s_rows[r * NP + k];
            if (pn > 0) {
              const float* src = pk + ((size_t)k * P1 + pn) * 3;
              kp[k * 3 + 0] = src[0] * scale;
              kp[k * 3 + 1] = src[1] * scale;
              kp[k * 3 + 2] = src[2];
            } else {
              kp[k * 3 + 0] = 0.f;
              kp[k * 3 + 1] = 0.f;
              kp[k * 3 + 2] = 0.f;
            }
          }
          scores[(size_t)b * max_people + o] = s_score[r] / denom;
          if (subsets) {
            int* sb = subsets + ((size_t)b * max_people + o) * (NP + 1);
            for (int k = 0; k < NP; k++) {
              const int pn = s_rows[r * NP + k];
              sb[k] = pn > 0 ? k * off + pn * 3 + 2 : 0;
            }
            sb[NP] = s_cnt[r];
          }
        }
      }
      total += all;
      __syncthreads();
    }
    if (total > 0) break;
  }
  if (tid == 0) n_people[b] = total < max_people ? total : max_people;   // as upstream: never more than the rows stored
  (void)s_total;
}

// ---------------------------------------------------------------------------
// optional: materialise upsampled maps (datum.poseHeatMaps export only -- the
// decode itself never needs them).  grid-stride over [nc][Ht][Wt] of one frame.
// ---------------------------------------------------------------------------
__global__ void upsample_maps_kernel(const float* __restrict__ planes, int hs, int ws, int Ht, int Wt, int nc,
                                     int n_heat_rel, int mode, float* __restrict__ out) {
  const size_t total = (size_t)nc * Ht * Wt;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % Wt;
    const size_t t = i / Wt;
    const int y = t % Ht;
    const int c = t / Ht;
    // channels [0, n_heat_rel) of the window are part/background maps, the rest PAFs
    out[i] = heatmap_scale_value(bicubic_at(planes + (size_t)c * hs * ws, hs, ws, Ht, Wt, x, y), c < n_heat_rel, mode);
  }
}

int ph_launch_upsample_maps(const float* planes, int hs, int ws, int Ht, int Wt, int nc, int n_heat_rel, int mode,
                            float* out, hipStream_t stream) {
  const size_t total = (size_t)nc * Ht * Wt;
  int blocks = (int)((total + 255) / 256);
  blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(upsample_maps_kernel, dim3(blocks), dim3(256), 0, stream, planes, hs, ws, Ht, Wt, nc, n_heat_rel,
                     mode, out);
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
  s->nms_keys = (size_t)B * m.n_parts * PH_V2_NMS_SPLIT * m.max_peaks * sizeof(int);
  s->nms_cnt = (size_t)B * m.n_parts * PH_V2_NMS_SPLIT * sizeof(int);
  s->n_accept = (size_t)B * sizeof(int);
  s->mrg_lists = m.connector == 1 ? (size_t)B * 2 * PH_V2_MRG_LIST_CAP * sizeof(MrgEntry) : 256;
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
  if (!(scale > 0.f)) {
    ph_set_error("decode_v2: keypoint scale %g must be positive", (double)scale);
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
  // upstream poseExtractorCaffe.cpp: nmsOffset = float(0.5 / double(mScaleNetToOutput)), so that a peak sits at
  // the pixel centre (+0.5) in OUTPUT-image units after the keypoints are multiplied by the scale
  const float nms_offset = (float)(0.5 / (double)scale);
  const int n_split = PH_V2_NMS_SPLIT;
  const int rows_per_split = (((Ht + n_split - 1) / n_split) + NMS_BAND - 1) / NMS_BAND * NMS_BAND;
  const int n_groups = (Wt - 1 + NMS_COLS - 1) / NMS_COLS;          // 62 output columns per wave and group
  static const int nms_g_env = [] {  // A/B override: only the instantiated widths 1, 2, 4 are taken, anything else is ignored
    const char* e = getenv("POSEHIP_NMS_G");
    const int v = e ? atoi(e) : 0;
    return v == 1 || v == 2 || v == 4 ? v : 0;
  }();
  // two columns per lane from 8 groups on: the two independent cubics pair up into packed fp32 operations
  // (0.29 vs 0.33 ms per 32 frames at 656 columns; four per lane is no faster).  POSEHIP_NMS_G overrides.
  const int nms_g = nms_g_env ? nms_g_env : (n_groups <= 7 ? 1 : (n_groups <= 32 ? 2 : 4));
  const int nms_waves = (n_groups + nms_g - 1) / nms_g;
  if (nms_waves > 16) {
    ph_set_error("decode_v2: target width %d beyond the NMS kernel's %d columns", Wt, 64 * NMS_COLS);
    return -1;
  }
  const size_t lds1 = (size_t)hs * ws * 4 + (size_t)(rows_per_split + 2) * sizeof(TapsP) +
                      ((size_t)(rows_per_split + 3) + (size_t)(NMS_BAND / 2) * (Wt / 2 + 1) + m.max_peaks) * 4;
  const size_t lds1b = (size_t)hs * ws * 4;
  // PAF planes in LDS as cubic coefficient tables (16 B per source pixel and channel) when they fit, else raw
  // (POSEHIP_PAF_RAW=1 forces the raw form for A/B runs)
  static const bool paf_raw = [] { const char* e = getenv("POSEHIP_PAF_RAW"); return e && atoi(e) != 0; }();
  const size_t lds2_rest = (size_t)2 * (m.max_peaks + 1) * 3 * 4 + (size_t)(Wt + Ht + 1) * sizeof(TapsP) + 64;
  const size_t lds2g = (size_t)PAF_LIST_CAP * 8 + (size_t)3 * (m.max_peaks + 1) * 8 + 2 * (m.max_peaks + 1) + 32;
  // ranges per limb in the scoring kernel: enough blocks to level a few heavy limbs out over the chip
  int paf_sub = 4096 / (m.n_pairs * B > 0 ? m.n_pairs * B : 1);
  paf_sub = paf_sub < 1 ? 1 : (paf_sub > 8 ? 8 : paf_sub);
  static const int paf_sub_env = [] {  // A/B override (1..8), anything else is ignored
    const char* e = getenv("POSEHIP_PAF_SUB");
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 8 ? v : 0;
  }();
  if (paf_sub_env) paf_sub = paf_sub_env;
  const size_t lim = 160 * 1024 - 1024;
  const bool paf_coef = !paf_raw && lds2_rest + (size_t)2 * hs * ws * 16 <= lim;
  const size_t lds2 = lds2_rest + (size_t)2 * hs * ws * (paf_coef ? 16 : 4);
  const int row_cap = PH_V2_ROW_CAP;
  const size_t lds3 = ((size_t)row_cap + 2 * (size_t)m.n_parts * (m.max_peaks + 1) + 3 * (size_t)m.max_peaks) * 4 +
                      (size_t)2 * (m.max_peaks + 2) * 2 + (size_t)row_cap * (m.n_parts + 1) * 2;
  const size_t npk = (size_t)m.n_parts * m.max_peaks, mcap = npk / 2 + 1;
  const size_t lds3m = 8 * (npk + mcap) + 4 * (npk + 4 * mcap + (size_t)m.n_parts * (m.max_peaks + 1)) +
                       2 * (npk + 1 + mcap + 1) + mcap * m.n_parts + mcap + 64;
  if (m.connector == 1 && (m.max_peaks > 255 || npk >= 65535 || lds3m > lim)) {
    ph_set_error("decode_v2: connector 1 needs max_peaks <= 255 and %zu B of LDS (have %zu)", lds3m, lim);
    return -1;
  }
  if (m.connector != 0 && m.connector != 1) {
    ph_set_error("decode_v2: connector %d (0 = createPeopleVector, 1 = pafVectorIntoPeopleVector)", m.connector);
    return -1;
  }
  if (lds1 > lim || lds1b > lim || lds2 > lim || lds2g > lim || lds3 > lim || hs > 65535 || ws > 65535) {
    ph_set_error("decode_v2: LDS budget exceeded (%zu/%zu/%zu B) for net output %dx%d target %dx%d", lds1, lds2,
                 lds3, hs, ws, Ht, Wt);
    return -1;
  }
  static int attr[4][PH_MAX_DEVICES];
  static int attr_c[5][PH_MAX_DEVICES];
  static int attr_m[PH_MAX_DEVICES];
  static int attr_n[3][PH_MAX_DEVICES];
  if (ph_ensure_lds((const void*)nms_detect_cols_kernel<1, true>, (int)lim, attr_c[0]) ||
      ph_ensure_lds((const void*)nms_detect_cols_kernel<2, true>, (int)lim, attr_c[1]) ||
      ph_ensure_lds((const void*)nms_detect_cols_kernel<4, true>, (int)lim, attr_c[2]) ||
      ph_ensure_lds((const void*)nms_detect_cols_kernel<1, false>, (int)lim, attr_n[0]) ||
      ph_ensure_lds((const void*)nms_detect_cols_kernel<2, false>, (int)lim, attr_n[1]) ||
      ph_ensure_lds((const void*)nms_detect_cols_kernel<4, false>, (int)lim, attr_n[2]) ||
      ph_ensure_lds((const void*)nms_refine_kernel, (int)lim, attr[1]) ||
      ph_ensure_lds((const void*)paf_score_kernel<false>, (int)lim, attr[2]) ||
      ph_ensure_lds((const void*)paf_score_kernel<true>, (int)lim, attr_c[3]) ||
      ph_ensure_lds((const void*)paf_greedy_kernel, (int)lim, attr_c[4]) ||
      ph_ensure_lds((const void*)assemble_kernel, (int)lim, attr[3]) ||
      ph_ensure_lds((const void*)assemble_merge_kernel, (int)lim, attr_m))
    return -1;
  PhV2Sizes sz;
  ph_v2_workspace_sizes(m, B, max_people, &sz);
  PH_CHECK(hipMemsetAsync(w.peaks, 0, sz.peaks, stream));
  PH_CHECK(hipMemsetAsync(w.overflow, 0, sizeof(int), stream));
  if (m.connector == 1) PH_CHECK(hipMemsetAsync(w.n_accept, 0, sz.n_accept, stream));
  if (ev) PH_CHECK(hipEventRecord(ev[0], stream));
  {
    // finite planes (every plane a network produces), then the stripes that kernel marked as not finite
    auto kern = nms_g == 1 ? nms_detect_cols_kernel<1, true>
                           : (nms_g == 2 ? nms_detect_cols_kernel<2, true> : nms_detect_cols_kernel<4, true>);
    hipLaunchKernelGGL(kern, dim3(m.n_parts, B, n_split), dim3(64 * nms_waves), lds1, stream, net_out, C, hs, ws, Ht,
                       Wt, m.nms_threshold, m.max_peaks, m.n_parts, n_split, rows_per_split, n_groups, w.nms_keys,
                       w.nms_cnt);
    auto kern_n = nms_g == 1 ? nms_detect_cols_kernel<1, false>
                             : (nms_g == 2 ? nms_detect_cols_kernel<2, false> : nms_detect_cols_kernel<4, false>);
    hipLaunchKernelGGL(kern_n, dim3(m.n_parts, B, n_split), dim3(64 * nms_waves), lds1, stream, net_out, C, hs, ws,
                       Ht, Wt, m.nms_threshold, m.max_peaks, m.n_parts, n_split, rows_per_split, n_groups,
                       w.nms_keys, w.nms_cnt);
  }
  hipLaunchKernelGGL(nms_refine_kernel, dim3(m.n_parts, B), dim3(128), lds1b, stream, net_out, C, hs, ws, Ht, Wt,
                     m.max_peaks, m.n_parts, n_split, nms_offset, (const int*)w.nms_keys, (const int*)w.nms_cnt,
                     w.peaks);
  if (ev) PH_CHECK(hipEventRecord(ev[1], stream));
  hipLaunchKernelGGL(paf_coef ? paf_score_kernel<true> : paf_score_kernel<false>, dim3(m.n_pairs, B, paf_sub),
                     dim3(PAF_THREADS), lds2, stream, net_out, C, hs, ws, Ht, Wt, (const float*)w.peaks, m,
                     w.pair_scores, paf_sub, m.connector == 1 ? w.n_accept : (int*)nullptr);
  if (m.connector == 0) {
    hipLaunchKernelGGL(paf_greedy_kernel, dim3(m.n_pairs, B), dim3(PAF_THREADS), lds2g, stream, (const float*)w.peaks,
                       m, (const float*)w.pair_scores, w.conn_idx, w.conn_score, w.n_conn);
    if (ev) PH_CHECK(hipEventRecord(ev[2], stream));
    hipLaunchKernelGGL(assemble_kernel, dim3(B), dim3(64), lds3, stream, (const float*)w.peaks, m,
                       (const int*)w.conn_idx, (const float*)w.conn_score, (const int*)w.n_conn, scale, max_people,
                       row_cap, w.n_people, w.keypoints, w.scores, w.subsets, w.overflow);
  } else {
    // CUDA-build connector: no per-limb matching; one globally ordered walk over all accepted pairs
    PH_CHECK(hipMemsetAsync(w.n_conn, 0, sz.n_conn, stream));  // (ph_decode_v2_connections then reports none)
    if (ev) PH_CHECK(hipEventRecord(ev[2], stream));
    hipLaunchKernelGGL(assemble_merge_kernel, dim3(B), dim3(MRG_THREADS), lds3m, stream, (const float*)w.peaks, m,
                       w.pair_scores, (const int*)w.n_accept, (MrgEntry*)w.mrg_lists, PH_V2_MRG_LIST_CAP, scale,
                       max_people, w.n_people, w.keypoints, w.scores, w.subsets);
  }
  if (ev) PH_CHECK(hipEventRecord(ev[3], stream));
  PH_CHECK(hipGetLastError());
  return 0;
}
