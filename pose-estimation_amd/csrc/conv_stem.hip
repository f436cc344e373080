// The VGG stem as ONE persistent kernel for gfx950: conv1_1 (3x3, 8-channel frame -> 64, ReLU) + conv1_2 (3x3,
// 64 -> 64, bias + activation) + the 2x2/2 max pool that follows, for networks whose first two layers have that
// shape (BODY_25 / COCO: openpose/opencv/skeleton.py:74-75 `net.forward()`,
// openpose/native/python/skeleton.py:447 `emplaceAndPop` -- the layers Caffe / cv2.dnn run first).
//
// Why its own kernel.  The general halo kernel's stem form (conv_mfma.hip, STEM = true) works on 16x16 tiles with
// two blocks per CU and streams the 64 x 576 weight matrix through LDS for EVERY tile (one kernel row at a time, a
// barrier per row); its wave tiles are 64 pixels x 32 channels: 6 fragment reads per 8 MFMAs = 3/4 of the LDS read
// rate at full MFMA rate.  It ran at 35 % MFMA occupancy, the blocks waiting for weight rows and for their frame
// patch almost half of their life.  Here:
//   * one block of 8 waves per CU, PERSISTENT: the whole weight matrix (72 KB bf16) is brought into LDS once per
//     block and stays; tiles are handed out by an atomic counter (a block that starts late, because another stream's
//     kernel held its CU, simply takes fewer tiles);
//   * a tile is 16 rows x 32 columns = 512 output pixels, a wave owns two image rows: 64 pixels x 64 channels,
//     8 fragment reads per 16 MFMAs (half the LDS read rate at full MFMA rate), and the 2x2 pool stays inside the
//     wave (row pairs in one lane's registers, column pairs in neighbouring lanes);
//   * the 18 x 34 x 64-channel input patch of conv1_2 is computed by conv1_1 straight into LDS (never in memory),
//     from the 20 x 36 frame pixels under it, staged in LDS once per tile (channels 0..3, 5.8 KB): the next tile's
//     are requested before the K loop of the current tile starts and written to LDS after it;
//   * the K loop (18 steps of 16 MFMAs per wave) has no barrier and no global load: two barriers per tile in all
//     (patch complete / patch free).
// LDS: patch 2 x 41,472 B (two 32-channel planes, rows 36 pixels apart) + weights 73,728 B + bias/slope + frame
// pixels 5,760 B = 159.4 KB of the CU's 160.
// Accumulation order per output = the halo kernel's (chunk-major, taps inside), so results are bit-identical to it.
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define ST_TH 16                       // tile rows
#define ST_TW 32                       // tile columns
#define ST_PC (ST_TW + 2)              // patch columns (34)
#define ST_PW 36                       // patch row stride in pixels: 32 + 4, one row further flips the swizzle bit
#define ST_NPX ((ST_TH + 2) * ST_PC)   // 612 patch pixels
#define ST_NRB ((ST_NPX + 15) / 16)    // 39 row blocks of 16 patch pixels
#define ST_PLANE ((ST_TH + 2) * ST_PW * 64)   // 41,472 B per 32-channel plane
#define ST_WBYTES (18 * 4096)          // 18 (tap, chunk) slices of 64 channels x 64 B
#define ST_SROW 144                    // staged pooled pixel: 64 channels bf16 + 16 B (bank spread)
#define ST_FH (ST_TH + 4)              // frame pixels under the patch: 20 rows ..
#define ST_FW (ST_TW + 4)              // .. x 36 columns
#define ST_FBYTES (ST_FH * ST_FW * 8)  // 5,760 B: channels 0..3 (8 bytes) of each
#define ST_LDS (2 * ST_PLANE + ST_WBYTES + 3 * 256 + 16 + ST_FBYTES)

#ifdef ST_STAMP
// Diagnostic build only (tools/stem512_stamp.py): cycles per phase, summed over the tiles of every block's wave 1.
__device__ unsigned long long st_dbg[16];
extern "C" int ph_debug_stem512_cycles(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(st_dbg), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(st_dbg), 16 * sizeof(unsigned long long));
}
#define ST_T(var) const long long var = __builtin_readcyclecounter()
#else
#define ST_T(var)
#endif

__device__ static inline float st_xor1(float v) {  // value of the neighbouring lane (lane ^ 1)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

__global__ __launch_bounds__(512, 1) void conv_stem512_kernel(const PhConvParams p, int ntiles, int tiles_x,
                                                              int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const wl = lds + 2 * ST_PLANE;
  float* const s_bias = (float*)(wl + ST_WBYTES);
  float* const s_slope = s_bias + 64;
  float* const s_bias0 = s_slope + 64;
  int* const s_tn = (int*)(s_bias0 + 64);
  char* const fpl = (char*)(s_tn + 4);   // the tile's frame pixels, channels 0..3: [ST_FH][ST_FW] x 8 B

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int Wp = p.in_Wp, Hp = p.in_Hp, pi = p.in_pad;
  int* const ctr = p.tile_ctr + p.tile_par;

  // ---- once per block: weights, biases, first tile numbers ---------------------------------------------------
  {
    u32x4 wv[9];  // 64 rows x 72 16-byte units = 9 per thread, all requested before the first is stored
#pragma unroll
    for (int q = 0; q < 9; q++) {
      const int v = tid + q * 512;
      const int n = v / 72, kv = v - n * 72;  // unit kv of weight row n: k = kv * 8 = tap * 64 + chunk * 32 + ..
      wv[q] = *(const u32x4*)(p.wt + (size_t)n * p.Kpad + kv * 8);
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
      const int v = tid + q * 512;
      const int n = v / 72, kv = v - n * 72;
      *(u32x4*)(wl + (kv >> 2) * 4096 + n * 64 + (((kv & 3) ^ (((n >> 2) & 1) << 1)) << 4)) = wv[q];
    }
  }
  if (tid < 64) {
    s_bias[tid] = p.bias[tid];
    s_slope[tid] = p.slope[tid];
    s_bias0[tid] = p.bias0[tid];
  }
  if (tid == 0) {
    if (blockIdx.x == 0) p.tile_ctr[p.tile_par ^ 1] = 0;  // the next launch's counter (this stream runs them in order)
    s_tn[0] = (int)gridDim.x + atomicAdd(ctr, 1);
  }
  // conv1_1 weights: 4 channel blocks x 3 K steps (K = 9 taps x 8 channels, padded to 96 with zero rows): 48
  // registers, held for the life of the block now that the frame pixels no longer wait in 60 registers across the K
  // loop (they were fetched again for every tile before: 12 more vector-memory instructions per wave and tile).
  bf16x8 wf[4][3];
  auto load_wf = [&]() {
#pragma unroll
    for (int nb = 0; nb < 4; nb++)
#pragma unroll
      for (int ks = 0; ks < 3; ks++) wf[nb][ks] = *(const bf16x8*)(p.wt0 + (nb * 16 + fr) * p.Kpad0 + ks * 32 + fq * 8);
  };
  load_wf();

  // ---- per-lane constants ------------------------------------------------------------------------------------
  // conv1_1 producer: wave w computes patch pixels j = 80 w .. 80 w + 79 (row blocks 5 w .. 5 w + 4); indices past
  // the last pixel (wave 7) repeat it: same inputs, same value, same address -- no lane is ever switched off
  // (row, column) of patch pixel j = (5 wave + u) * 16 + fr, clamped to the last pixel.  Recomputed where it is used
  // (a handful of VALU operations per tile) instead of being held in registers across the K loop; the empty asm
  // keeps the compiler from hoisting the results back out of the tile loop.
  auto patch_rc = [&](int u, int& r, int& c) {
    int f = fr;
    asm volatile("" : "+v"(f));
    const int j = (wave * 5 + u) * 16 + f;
    const int jc = j < ST_NPX ? j : ST_NPX - 1;
    r = (int)(((unsigned)jc * 1928u) >> 16);   // jc / 34 for jc < 2048
    c = jc - r * ST_PC;
  };
  // conv1_2 consumer: fragment offsets of the lane inside a patch row / a weight slice
  int o0[3];
#pragma unroll
  for (int kw = 0; kw < 3; kw++) o0[kw] = (fr + kw) * 64 + ((fq * 16) ^ ((((fr + kw) >> 2) & 1) << 5));
  const int wrd = fr * 64 + ((fq * 16) ^ ((fr & 4) << 3));
  const char* const pbase = lds + wave * (2 * ST_PW * 64);
  // staging of the pooled rows: inside the patch pixels this wave itself rewrites for the next tile
  char* const stg = lds + ((80 * wave / ST_PC) * ST_PW + (80 * wave) % ST_PC) * 64;

  const int tiles_per_image = tiles_x * tiles_y;
  // The frame pixels under a tile's patch (20 x 36, the first four of their 8 channels: ph_stem512_applies) come
  // into LDS ONCE per tile -- two 8-byte loads per lane, requested before the K loop of the previous tile, written
  // to `fpl` after it -- and conv1_1's fragments are read from there.  (Round 3's first form read every tap of every
  // patch pixel from global memory straight into the fragments: 15 scattered 16-byte loads per lane and tile, issued
  // by all 8 waves at once -- 4.6 thousand cycles of a tile's 24.5 thousand went into issuing them, whatever their
  // size: the CU's vector-memory instruction rate, not its bytes)
  bf16x4 fpx[2];
  int nx_b = 0, nx_y0 = 0, nx_x0 = 0;  // image and origin of the tile whose frame pixels were requested last
  auto issue_loads = [&](int t) {
    const int b = t / tiles_per_image;
    const int r2 = t - b * tiles_per_image;
    const int ty = r2 / tiles_x, tx = r2 - ty * tiles_x;
    nx_b = b;
    nx_y0 = ty * ST_TH;
    nx_x0 = tx * ST_TW;
    // frame pixel under tap (0,0) of patch pixel (0,0): image (ty*16 - 2, tx*32 - 2)
    const int Q0 = (b * Hp + ty * ST_TH - 2 + pi) * Wp + tx * ST_TW - 2 + pi;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      int idx = tid + q * 512;
      idx = idx < ST_FH * ST_FW ? idx : ST_FH * ST_FW - 1;   // (past the last pixel: it again, no lane switched off)
      const int fy = (int)(((unsigned)idx * 1821u) >> 16);   // idx / 36 for idx < 720
      int Q = Q0 + fy * Wp + (idx - fy * ST_FW);
      Q = Q < 0 ? 0 : (Q < p.NQ ? Q : p.NQ - 1);
      fpx[q] = *(const bf16x4*)((const char*)p.in + (size_t)(unsigned)Q * 16u);
    }
  };
  auto store_frame = [&]() {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      int idx = tid + q * 512;
      idx = idx < ST_FH * ST_FW ? idx : ST_FH * ST_FW - 1;
      *(bf16x4*)(fpl + idx * 8) = fpx[q];
    }
  };
  // conv1_1's operand of K step ks for this lane: K group fq of 8 = the 8 channels of tap ks * 4 + fq, of which
  // 4..7 are the frame buffer's padding (zero); taps 9..11 carry zero weights: any finite pixel serves
  int ftap[3];
#pragma unroll
  for (int ks = 0; ks < 3; ks++) {
    int tp = ks * 4 + fq;
    tp = tp < 8 ? tp : 8;
    const int th = (tp * 11) >> 5;  // tp / 3 for tp <= 8
    ftap[ks] = (th * ST_FW + (tp - th * 3)) * 8;
  }

  // pooled rows of the previous tile, stored one phase later (see P3)
  u32x4 pend[2];
  int pend_b = 0, pend_row = 0x7fffffff, pend_col0 = 0;
  const PhDst ds0 = p.dst[0];  // (nearly always the only destination: kept in scalar registers)
  auto flush_rows = [&]() {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int v = lane + 64 * q;
      const int pcol = pend_col0 + (v >> 3);
      if (pend_row < (p.H >> 1) && pcol < (p.W >> 1)) {
        // (element offsets fit 31 bits: ph_stem512_applies checks the destination sizes)
        const unsigned off0 = (unsigned)(((pend_b * ds0.Hp + pend_row) * ds0.Wp + pcol) * ds0.Cs + (v & 7) * 8 + ds0.org);
        *(u32x4*)((char*)ds0.base + (size_t)off0 * 2u) = pend[q];
        for (int d = 1; d < p.n_dst; d++) {
          const PhDst ds = p.dst[d];
          *(u32x4*)(ds.base + (size_t)ds.org + ((size_t)(pend_b * ds.Hp + pend_row) * ds.Wp + pcol) * ds.Cs + (v & 7) * 8) =
              pend[q];
        }
      }
    }
  };

#ifdef ST_STAMP
  long long sacc_[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  int t = blockIdx.x;
  issue_loads(t < ntiles ? t : ntiles - 1);
  store_frame();
  __syncthreads();  // weights, biases, s_tn[0] and the first tile's frame pixels are in LDS
  int it = 0;
  while (t < ntiles) {
    const int tb = nx_b, ty0 = nx_y0, tx0 = nx_x0;  // (computed when this tile's frame pixels were requested)

    ST_T(c0_);
#ifdef ST_STAMP
    long long su_[4] = {0, 0, 0, 0};
#endif
    // ---- P1: conv1_1 -> the 18 x 34 x 64 patch in LDS ---------------------------------------------------------
    {
      const float relu0_floor = p.act0 ? 0.f : -__builtin_inff();
      f32x4 b0[4];
#pragma unroll
      for (int nb = 0; nb < 4; nb++) b0[nb] = *(const f32x4*)(s_bias0 + nb * 16 + fq * 4);
#pragma unroll
      for (int u = 0; u < 5; u++) {
        int r, c;
        patch_rc(u, r, c);
        const int y = ty0 - 1 + r, x = tx0 - 1 + c;
        // channel nb*16 + fq*4 + e of patch pixel pj: plane nb >> 1, 16-byte slot (nb & 1)*2 + (fq >> 1), swizzled by
        // bit 2 of pj; the nb-dependent bit is applied at the store
        const int pj = r * ST_PW + c;
        const int wofs = pj * 64 + (fq & 1) * 8 + (((fq >> 1) << 4) ^ (((pj >> 2) & 1) << 5));
        // outside the image the patch holds conv1_2's zero padding
        const bool inside = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
        f32x4 a0[4];
#pragma unroll
        for (int nb = 0; nb < 4; nb++) a0[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x4 zero4;
#pragma unroll
        for (int e = 0; e < 4; e++) zero4[e] = (bf16)0.f;
        const char* const fsrc = fpl + (r * ST_FW + c) * 8;
        bf16x8 xf[3];
#pragma unroll
        for (int ks = 0; ks < 3; ks++)
          xf[ks] = __builtin_shufflevector(*(const bf16x4*)(fsrc + ftap[ks]), zero4, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int ks = 0; ks < 3; ks++)   // four independent accumulation chains
#pragma unroll
          for (int nb = 0; nb < 4; nb++) a0[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], xf[ks], a0[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 4; nb++) {
          // bias as 2-vectors (packed adds), rounding two values per instruction, and the outside-the-image zero
          // applied to the packed words (bf16 +0 is the zero word): 10 vector operations per 4 values where the
          // element-wise form compiled to 18 -- conv1_1 is issue-bound on its SIMD (the two waves that share one
          // take turns: the second reaches barrier A 3.6 thousand cycles after the first)
          typedef float st_f2 __attribute__((ext_vector_type(2)));
          typedef __bf16 st_h2 __attribute__((ext_vector_type(2)));
          unsigned w[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const st_f2 av = {a0[nb][2 * h], a0[nb][2 * h + 1]}, bv2 = {b0[nb][2 * h], b0[nb][2 * h + 1]};
            const st_f2 sv2 = av + bv2;
            const st_h2 o = {(bf16)fmaxf(sv2.x, relu0_floor), (bf16)fmaxf(sv2.y, relu0_floor)};
            w[h] = inside ? __builtin_bit_cast(unsigned, o) : 0u;
          }
          *(uint2*)(lds + (nb >> 1) * ST_PLANE + (wofs ^ ((nb & 1) << 5))) = make_uint2(w[0], w[1]);
        }
#ifdef ST_STAMP
        if (u == 0) su_[0] = __builtin_readcyclecounter();
        if (u == 4) su_[1] = __builtin_readcyclecounter();
#endif
      }
    }
    // ---- the next tile: its number came in during the previous tile, its frame pixels travel during this K loop and
    // are written to LDS after it.  The tile after that is asked for FIRST, while this wave has no other memory
    // operation in flight: the compiler waits for everything outstanding where the counter's answer is consumed
    const int tn = __builtin_amdgcn_readfirstlane(s_tn[it & 1]);
    int tnn = ntiles;
    if (tid == 0 && tn < ntiles) tnn = (int)gridDim.x + atomicAdd(ctr, 1);
#ifdef ST_STAMP
    su_[2] = __builtin_readcyclecounter();
#endif
    flush_rows();  // the previous tile's pooled rows (none before the first tile: pend_row is out of range)
#ifdef ST_STAMP
    su_[3] = __builtin_readcyclecounter();
#endif
    issue_loads(tn < ntiles ? tn : ntiles - 1);  // (past the last tile: a harmless repeat, no branch around the loads)
    ST_T(c1_);
    __syncthreads();  // A: the patch is complete
    ST_T(c2_);

    // ---- P2: conv1_2, 18 K steps of 16 MFMAs, everything in LDS ------------------------------------------------
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments of step s + 1 are requested before the MFMAs of step s are issued (two fragment sets); the
    // sched_barriers keep the compiler from hoisting more reads than that (it then spills the next tile's pixels)
    bf16x8 bfr[2][4], af[2][4];
    auto read_frags = [&](int s, int set) {
      const int c = s / 9, tp = s - c * 9;
      const int kh = tp / 3, kw = tp - kh * 3;
#pragma unroll
      for (int j = 0; j < 4; j++) bfr[set][j] = *(const bf16x8*)(wl + (tp * 2 + c) * 4096 + j * 1024 + wrd);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int rowi = (i >> 1) + kh;
        af[set][i] = *(const bf16x8*)(pbase + c * ST_PLANE + (rowi * ST_PW + (i & 1) * 16) * 64 + (o0[kw] ^ ((rowi & 1) << 5)));
      }
    };
    read_frags(0, 0);
#pragma unroll
    for (int s = 0; s < 18; s++) {
      if (s + 1 < 18) read_frags(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_frame();  // the next tile's frame pixels (every wave left P1, the last reader of `fpl`, before barrier A)
    if (tid == 0) s_tn[(it + 1) & 1] = tnn;
    ST_T(c3_);
    __syncthreads();  // B: every wave has read the patch for the last time
    ST_T(c4_);

    // ---- P3: bias + activation, 2x2 max pool inside the wave, rows out through this wave's own LDS strip ---------
    // acc[i][j][e]: image row ty0 + 2 wave + (i >> 1), column tx0 + (i & 1) * 16 + fr, channel j * 16 + fq * 4 + e
    f32x4 bv[4], sv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      bv[j] = *(const f32x4*)(s_bias + j * 16 + fq * 4);
      sv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.act == 2) sv[j] = *(const f32x4*)(s_slope + j * 16 + fq * 4);
    }
    const float relu_floor = p.act ? 0.f : -__builtin_inff();
    auto pool_rows = [&](auto monotone) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float m[2];
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {  // jj = 0: channel block k (kept by even columns), 1: block 2 + k (odd columns)
              const int j = jj * 2 + k;
              float mm;
              if (decltype(monotone)::value) {
                // x -> act(x + bias) -> bf16 is monotone for "none" and ReLU: the pooled value is that of the
                // largest of the four accumulators, bit for bit
                mm = fmaxf(acc[h][j][e], acc[h + 2][j][e]);
                mm = fmaxf(mm, st_xor1(mm)) + bv[j][e];
                mm = fmaxf(mm, relu_floor);
              } else {  // PReLU slopes may be negative: activate, round, then pool, as the layers are defined
                float v0 = acc[h][j][e] + bv[j][e], v1 = acc[h + 2][j][e] + bv[j][e];
                v0 = v0 > 0.f ? v0 : v0 * sv[j][e];
                v1 = v1 > 0.f ? v1 : v1 * sv[j][e];
                mm = fmaxf((float)(bf16)v0, (float)(bf16)v1);
                mm = fmaxf(mm, st_xor1(mm));
              }
              m[jj] = mm;
            }
            o[e] = (bf16)((fr & 1) ? m[1] : m[0]);
          }
          const int px = h * 8 + (fr >> 1);
          *(bf16x4*)(stg + px * ST_SROW + ((((fr & 1) * 2 + k) * 16 + fq * 4) << 1)) = o;
        }
      }
    };
    if (p.act != 2) pool_rows(std::true_type{});
    else pool_rows(std::false_type{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private strip: LDS operations of one wave stay in order)
    // The rows are not stored yet: a global store issued here would still be in flight when P1 of the next tile
    // waits for its loads (the memory counter is in order, and the compiler waits for zero outstanding operations at
    // the loop head) -- measured: 7 of 12 thousand cycles of P1.  They wait in registers and go out after the next
    // tile's conv1_1, right before that tile's K loop.
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int v = lane + 64 * q;
      pend[q] = *(const u32x4*)(stg + (v >> 3) * ST_SROW + (v & 7) * 16);
    }
    pend_b = tb;
    pend_row = (ty0 >> 1) + wave;
    pend_col0 = tx0 >> 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is read before P1 of the next tile rewrites it
#ifdef ST_STAMP
    {  // summed in registers, written once after the last tile: an atomic per phase and tile sat in the in-order memory
       // counter in front of the next tile's frame pixels
      const long long c5_ = __builtin_readcyclecounter();
      sacc_[0] += c1_ - c0_;        // P1 (frame pixels from LDS) + previous rows stored + next loads issued
      sacc_[1] += c2_ - c1_;        // barrier A
      sacc_[2] += c3_ - c2_;        // P2: K loop (+ the next tile's frame pixels written to LDS)
      sacc_[3] += c4_ - c3_;        // barrier B
      sacc_[4] += c5_ - c4_;        // P3
      sacc_[5] += 1;
      sacc_[6] += su_[0] - c0_;     // P1: up to the end of row block 0
      sacc_[7] += su_[1] - su_[0];  // P1: row blocks 1..4
      sacc_[8] += su_[2] - su_[1];  // next tile number from LDS, the one after from the counter
      sacc_[9] += su_[3] - su_[2];  // previous tile's rows stored
      sacc_[10] += c1_ - su_[3];    // address arithmetic, next tile's loads issued
    }
#endif
    t = tn;
    it++;
  }
  flush_rows();
#ifdef ST_STAMP
  if (lane == 0 && wave == 1)
    for (int q = 0; q < 11; q++) atomicAdd(&st_dbg[q], (unsigned long long)sacc_[q]);
#endif
}

// POSEHIP_STEM_KERNEL=0: the halo kernel's stem form everywhere (A/B runs, and the bit-equality test of the two)
static bool stem512_enabled() {  // (read at every build / launch: the test switches it inside one process)
  const char* e = getenv("POSEHIP_STEM_KERNEL");
  return !(e && e[0] == '0');
}

// The persistent stem kernel serves the shape it is built for; everything else stays on the halo kernel's stem form.
bool ph_stem512_applies(const PhConvParams& p) {
  return stem512_enabled() && p.stem && p.pool && p.cout == 64 && p.cin_read == 64 && p.Kpad == 576 && p.Kpad0 >= 96 &&
         p.cin0_span <= 4 &&   // (the kernel stages channels 0..3 of the frame pixels)
         p.n_dst >= 1 && p.tile_ctr && !(p.H & 1) && !(p.W & 1) &&
         (unsigned long long)p.B * p.dst[0].Hp * p.dst[0].Wp * p.dst[0].Cs < 0x3fffffffull &&
         (unsigned long long)p.NQ * 16ull < 0xffffffffull;
}

int ph_launch_stem512(const PhConvParams& p, hipStream_t stream) {
  static int attr[PH_MAX_DEVICES];
  static int n_cu[PH_MAX_DEVICES];
  if (ph_ensure_lds((const void*)conv_stem512_kernel, ST_LDS, attr)) return -1;
  int dev = 0;
  PH_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= PH_MAX_DEVICES) dev = 0;
  if (!n_cu[dev]) {
    hipDeviceProp_t prop;
    PH_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int tiles_x = (p.W + ST_TW - 1) / ST_TW, tiles_y = (p.H + ST_TH - 1) / ST_TH;
  const long long ntiles = (long long)p.B * tiles_x * tiles_y;
  if (ntiles < 1 || ntiles > 0x3fffffff) {
    ph_set_error("ph_launch_stem512: %lld tiles", ntiles);
    return -1;
  }
  const int grid = ntiles < n_cu[dev] ? (int)ntiles : n_cu[dev];
  hipLaunchKernelGGL(conv_stem512_kernel, dim3(grid), dim3(512), ST_LDS, stream, p, (int)ntiles, tiles_x, tiles_y);
  PH_CHECK(hipGetLastError());
  return 0;
}
