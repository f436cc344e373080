// The two 1x1 convolutions that close every refinement stage of the BODY_25 / COCO networks (Mconv6: Cin -> 256 / 512
// with PReLU, Mconv7: -> 52 / 26 / 38 / 19 output maps; part of the forward the reference delegates to cv2.dnn /
// Caffe, openpose/opencv/skeleton.py:74-75, openpose/native/python/skeleton.py:447) as ONE kernel for gfx950.
//
// Why.  As two launches the wide intermediate map (512 channels x 120 704 pixels at batch 32 = 123 MB in bf16) is
// written by the first and read back by the second; both are far from the MFMA roof (30 % / 16 % busy): a 1x1 step
// moves a whole pixel chunk + weight slice through the CU's load path per 16 MFMAs of a wave, and the second layer is
// 25 us of fixed cost around four K steps.  Fusing them removes the 123 MB round trip and one launch, not that
// bound: measured 89 us against 76 + 25 us, each alone on the chip (profiles/r03_pmc_conv_counters.md; DESIGN.md
// section 9 has four further forms of this kernel that were built and did not beat it).  A block owns 256 pixels
// and walks the wide layer's output in SLABS of 128 channels:
//   for every slab:  K loop over the input chunks (pixel chunk + weight slice per step through 3-deep LDS rings, the
//                    1x1 form of the halo kernel's pipeline) -> bias + activation -> the slab as bf16 in an LDS tile
//                    -> the narrow layer's partial product with its 128 x 64 weight slab (4 K steps, 8 MFMAs each)
//                    accumulated in registers;
//   at the end:      the narrow layer's bias / activation, fp32 NCHW network output and / or bf16 NHWC windows.
// The wide map never exists in memory, one launch replaces two, and the loads of the next slab's first K steps are in
// flight while the partial product runs (the rings are separate from the slab tile).  One block of 8 waves per CU
// (LDS: rings 72 KB + slab tile 68 KB + weight slab 16 KB), 64 + 32 accumulator registers.
// Rounding and accumulation order are those of the two separate kernels (the intermediate is rounded to bf16, the
// narrow layer accumulates its K in ascending chunks of 32), so the fused result is bit-identical to theirs.
#include "common.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define PP_BM 256
#define PP_XBYTES (16 * 1024)            // one pixel chunk: 256 pixels x 32 channels
#define PP_WBYTES (8 * 1024)             // one weight slice: 128 channels x 32 k
#define PP_W2BYTES (16 * 1024)           // the narrow layer's slab: 4 slices of 64 channels x 32 k
#define PP_YROW 272                      // slab tile row: 128 channels bf16 + 16 B
#define PP_MAXC 512                      // widest intermediate layer (its bias / slopes stay in LDS)
#define PP_LDS (3 * PP_XBYTES + 3 * PP_WBYTES + PP_W2BYTES + PP_BM * PP_YROW + 2 * PP_MAXC * 4)   // = 160 KB

#define PP_WAIT_VMCNT(n)                                              \
  do {                                                                \
    switch (n) {                                                      \
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; \
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break; \
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break; \
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break; \
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break; \
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break; \
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
    }                                                                 \
  } while (0)

__global__ __launch_bounds__(512, 1) void conv1x1_pair_kernel(const PhPairParams q) {
  const PhConvParams& p = q.a;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const xring = lds;
  char* const wring = xring + 3 * PP_XBYTES;
  char* const w2buf = wring + 3 * PP_WBYTES;
  char* const ytile = w2buf + PP_W2BYTES;
  float* const s_b1 = (float*)(ytile + PP_BM * PP_YROW);   // wide layer: bias, PReLU slopes [PP_MAXC each]
  float* const s_s1 = s_b1 + PP_MAXC;
  int2* const s_row = (int2*)xring;   // epilogue only (the rings are dead by then): per pixel row {image or -1, y << 16 | x}

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int fr = lane & 15, fq = lane >> 4;
  const int Wp = p.in_Wp, Hp = p.in_Hp, pi = p.in_pad;
  const int m0 = blockIdx.x * PP_BM;       // first (unpadded) pixel index of the tile
  const int nch = p.cin_read >> 5;         // K steps per slab
  const int n_slabs = p.cout >> 7;
  const int total = n_slabs * nch;

  // ---- staging addresses ------------------------------------------------------------------------------------------
  const int lchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);   // source 16-byte unit of this lane's LDS slot (swizzled)
  unsigned x_src[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    int m = m0 + (wave + 8 * u) * 16 + (lane >> 2);
    m = m < p.M ? m : p.M - 1;
    const int hw = p.H * p.W;
    const int b = m / hw;
    const int rem = m - b * hw;
    const int y = rem / p.W;
    const int Q = (b * Hp + y + pi) * Wp + (rem - y * p.W) + pi;
    x_src[u] = ((unsigned)Q * (unsigned)p.in_Cs + (unsigned)(p.in_coff + lchunk * 8)) * 2u;   // byte offset
  }
  const unsigned w_src = ((unsigned)(wave * 16 + (lane >> 2)) * (unsigned)p.Kpad + (unsigned)(lchunk * 8)) * 2u;
  unsigned w2_src[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int piece = wave + 8 * u;          // 16 pieces: slice piece >> 2, rows (piece & 3) * 16 ..
    w2_src[u] = ((unsigned)((piece & 3) * 16 + (lane >> 2)) * (unsigned)q.Kpad2 + (unsigned)((piece >> 2) * 32 + lchunk * 8)) * 2u;
  }
  auto load_step = [&](int g) -> int {   // operands of global step g (slab g / nch, chunk g % nch) -> ring slot g % 3
    const int n = g / nch, c = g - n * nch;
    const int slot = g % 3;
#pragma unroll
    for (int u = 0; u < 2; u++)
      __builtin_amdgcn_global_load_lds((glb_void*)((const char*)p.in + (x_src[u] + (unsigned)(c * 64))),
                                       (lds_void*)(xring + slot * PP_XBYTES + (wave + 8 * u) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(
        (glb_void*)((const char*)p.wt + (w_src + (unsigned)(n * 128) * (unsigned)p.Kpad * 2u + (unsigned)(c * 64))),
        (lds_void*)(wring + slot * PP_WBYTES + wave * 1024), 16, 0, 0);
    return 3;
  };
  auto load_w2 = [&](int n) -> int {     // the narrow layer's weights for slab n (k = n * 128 .. + 127)
#pragma unroll
    for (int u = 0; u < 2; u++)
      __builtin_amdgcn_global_load_lds((glb_void*)((const char*)q.wt2 + (w2_src[u] + (unsigned)(n * 256))),
                                       (lds_void*)(w2buf + (wave + 8 * u) * 1024), 16, 0, 0);
    return 2;
  };

  // ---- fragment read offsets ----------------------------------------------------------------------------------------
  const int swz = (fq * 16) ^ ((fr & 4) << 3);
  const int x_rd = (wm * 64 + fr) * 64 + swz;          // + i * 1024
  const int w_rd = (wn * 64 + fr) * 64 + swz;          // + j * 1024
  const int w2_rd = (wn * 32 + fr) * 64 + swz;         // + k2 * 4096 + j * 1024
  const int y_rd = (wm * 64 + fr) * PP_YROW + fq * 16; // + i * 16 * PP_YROW + k2 * 64
  const int y_wr = (wm * 64 + fr) * PP_YROW + (wn * 64 + fq * 4) * 2;   // + i * 16 * PP_YROW + j * 32

  for (int k = tid; k < p.cout; k += 512) {   // (visible to every wave after the first K step's barrier)
    s_b1[k] = p.bias[k];
    s_s1[k] = p.slope[k];
  }
  f32x4 z[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) z[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: operands of steps 0 and 1 ----------------------------------------------------------------------------
  int prev_cnt = 0;
  load_step(0);
  if (total > 1) prev_cnt = load_step(1);
  int g = 0;
  for (int n = 0; n < n_slabs; n++) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; c++, g++) {
      PP_WAIT_VMCNT(prev_cnt);   // everything issued before the previous step has landed: step g's operands are in LDS
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      prev_cnt = 0;
      if (g + 2 < total) prev_cnt += load_step(g + 2);
      if (c == 0) prev_cnt += load_w2(n);   // (every wave has left the previous slab's partial product: barrier above)
      const char* xb = xring + (g % 3) * PP_XBYTES + x_rd;
      const char* wb = wring + (g % 3) * PP_WBYTES + w_rd;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int j = 0; j < 4; j++) bfr[j] = *(const bf16x8*)(wb + j * 1024);
#pragma unroll
      for (int i = 0; i < 4; i++) af[i] = *(const bf16x8*)(xb + i * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the slab: bias + activation -> bf16 tile in LDS (what the separate kernel would have stored) -----------------
    // (the tile's previous readers have passed this slab's first step barrier)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x4 bv = *(const f32x4*)(s_b1 + n * 128 + wn * 64 + j * 16 + fq * 4);
      f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.act == 2) sv = *(const f32x4*)(s_s1 + n * 128 + wn * 64 + j * 16 + fq * 4);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float v = acc[i][j][e] + bv[e];
          if (p.act) v = v > 0.f ? v : v * sv[e];
          o[e] = (bf16)v;
        }
        *(bf16x4*)(ytile + y_wr + i * 16 * PP_YROW + j * 32) = o;
      }
    }
    // the weight slab was requested in this slab's first step: the third step's counted wait and barrier cover it
    // (the host only sends layers of >= 3 chunks here; shorter ones drain the queue instead)
    if (nch < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // tile complete.  A plain s_barrier, not __syncthreads: the next slab's first operands stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- partial product of the narrow layer: z += W2[:, slab] x tile ---------------------------------------------------
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
      bf16x8 wf[2], yf[4];
#pragma unroll
      for (int j = 0; j < 2; j++) wf[j] = *(const bf16x8*)(w2buf + k2 * 4096 + w2_rd + j * 1024);
#pragma unroll
      for (int i = 0; i < 4; i++) yf[i] = *(const bf16x8*)(ytile + y_rd + i * 16 * PP_YROW + k2 * 64);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) z[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], yf[i], z[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue of the narrow layer ----------------------------------------------------------------------------------------
  // z[i][j][e]: pixel row wm*64 + i*16 + fr, channel wn*32 + j*16 + fq*4 + e
  const int cbase = wn * 32 + fq * 4;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const f32x4 bv = *(const f32x4*)(q.bias2 + cbase + j * 16);   // (both arrays are padded to 128 entries and more)
    f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q.act2 == 2) sv = *(const f32x4*)(q.slope2 + cbase + j * 16);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float v = z[i][j][e] + bv[e];
        if (q.act2) v = v > 0.f ? v : v * sv[e];
        z[i][j][e] = v;
      }
  }
  __syncthreads();   // every wave has left the last partial product: rings and tile are free
  if (tid < PP_BM) {
    const int m = m0 + tid;
    const int hw = p.H * p.W;
    const int mc = m < p.M ? m : p.M - 1;
    const int b = mc / hw;
    const int rem = mc - b * hw;
    const int y = rem / p.W;
    s_row[tid] = make_int2(m < p.M ? b : -1, (y << 16) | (rem - y * p.W));
  }
  __syncthreads();   // row table
  if (q.f32_out2) {  // network output: fp32 NCHW straight from the accumulators (16 consecutive pixels per store)
    const int hw = p.H * p.W;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int2 ri = s_row[wm * 64 + i * 16 + fr];
      if (ri.x < 0) continue;
      float* o = q.f32_out2 + ((size_t)ri.x * q.f32_C2 + q.f32_coff2 + cbase) * hw + (ri.y >> 16) * p.W + (ri.y & 0xffff);
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (cbase + j * 16 + e < q.cout2) o[(size_t)(j * 16 + e) * hw] = z[i][j][e];
    }
  }
  if (q.n_dst2 > 0) {
    constexpr int ZROW = 64 * 2 + 16;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (bf16)z[i][j][e];
        *(bf16x4*)(ytile + (wm * 64 + i * 16 + fr) * ZROW + (cbase + j * 16) * 2) = o;
      }
    __syncthreads();
    for (int d = 0; d < q.n_dst2; d++) {
      const PhDst ds = q.dst2[d];
      bf16* const dbase = ds.base + (size_t)ds.org;
      for (int v = tid; v < PP_BM * 8; v += 512) {
        const int row = v >> 3, cv = (v & 7) * 8;
        const int2 ri = s_row[row];
        if (ri.x < 0 || cv >= q.cout2) continue;
        const char* sp = ytile + row * ZROW + cv * 2;
        bf16* o = dbase + ((size_t)(ri.x * ds.Hp + (ri.y >> 16)) * ds.Wp + (ri.y & 0xffff)) * ds.Cs + cv;
        if (cv + 8 <= q.cout2) {
          *(u32x4*)o = *(const u32x4*)sp;
        } else {
          for (int e = 0; e < q.cout2 - cv; e++) o[e] = ((const bf16*)sp)[e];
        }
      }
    }
  }
}

// POSEHIP_PAIR_KERNEL=0: the two 1x1 layers as separate launches (A/B runs and the bit-equality test)
bool ph_pair_enabled() {
  const char* e = getenv("POSEHIP_PAIR_KERNEL");
  return !(e && e[0] == '0');
}

int ph_launch_pair(const PhPairParams& q, hipStream_t stream) {
  static int attr[PH_MAX_DEVICES];
  if (ph_ensure_lds((const void*)conv1x1_pair_kernel, PP_LDS, attr)) return -1;
  const int n_mtiles = (q.a.M + PP_BM - 1) / PP_BM;
  hipLaunchKernelGGL(conv1x1_pair_kernel, dim3(n_mtiles), dim3(512), PP_LDS, stream, q);
  PH_CHECK(hipGetLastError());
  return 0;
}
