// Implicit-GEMM convolution for gfx950 (CDNA4): bf16 NHWC activations held in
// zero-bordered buffers, bf16 weights [Cout][KH*KW*Cin], fp32 MFMA accumulation.
//
// Replaces the network forward the reference delegates to cv2.dnn / Caffe
// (openpose/opencv/skeleton.py:74-75 `net.forward()`;
//  openpose/native/python/skeleton.py:447 `emplaceAndPop`).
//
// GEMM view: M = B*H*W output pixels, N = Cout, K = KH*KW*cin_read.  The A
// operand is gathered on the fly: every lane of a `global_load_lds_dwordx4`
// computes its own source address (pixel row base + tap offset + channel), so
// im2col never exists in memory and the zero border of the activation buffers
// supplies the conv padding without any masking.  The XOR swizzle that makes
// the ds_read_b128 fragment reads conflict-free is applied on the per-lane
// SOURCE address (LDS-DMA destinations are lane-linear).
//
// Block = 256 threads = 4 waves (2 along M x 2 along N); tile 128 x BN x 64,
// double-buffered LDS, v_mfma_f32_16x16x32_bf16.  Epilogue: bias + ReLU/PReLU,
// optional fp32 NCHW store straight from the accumulators (network outputs),
// bf16 tile through LDS -> 16-byte coalesced NHWC stores into up to PH_MAX_DST
// channel windows (concat-by-offset, so Caffe Concat layers cost nothing).
#include "common.h"
#define PH_AUX_SC0_SC1 17   // buffer-store cache policy on gfx94x/gfx950: bit 0 = sc0, bit 4 = sc1 (system scope)

#define BM 128
#define BK 64
#define NTHREADS 256

// Timing experiments only (PhConvParams::debug & 8, the DBG instantiations): per-phase cycle sums over all blocks.
// (The in-kernel stamp / timeline builds of rounds 1-2, -DPH_LIN_STAMP and -DPH_TIMELINE, were taken out of this file
// in round 3; their results are in DESIGN 4 / 8 and profiles/r01_conv_timeline.txt, their code in the history.)
__device__ unsigned long long ph_conv_dbg[8];
extern "C" int ph_debug_conv_cycles(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[8] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ph_conv_dbg), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ph_conv_dbg), 8 * sizeof(unsigned long long));
}

template <int NREP>  // BN = 32 * NREP
__global__ __launch_bounds__(NTHREADS) void conv_igemm_kernel(const PhConvParams p) {
  constexpr int BN = 32 * NREP;
  constexpr int A_BYTES = BM * BK * 2;       // 16 KB
  constexpr int B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CROW = BN + 8;               // bf16 elements per epilogue row
  constexpr int EPI_BYTES = BM * CROW * 2;
  constexpr int LDS_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const bool dbg_t = (p.debug & 8) != 0;
  long long t0 = 0, t1 = 0, t2 = 0, tw = 0;
  if (dbg_t) t0 = clock64();

  // XCD-aware bijective remap: blocks that share an XCD (id % 8) get a
  // contiguous run of logical tiles so the N tiles of one M tile and
  // neighbouring M tiles (shared halo rows) hit the same L2.
  const unsigned nblk = gridDim.x;
  const unsigned bid = blockIdx.x;
  const unsigned xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int mt = logical / p.n_ntiles;
  const int nt = logical - mt * p.n_ntiles;
  const int m0 = mt * BM;
  const int n0 = nt * BN;

  // ---- per-thread staging addresses ---------------------------------------
  const int srow = tid >> 3;                 // 0..31: row inside a 32-row group
  const int chunk = (tid & 7) ^ (srow & 7);  // source 16-B chunk (swizzled)
  const int HW = p.H * p.W;
  unsigned a_off[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int m = m0 + i * 32 + srow;
    m = m < p.M ? m : p.M - 1;
    const int b = m / HW;
    const int rem = m - b * HW;
    const int y = rem / p.W;
    const int x = rem - y * p.W;
    a_off[i] = (unsigned)p.in_org + (unsigned)((b * p.in_Hp + y) * p.in_Wp + x) * (unsigned)p.in_Cs;
  }
  unsigned w_off[NREP];
#pragma unroll
  for (int i = 0; i < NREP; i++) w_off[i] = (unsigned)(n0 + i * 32 + srow) * (unsigned)p.Kpad;

  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  auto stage = [&](int kt, int buf) {
    const int k = kt * BK + chunk * 8;
    unsigned off = 0;
    if (k < p.K) {
      const unsigned tap = ph_div((unsigned)k, p.magic_cin, (unsigned)p.cin_read);
      const unsigned c = (unsigned)k - tap * (unsigned)p.cin_read;
      const unsigned kh = ph_div(tap, p.magic_kw, (unsigned)p.KW);
      const unsigned kw = tap - kh * (unsigned)p.KW;
      off = (kh * (unsigned)p.in_Wp + kw) * (unsigned)p.in_Cs + c;
    }
    char* sa = lds + buf * STAGE + wave * 1024;
    if (!(p.debug & 1) || kt == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        __builtin_amdgcn_global_load_lds((glb_void*)(p.in + a_off[i] + off), (lds_void*)(sa + i * 4096), 16, 0, 0);
    }
    char* sb = lds + buf * STAGE + A_BYTES + wave * 1024;
    if (!(p.debug & 2) || kt == 0) {
#pragma unroll
      for (int i = 0; i < NREP; i++)
        __builtin_amdgcn_global_load_lds((glb_void*)(p.wt + w_off[i] + k), (lds_void*)(sb + i * 4096), 16, 0, 0);
    }
  };

  // ---- fragment read addresses ---------------------------------------------
  // row & 7 == lane & 7 for every fragment row (tile bases are multiples of 8)
  const int fr = lane & 15, fq = lane >> 4, sw = lane & 7;
  const int a_rd0 = (wm * 64 + fr) * 128 + (((0 + fq) ^ sw) << 4);
  const int a_rd1 = (wm * 64 + fr) * 128 + (((4 + fq) ^ sw) << 4);
  const int b_rd0 = A_BYTES + (wn * (BN / 2) + fr) * 128 + (((0 + fq) ^ sw) << 4);
  const int b_rd1 = A_BYTES + (wn * (BN / 2) + fr) * 128 + (((4 + fq) ^ sw) << 4);

  f32x4 acc[4][NREP];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NREP; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* base = lds + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 af[4], bfr[NREP];
      const int ao = ks ? a_rd1 : a_rd0;
      const int bo = ks ? b_rd1 : b_rd0;
#pragma unroll
      for (int i = 0; i < 4; i++) af[i] = *(const bf16x8*)(base + ao + i * 2048);
#pragma unroll
      for (int j = 0; j < NREP; j++) bfr[j] = *(const bf16x8*)(base + bo + j * 2048);
      if (p.debug & 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("" ::"v"(af[i]));
#pragma unroll
        for (int j = 0; j < NREP; j++) asm volatile("" ::"v"(bfr[j]));
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < NREP; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop: one barrier per K tile -----------------------------------
  const int nk = p.nk;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  if (dbg_t) t1 = clock64();
  for (int kt = 0; kt < nk - 1; kt++) {
    stage(kt + 1, buf ^ 1);
    compute(buf);
    long long ta = 0;
    if (dbg_t) ta = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA of tile kt+1 landed
    __syncthreads();
    if (dbg_t) tw += clock64() - ta;
    buf ^= 1;
  }
  compute(buf);
  if (dbg_t) t2 = clock64();

  // ---- epilogue --------------------------------------------------------------
  // C layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + j
  const int ccol = wn * (BN / 2) + fr;
  float bias_v[NREP], slope_v[NREP];
#pragma unroll
  for (int j = 0; j < NREP; j++) {
    bias_v[j] = p.bias[n0 + ccol + j * 16];
    slope_v[j] = p.act == 2 ? p.slope[n0 + ccol + j * 16] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NREP; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float v = acc[i][j][e] + bias_v[j];
        if (p.act) v = v > 0.f ? v : v * slope_v[j];
        acc[i][j][e] = v;
      }

  if (p.f32_out) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int m = m0 + wm * 64 + i * 16 + fq * 4 + e;
        if (m < p.M) {
          const int b = m / HW;
          const int rem = m - b * HW;
#pragma unroll
          for (int j = 0; j < NREP; j++) {
            const int n = n0 + ccol + j * 16;
            if (n < p.cout)
              p.f32_out[((size_t)b * p.f32_C + p.f32_coff + n) * HW + rem] = acc[i][j][e];
          }
        }
      }
  }

  if (p.n_dst > 0) {
    __syncthreads();  // all fragment reads of the last K tile are done
    bf16* ct = (bf16*)lds;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NREP; j++)
#pragma unroll
        for (int e = 0; e < 4; e++)
          ct[(wm * 64 + i * 16 + fq * 4 + e) * CROW + ccol + j * 16] = (bf16)acc[i][j][e];
    __syncthreads();
    // 16-byte row segments: BN/8 vectors per pixel row
    constexpr int VPR = BN / 8;
    const int nvalid = p.cout - n0;            // valid channels in this N tile
    for (int v = tid; v < BM * VPR; v += NTHREADS) {
      const int row = v / VPR;
      const int cv = (v - row * VPR) * 8;
      const int m = m0 + row;
      if (m >= p.M || cv >= nvalid) continue;
      const int b = m / HW;
      const int rem = m - b * HW;
      const int y = rem / p.W;
      const int x = rem - y * p.W;
      const u32x4 val = *(const u32x4*)(ct + row * CROW + cv);
      for (int d = 0; d < p.n_dst; d++) {
        const PhDst& ds = p.dst[d];
        bf16* o = ds.base + (size_t)ds.org + ((size_t)(b * ds.Hp + y) * ds.Wp + x) * ds.Cs + n0 + cv;
        if (cv + 8 <= nvalid) {
          *(u32x4*)o = val;
        } else {
          const bf16* s = ct + row * CROW + cv;
          for (int e = 0; e < nvalid - cv; e++) o[e] = s[e];
        }
      }
    }
  }
  if (dbg_t) {
    __syncthreads();
    if (tid == 0) {
      const long long t3 = clock64();
      atomicAdd(&ph_conv_dbg[0], (unsigned long long)(t1 - t0));
      atomicAdd(&ph_conv_dbg[1], (unsigned long long)(t2 - t1));
      atomicAdd(&ph_conv_dbg[2], (unsigned long long)(t3 - t2));
      atomicAdd(&ph_conv_dbg[3], (unsigned long long)tw);
      atomicAdd(&ph_conv_dbg[4], 1ull);
    }
  }
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// 3x3 halo-patch kernel.  The im2col gather above re-reads every input pixel 9x
// through the L2 -> LDS path, and that path (~35-40 B/clk/CU measured), not the
// MFMA, bounds it.  Here a block owns 256 output pixels x BN channels and, per
// 32-channel chunk of Cin, brings the input PATCH (the pixels any of the 9 taps
// touch) into LDS once; the 9 taps then read shifted windows of the patch, and
// only the 128 x 32 weight slice of each (tap, chunk) is streamed per K step.
//
// Output pixels are enumerated either as a 256-long run of the PADDED linear
// pixel index (mode 0: the patch is the run widened by Wp+1 on both sides; the
// outputs that fall on border pixels are computed and dropped -- 46x82 maps: 7 %),
// or as 16x16 tiles (mode 1: 18x18 patch).  4 waves as 2 (M) x 2 (N), wave tile
// 128 x BN/2 (acc 8 x NREP MFMA tiles), two patch buffers (chunk c+1 arrives in 8
// pieces during the 9 steps of chunk c), three weight-slice slots, counted vmcnt,
// one barrier per K step, 2 blocks per CU.
// LDS rows are 64 B; k-chunk c of patch pixel / weight row r sits in slot
// c ^ ((r >> 1) & 2): conflict-free ds_read_b128 for ANY 16 consecutive rows.
// ---------------------------------------------------------------------------
#define HALO_BM 256

#define PH_WAIT_VMCNT(n)                                            \
  do {                                                              \
    switch (n) {                                                    \
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; \
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break; \
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break; \
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break; \
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break; \
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break; \
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break; \
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
    }                                                               \
  } while (0)

// TAPS = 9: 3x3 conv (halo patch, 9 K steps per 32-channel chunk).  TAPS = 1: 1x1 conv through the same
// pipeline: the "patch" of a chunk is just the tile's 256 pixels, lives for one K step, and rotates through
// three buffers (halo_mode 2: tiles are 256-long runs of the unpadded pixel index).
//
// LIN: branch-free steady state for the padded-linear form (halo_mode 0) at BN = 128.  Measured on conv4_2
// (ablations, tools/conv_bench.py): the K step's non-MFMA part (barrier, LDS-DMA issue under uniform branches,
// ~44 VALU of fragment addressing, 12 fragment reads and a full lgkmcnt(0) wait, a branch tree for the counted
// vmcnt) took as long as its 32 MFMAs and overlapped with them hardly at all.  In LIN every wave issues the
// same number of LDS-DMA instructions in every step (out-of-range pieces are clamped to a duplicate of the last
// valid one, loads past the end of K re-fetch valid data into a free slot), so the vmcnt immediates, the weight
// slot and the fragment offsets are compile-time constants and the loop body has no branches.
template <int NREP, int MW, bool DBG, bool STEM = false, int TAPS = 9>  // MW waves along M x 2 along N
__global__ __launch_bounds__(128 * MW, MW) void conv3x3_halo_kernel(const PhConvParams p) {
  constexpr bool LINC = !STEM && !DBG && TAPS == 9;  // instantiations with the LIN loop
  const bool LIN = LINC && p.lin;
  constexpr int BN = 32 * NREP;
  constexpr int NW = 2 * MW;                   // waves per block
  constexpr int NT = 64 * NW;
  constexpr int IB = 16 / MW;                  // 16-pixel row blocks per wave
  constexpr int NPC = 32 / NW;                 // patch pieces (one LDS-DMA instruction per wave each)
  constexpr int NBW = (8 + NW - 1) / NW;       // weight staging instructions per wave per step (max)
  constexpr bool PIPE = NREP < 4;              // cross-step fragment prefetch (where registers allow it)
  constexpr int NBI = BN / 16;                 // 16-row weight staging instructions per K step
  constexpr int B_STAGE = NBI * 1024;
  constexpr int NBS = 3;
  constexpr int CROWB = BN * 2 + 16;
  constexpr int NPB = TAPS == 9 ? 2 : 3;       // patch buffers
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int patch_bytes = p.NI * 1024;
  char* const bring = lds + NPB * patch_bytes;
  // (STEM keeps 2 slots x 3 taps of weight slices instead of 3 slots x 1 tap)
  float* const s_bias = (float*)(bring + (STEM ? 6 : NBS) * B_STAGE);
  float* const s_slope = s_bias + 128;
  char* const s_in = (char*)(s_slope + 128);   // STEM: 20x20 pixel x 16 B input patch of the producer conv

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % MW, wn = wave / MW;
  const bool dbg_t = DBG && (p.debug & 8) != 0;
  long long t0 = 0, t1 = 0, t2 = 0, tw = 0;
  if (dbg_t) t0 = clock64();

  const unsigned nblk = gridDim.x;
  const unsigned bid = blockIdx.x;
  const unsigned xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int mt = logical / p.n_ntiles;
  const int nt = logical - mt * p.n_ntiles;
  const int n0 = nt * BN;
  const int Wp = p.in_Wp, Hp = p.in_Hp, pi = p.in_pad;

  // tile geometry: centre index of output row 0 / of the patch origin
  int Qc0;               // centre index of output row 0
  int tb = 0, ty0 = 0, tx0 = 0;  // mode 1: image and origin of the 16x16 tile
  if (TAPS == 1) {
    Qc0 = mt * HALO_BM;  // halo_mode 2: first unpadded pixel index m of the tile
  } else if (p.halo_mode == 0) {
    Qc0 = p.q_first + mt * HALO_BM;
  } else {
    const int tx = mt % p.tiles_x;
    const int t2 = mt / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    tb = t2 / p.tiles_y;
    ty0 = ty * 16;
    tx0 = tx * 16;
    Qc0 = (tb * Hp + ty0 + pi) * Wp + tx0 + pi;
  }
  const int Qbase = Qc0 - Wp - 1;

  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  if (wave < 4) {  // bias / slope -> LDS
    const float* src = (wave < 2 ? p.bias : p.slope) + n0 + (wave & 1) * 64 + lane;
    float* dst = (wave < 2 ? s_bias : s_slope) + (wave & 1) * 64;
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 4, 0, 0);
  }

  // ---- staging addresses -----------------------------------------------------
  const int lchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);   // source k-chunk of this lane's slot
  unsigned a_src[NPC];
#pragma unroll
  for (int u = 0; u < NPC; u++) {
    int idx = u * NW + wave;
    idx = idx < p.NI ? idx : p.NI - 1;
    const int j = idx * 16 + (lane >> 2);
    int Q;
    if (TAPS == 1) {
      int m = Qc0 + j;
      m = m < p.M ? m : p.M - 1;
      const int hw = p.H * p.W;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int y = rem / p.W;
      Q = (b * Hp + y + pi) * Wp + (rem - y * p.W) + pi;
    } else {
      int jr = (int)ph_div((unsigned)j, p.magic_pr, (unsigned)p.PR);
      jr = jr < p.PCLIP ? jr : p.PCLIP;
      Q = Qbase + j + jr * (Wp - p.PR);
    }
    Q = Q < 0 ? 0 : (Q < p.NQ ? Q : p.NQ - 1);
    a_src[u] = ((unsigned)Q * (unsigned)p.in_Cs + (unsigned)(p.in_coff + lchunk * 8)) * 2u;  // BYTE offset
  }
  unsigned w_src[NBW];
#pragma unroll
  for (int i = 0; i < NBW; i++) {
    int piece = wave + i * NW;
    if (LIN) piece = piece < NBI ? piece : NBI - 1;  // duplicate of the last piece instead of no instruction
    int n = n0 + piece * 16 + (lane >> 2);
    n = n < p.cout_pad ? n : p.cout_pad - 1;
    w_src[i] = ((unsigned)n * (unsigned)p.Kpad + (unsigned)(lchunk * 8)) * 2u;  // BYTE offset
  }
  int nb_w = 0;  // weight instructions of this wave per step
#pragma unroll
  for (int i = 0; i < NBW; i++) nb_w += wave + i * NW < NBI ? 1 : 0;

  const int nchunks = p.cin_read >> 5;
  const int ns = nchunks * TAPS;

  auto load_b = [&](int t, int cc, int slot, bool lin = false) {  // weight slice of (tap t, chunk cc) -> slice slot `slot`
    const unsigned k0 = (unsigned)(t * p.cin_read + cc * 32) * 2u;
    const char* wbase = (const char*)p.wt;  // uniform base + 32-bit lane offset: saddr addressing, no 64-bit VGPR pairs
    char* sb = bring + slot * B_STAGE;
#pragma unroll
    for (int i = 0; i < NBW; i++) {
      int piece = wave + i * NW;
      if (LIN) piece = piece < NBI ? piece : NBI - 1;
      if (lin || piece < NBI)
        __builtin_amdgcn_global_load_lds((glb_void*)(wbase + (w_src[i] + k0)), (lds_void*)(sb + piece * 1024), 16, 0, 0);
    }
  };
  auto load_a = [&](int u, int cc, int buf, bool lin = false) -> int {  // piece u of the patch of chunk cc
    int piece = u * NW + wave;
    if (lin) {
      piece = piece < p.NI ? piece : p.NI - 1;  // duplicate of the last piece (a_src is clamped the same way)
    } else if (piece >= p.NI) {
      if (!LIN) return 0;
      piece = p.NI - 1;
    }
    char* sa = lds + buf * patch_bytes + piece * 1024;
    __builtin_amdgcn_global_load_lds((glb_void*)((const char*)p.in + (a_src[u] + (unsigned)(cc * 64))), (lds_void*)sa, 16, 0,
                                     0);
    return 1;
  };

  // ---- fragment read addresses ---------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int SI = TAPS == 1 ? 16 : p.SI;
  const int pjl = wm * IB * SI + fr;            // + kh*SKH + kw + i*SI
  const int b_rd = (wn * (BN / 2) + fr) * 64 + ((fq * 16) ^ ((fr & 4) << 3));

  f32x4 acc[IB][NREP];
#pragma unroll
  for (int i = 0; i < IB; i++)
#pragma unroll
    for (int j = 0; j < NREP; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: whole patch of chunk 0, weight slices of steps 0 and 1 -------
  if constexpr (!STEM) {
#pragma unroll
    for (int u = 0; u < NPC; u++) load_a(u, 0, 0);
  }
  if constexpr (STEM) {
#pragma unroll
    for (int kw = 0; kw < 3; kw++) load_b(kw, 0, kw);  // kernel row 0 of chunk 0 -> slices 0..2
  } else {
    load_b(0, 0, 0);
  }
  int prev_cnt = 0;  // LDS-DMA instructions this wave issued in the previous step
  if (!STEM && ns > 1) {
    load_b(TAPS == 1 ? 0 : 1, TAPS == 1 ? 1 : 0, 1);
    prev_cnt = nb_w;
    if constexpr (TAPS == 1) {
#pragma unroll
      for (int u = 0; u < NPC; u++) prev_cnt += load_a(u, 1, 1);
    }
  }
  if constexpr (STEM) {
    // ---- stem: this conv's 18x18 x 64-channel input patch = ReLU(conv3x3(8-channel frame patch)) ----
    // 20x20 frame pixels (16 B each) -> s_in; border pixels of the 18x18 patch that fall outside the
    // image are this conv's zero padding, so whatever the clamped loads bring for them is masked below
    static_assert(!STEM || (NREP == 2 && MW == 4), "stem fusion is built for 64 -> 64 channels, 8 waves");
    if (wave < 7) {
      int jj = wave * 64 + lane;
      jj = jj < 400 ? jj : 399;
      const int rr = jj / 20, c2 = jj - rr * 20;
      int Q = (tb * Hp + ty0 - 2 + rr + pi) * Wp + tx0 - 2 + c2 + pi;
      Q = Q < 0 ? 0 : (Q < p.NQ ? Q : p.NQ - 1);
      __builtin_amdgcn_global_load_lds((glb_void*)((const char*)p.in + (unsigned)Q * 16u), (lds_void*)(s_in + wave * 1024),
                                       16, 0, 0);
    }
    const int lf = lane & 15, lq = lane >> 4;
    // each wave owns ONE 16-channel block of the producer conv (nb = wave & 3) and every other row block
    // (wave >> 2): 3 weight fragments per lane instead of 12 -- a quarter of the weight bytes every block
    // pulls from L2 before it can start
    const int nb = wave & 3;
    bf16x8 wf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ks++) wf[ks] = *(const bf16x8*)(p.wt0 + (nb * 16 + lf) * p.Kpad0 + ks * 32 + lq * 8);
    const f32x4 b0 = *(const f32x4*)(p.bias0 + nb * 16 + lq * 4);
    int toff[3];
#pragma unroll
    for (int ks = 0; ks < 3; ks++) {
      int t = ks * 4 + lq;
      t = t < 8 ? t : 8;  // taps 9..11 carry zero weights: read any finite pixel
      toff[ks] = ((t / 3) * 20 + (t % 3)) * 16;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    prev_cnt = 0;  // everything issued so far has landed
    for (int rb = wave >> 2; rb < 21; rb += NW / 4) {
      const int j = rb * 16 + lf;
      const int jc = j < 324 ? j : 323;
      const int r = jc / 18, c = jc - r * 18;
      const char* src = s_in + (r * 20 + c) * 16;
      bf16x8 xf[3];
#pragma unroll
      for (int ks = 0; ks < 3; ks++) xf[ks] = *(const bf16x8*)(src + toff[ks]);
      const int y = ty0 - 1 + r, x = tx0 - 1 + c;
      const bool inside = j < 324 && tb < p.B && y >= 0 && y < p.H && x >= 0 && x < p.W;
      f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ks++) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf[ks], a0, 0, 0, 0);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float v = a0[e] + b0[e];
        if (p.act0) v = v > 0.f ? v : 0.f;
        o[e] = (bf16)(inside ? v : 0.f);
      }
      // channel nb*16 + lq*4 + e of patch pixel j: chunk nb >> 1, 16-B slot (nb & 1)*2 + (lq >> 1), swizzled
      const int slot = ((nb & 1) * 2 + (lq >> 1)) ^ (((j >> 2) & 1) << 1);
      if (j < 324) *(bf16x4*)(lds + (nb >> 1) * patch_bytes + j * 64 + slot * 16 + (lq & 1) * 8) = o;
    }
    // (the first K step's barrier makes the patch visible to every wave)
  }
  if (dbg_t) t1 = clock64();

  bf16x8 af[IB], bfr[NREP];
#pragma unroll
  for (int i = 0; i < IB; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) af[i][e] = (bf16)0.f;
#pragma unroll
  for (int j = 0; j < NREP; j++)
#pragma unroll
    for (int e = 0; e < 8; e++) bfr[j][e] = (bf16)0.f;

  if constexpr (STEM) {
    // Both patch chunks are resident and only 4 KB weight slices stream, so the K loop is barrier-bound:
    // one barrier per KERNEL ROW (3 taps, 24 MFMAs per wave) instead of per tap; weight slices of row s+1
    // are fetched into the other half of the 6-slice buffer while row s computes.
#pragma unroll 1
    for (int s = 0; s < 6; s++) {
      const int cc = s / 3, kh = s - cc * 3;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 1 < 6) {
        const int c1 = (s + 1) / 3, k1 = (s + 1) - c1 * 3;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) load_b(k1 * 3 + kw, c1, ((s + 1) & 1) * 3 + kw);
      }
      const char* patch = lds + cc * patch_bytes;
#pragma unroll
      for (int kw = 0; kw < 3; kw++) {
        const char* bb = bring + ((s & 1) * 3 + kw) * B_STAGE + b_rd;
#pragma unroll
        for (int j = 0; j < NREP; j++) bfr[j] = *(const bf16x8*)(bb + j * 1024);
#pragma unroll
        for (int i = 0; i < IB; i++) {
          const int pj = pjl + kh * 18 + kw + i * 18;
          af[i] = *(const bf16x8*)(patch + pj * 64 + ((fq * 16) ^ ((pj & 4) << 3)));
        }
#pragma unroll
        for (int i = 0; i < IB; i++)
#pragma unroll
          for (int j = 0; j < NREP; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  }
  if (LINC && LIN) {
    // prologue issued, in this order: patch of chunk 0 (NPC per wave), weight slice 0 (NBW), weight slice 1 (NBW)
    const char* const bb0 = bring + b_rd;
    const int astep = SI * 64;  // row block i of a tap sits SI pixels further on (SI = 16 or 24: the swizzle bit, bit 2
                                // of the pixel index, is the same for every i)
#pragma unroll 1
    for (int cc = 0; cc < nchunks; cc++) {
      const char* patch = lds + (cc & 1) * patch_bytes;
      const int nbuf = (cc + 1) & 1;
      const int cn = cc + 1 < nchunks ? cc + 1 : cc;  // past the end: re-fetch this chunk into the free buffer
#pragma unroll
      for (int t = 0; t < 9; t++) {
        // everything issued before the previous step has landed (previous step: NBW weight + 1 patch piece,
        // none of the latter in tap 8)
        PH_WAIT_VMCNT(NBW + (((t + 8) % 9) < NPC ? 1 : 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* bb = bb0 + (t % 3) * B_STAGE;
        int pv = pjl;
        asm volatile("" : "+v"(pv));  // (keeps the 9 tap offsets out of registers the accumulators need)
        const int pj0 = pv + (t / 3) * p.SKH + (t % 3);
        const char* ap = patch + pj0 * 64 + ((fq * 16) ^ ((pj0 & 4) << 3));
        // weight slice of step s+2 -> slot (t+2)%3 (free since the barrier); piece t of the next chunk's patch
        int t2 = t + 2, c2n = cc;
        if (t2 >= 9) {
          t2 -= 9;
          c2n = cn;
        }
#pragma unroll
        for (int j = 0; j < NREP; j++) bfr[j] = *(const bf16x8*)(bb + j * 1024);
#pragma unroll
        for (int i = 0; i < IB; i++) af[i] = *(const bf16x8*)(ap + i * astep);
        load_b(t2, c2n, (t + 2) % 3, true);
        if (t < NPC) load_a(t, cn, nbuf, true);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < IB; i++)
#pragma unroll
          for (int j = 0; j < NREP; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // (or the MFMAs sink below the next barrier and the fragments double up)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-fetches past the end of K
  }
  int bslot = 0;
  int pbuf = 0;  // patch buffer of chunk cc
  for (int cc = 0; !STEM && !(LINC && LIN) && cc < nchunks; cc++) {
    const char* patch = lds + pbuf * patch_bytes;
#pragma unroll
    for (int t = 0; t < TAPS; t++) {
      const int s = cc * TAPS + t;
      long long ta = 0;
      if (dbg_t) ta = clock64();
      // everything issued before the previous step has landed
      PH_WAIT_VMCNT(prev_cnt);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (dbg_t) tw += clock64() - ta;
      // issue: weight slice of step s+2, piece t of the next chunk's patch
      prev_cnt = 0;
      if (s + 2 < ns && !(DBG && (p.debug & 2))) {
        int t2n = t + 2, c2n = cc;
        while (t2n >= TAPS) { t2n -= TAPS; c2n++; }
        int sl = bslot + 2;
        sl = sl >= NBS ? sl - NBS : sl;
        load_b(t2n, c2n, sl);
        prev_cnt = nb_w;
      }
      if constexpr (TAPS == 1) {  // the whole patch of chunk cc+2
        if (cc + 2 < nchunks) {
          int nb2 = pbuf + 2;
          nb2 = nb2 >= NPB ? nb2 - NPB : nb2;
#pragma unroll
          for (int u = 0; u < NPC; u++) prev_cnt += load_a(u, cc + 2, nb2);
        }
      } else {
        if (!STEM && t < NPC && cc + 1 < nchunks && !(DBG && (p.debug & 1)))
          prev_cnt += load_a(t, cc + 1, (cc + 1) & 1);
      }

      // (the asm keeps the 72 (tap, i) read addresses from being hoisted out of the chunk loop
      // into registers the accumulators need)
      int pv = pjl;
      asm volatile("" : "+v"(pv));
      const int pj0 = TAPS == 1 ? pv : pv + (t / 3) * p.SKH + (t % 3);
      const char* bb = bring + bslot * B_STAGE + b_rd;
      if constexpr (PIPE) {
        // Software pipeline across the barrier: the fragments of step s are read here, right
        // after the barrier that makes them visible, while the MFMAs issued in between are those
        // of step s-1 (fragments already in registers); A fragment i is re-read into the
        // registers the MFMAs of row block i have just consumed.  The first pass multiplies
        // zero fragments.
        bf16x8 bnx[NREP];
#pragma unroll
        for (int j = 0; j < NREP; j++) bnx[j] = *(const bf16x8*)(bb + j * 1024);
#pragma unroll
        for (int i = 0; i < IB; i++) {
          if (!(DBG && (p.debug & 4))) {
#pragma unroll
            for (int j = 0; j < NREP; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
          } else {
            asm volatile("" ::"v"(af[i]));
          }
          const int pj = pj0 + i * SI;
          af[i] = *(const bf16x8*)(patch + pj * 64 + ((fq * 16) ^ ((pj & 4) << 3)));
        }
        // pin the interleave (NREP MFMAs, one A-fragment read, ...): left alone the scheduler
        // batches all reads behind all MFMAs and the next step starts with their latency exposed
#pragma unroll
        for (int i = 0; i < IB; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, NREP, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int j = 0; j < NREP; j++) bfr[j] = bnx[j];
      } else {
        // all 8 + NREP fragment reads of the step are issued up front and the MFMAs wait for
        // them one by one (counted lgkmcnt); the sched_barrier keeps the compiler from sinking
        // the reads next to their uses
#pragma unroll
        for (int j = 0; j < NREP; j++) bfr[j] = *(const bf16x8*)(bb + j * 1024);
#pragma unroll
        for (int i = 0; i < IB; i++) {
          const int pj = pj0 + i * SI;
          af[i] = *(const bf16x8*)(patch + pj * 64 + ((fq * 16) ^ ((pj & 4) << 3)));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG && (p.debug & 4))) {
#pragma unroll
          for (int i = 0; i < IB; i++)
#pragma unroll
            for (int j = 0; j < NREP; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < IB; i++) asm volatile("" ::"v"(af[i]));
        }
      }
      bslot = bslot + 1 == NBS ? 0 : bslot + 1;
    }
    pbuf = pbuf + 1 == NPB ? 0 : pbuf + 1;
  }
  if (PIPE && !STEM && !(LINC && LIN)) {  // MFMAs of the last step
#pragma unroll
    for (int i = 0; i < IB; i++)
#pragma unroll
      for (int j = 0; j < NREP; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
  }
  if (dbg_t) t2 = clock64();

  // ---- epilogue --------------------------------------------------------------
  // acc[i][j][e]: pixel row (wm*IB*16 + i*16 + fr), channel (wn*BN/2 + j*16 + fq*4 + e)
  const int cbase = wn * (BN / 2) + fq * 4;
#pragma unroll
  for (int j = 0; j < NREP; j++) {
    const f32x4 bv = *(const f32x4*)(s_bias + cbase + j * 16);
    f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.act == 2) sv = *(const f32x4*)(s_slope + cbase + j * 16);
#pragma unroll
    for (int i = 0; i < IB; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float v = acc[i][j][e] + bv[e];
        if (p.act) v = v > 0.f ? v : v * sv[e];
        acc[i][j][e] = v;
      }
  }
  __syncthreads();  // all fragment reads done, no LDS-DMA in flight (the last step waited for vmcnt(0))
  // per output row: {batch index or -1, y << 16 | x} (interior coordinates)
  int2* s_row = (int2*)(lds + HALO_BM * CROWB);  // behind the staged output tile (which may cover every other region)
  if (tid >= HALO_BM) {
  } else if (TAPS == 1) {  // output row tid is unpadded pixel index Qc0 + tid
    const int m = Qc0 + tid;
    const int hw = p.H * p.W;
    const int mc = m < p.M ? m : p.M - 1;
    const int b = p.magic_e1 ? (int)__umulhi((unsigned)mc, p.magic_e1) : mc / hw;
    const int rem = mc - b * hw;
    const int y = p.magic_e1 ? (int)__umulhi((unsigned)rem, p.magic_e2) : rem / p.W;
    s_row[tid] = make_int2(m < p.M ? b : -1, (y << 16) | (rem - y * p.W));
  } else if (p.halo_mode == 0) {  // output row tid is padded-linear pixel Qc0 + tid: border pixels are dropped
    const int Qc = Qc0 + tid;
    const int HWp = Hp * Wp;
    const int b = p.magic_e1 ? (int)__umulhi((unsigned)Qc, p.magic_e1) : Qc / HWp;
    const int rem = Qc - b * HWp;
    const int yp = p.magic_e1 ? (int)__umulhi((unsigned)rem, p.magic_e2) : rem / Wp;
    const int xp = rem - yp * Wp;
    const bool ok = b < p.B && yp >= pi && yp < p.H + pi && xp >= pi && xp < p.W + pi;
    s_row[tid] = make_int2(ok ? b : -1, ((yp - pi) << 16) | ((xp - pi) & 0xffff));
  } else {                 // output row tid is pixel (tid >> 4, tid & 15) of the 16x16 tile
    const int y = ty0 + (tid >> 4), x = tx0 + (tid & 15);
    const bool ok = tb < p.B && y < p.H && x < p.W;
    s_row[tid] = make_int2(ok ? tb : -1, (y << 16) | x);
  }
  constexpr int VPR = BN / 8;
  const int nvalid = p.cout - n0;
  if (p.f32_out) {  // network output: fp32 NCHW straight from the accumulators (16 consecutive pixels per store)
    __syncthreads();
    const int hw = p.H * p.W;
#pragma unroll
    for (int i = 0; i < IB; i++) {
      const int2 ri = s_row[wm * IB * 16 + i * 16 + fr];
      if (ri.x < 0) continue;
      float* o = p.f32_out + ((size_t)ri.x * p.f32_C + p.f32_coff + n0 + cbase) * hw + (ri.y >> 16) * p.W + (ri.y & 0xffff);
#pragma unroll
      for (int j = 0; j < NREP; j++)
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (cbase + j * 16 + e < nvalid) o[(size_t)(j * 16 + e) * hw] = acc[i][j][e];
    }
  }
  if (p.n_dst > 0) {
    // the whole 256 x BN tile goes through LDS at once (bias/slope and the K-loop buffers are dead by now)
#pragma unroll
    for (int i = 0; i < IB; i++)
#pragma unroll
      for (int j = 0; j < NREP; j++) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (bf16)acc[i][j][e];
        *(bf16x4*)(lds + (wm * IB * 16 + i * 16 + fr) * CROWB + (cbase + j * 16) * 2) = o;
      }
    __syncthreads();
    if (p.pool) {
      // 2x2/2 max pool of the 16x16 tile -> 8 x 8 pooled pixels
      // (destination loop outside: its descriptor is read once, not once per row)
      for (int d = 0; d < p.n_dst; d++) {
        const PhDst ds = p.dst[d];
        bf16* const dbase = ds.base + (size_t)ds.org + n0;
        for (int v = tid; v < 64 * VPR; v += NT) {
          const int prow = v / VPR;
          const int cv = (v - prow * VPR) * 8;
          const int r0 = (prow >> 3) * 32 + (prow & 7) * 2;
          const int2 ri = s_row[r0];
          if (ri.x < 0 || cv >= nvalid) continue;
          const int y = ri.y >> 17, x = (ri.y & 0xffff) >> 1;
          const char* sp = lds + r0 * CROWB + cv * 2;
          const bf16x8 v00 = *(const bf16x8*)sp, v01 = *(const bf16x8*)(sp + CROWB);
          const bf16x8 v10 = *(const bf16x8*)(sp + 16 * CROWB), v11 = *(const bf16x8*)(sp + 17 * CROWB);
          bf16x8 mx;
#pragma unroll
          for (int e = 0; e < 8; e++)
            mx[e] = (bf16)fmaxf(fmaxf((float)v00[e], (float)v01[e]), fmaxf((float)v10[e], (float)v11[e]));
          bf16* o = dbase + ((size_t)(ri.x * ds.Hp + y) * ds.Wp + x) * ds.Cs + cv;
          if (cv + 8 <= nvalid) {
            *(bf16x8*)o = mx;
          } else {
            for (int e = 0; e < nvalid - cv; e++) o[e] = mx[e];
          }
        }
      }
    } else {
      for (int d = 0; d < p.n_dst; d++) {
        const PhDst ds = p.dst[d];
        bf16* const dbase = ds.base + (size_t)ds.org + n0;
        // Full 16-byte stores go out as system-scope write-through buffer stores: the L2 of an XCD is written back
        // at the end of every kernel (the L2s are not coherent with each other), and lines that are already in
        // memory by then shorten that tail (-1.4 us per 128->128 launch on its own; neutral with batches in flight)
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, 0x7fffffff, 0x00027000);
#pragma unroll 4
        for (int v = tid; v < HALO_BM * VPR; v += NT) {
          const int row = v / VPR;
          const int cv = (v - row * VPR) * 8;
          const int2 ri = s_row[row];
          if (ri.x < 0 || cv >= nvalid) continue;
          const int y = ri.y >> 16, x = ri.y & 0xffff;
          const char* sp = lds + row * CROWB + cv * 2;
          const u32x4 val = *(const u32x4*)sp;
          bf16* o = dbase + ((size_t)(ri.x * ds.Hp + y) * ds.Wp + x) * ds.Cs + cv;
          if (cv + 8 <= nvalid) {
            __builtin_amdgcn_raw_buffer_store_b128(val, drsrc, (((ri.x * ds.Hp + y) * ds.Wp + x) * ds.Cs + cv) * 2, 0,
                                                   PH_AUX_SC0_SC1);
          } else {
            for (int e = 0; e < nvalid - cv; e++) o[e] = ((const bf16*)sp)[e];
          }
        }
      }
    }
  }
  if (dbg_t) {
    __syncthreads();
    if (tid == 0) {
      const long long t3 = clock64();
      atomicAdd(&ph_conv_dbg[0], (unsigned long long)(t1 - t0));
      atomicAdd(&ph_conv_dbg[1], (unsigned long long)(t2 - t1));
      atomicAdd(&ph_conv_dbg[2], (unsigned long long)(t3 - t2));
      atomicAdd(&ph_conv_dbg[3], (unsigned long long)tw);
      atomicAdd(&ph_conv_dbg[4], 1ull);
    }
  }
}

static bool no_lin_env() {  // POSEHIP_NO_LIN=1: A/B switch, the generic K loop everywhere
  static const char* e = getenv("POSEHIP_NO_LIN");
  return e && e[0] == '1';
}

// The epilogue's row table splits a pixel index < n_max by d1 (pixels per image), the rest by d2 (pixels per row):
// multiply-high by ceil(2^32 / d) is exact while n * d < 2^32 (common.h); otherwise the kernel divides.
static void ph_epilogue_magic(PhConvParams& p, unsigned long long n_max, int d1, int d2) {
  const bool ok = d1 > 1 && d2 > 1 && n_max * (unsigned long long)d1 < 0x100000000ull &&
                  (unsigned long long)d1 * (unsigned long long)d2 < 0x100000000ull;
  p.magic_e1 = ok ? ph_magic((unsigned)d1) : 0u;
  p.magic_e2 = ok ? ph_magic((unsigned)d2) : 0u;
}

// Geometry of the halo kernel for one conv; returns false if the op does not fit it.
bool ph_conv_halo_plan(PhConvParams& p, int kh, bool has_f32) {
  p.magic_e1 = p.magic_e2 = 0;
  for (int d = 0; d < p.n_dst; d++)   // the epilogue's buffer stores address a destination with 31-bit byte offsets
    if ((unsigned long long)p.B * p.dst[d].Hp * p.dst[d].Wp * p.dst[d].Cs * 2ull >= 0x7fffffffull) return false;
  if (kh == 1 && p.KW == 1) {  // 1x1: halo_mode 2, TAPS = 1 instantiation
    if ((p.cin_read & 31) || p.pool || p.stem || (!has_f32 && p.n_dst < 1)) return false;
    if ((unsigned long long)p.NQ * p.in_Cs * 2ull >= 0xffffffffull ||
        (unsigned long long)p.cout_pad * p.Kpad * 2ull >= 0xffffffffull)
      return false;
    p.halo_mode = 2;
    p.lin = 0;
    p.NI = 16;
    p.SKH = 0;
    p.SI = 16;
    p.PR = 1;
    p.PCLIP = 0;
    p.magic_pr = ph_magic(1);
    p.n_mtiles = (p.M + HALO_BM - 1) / HALO_BM;
    ph_epilogue_magic(p, (unsigned long long)p.n_mtiles * HALO_BM, p.H * p.W, p.W);
    return true;
  }
  if (kh != 3 || p.KW != 3 || (p.cin_read & 31) || has_f32 || p.in_pad < 1 || p.n_dst < 1) return false;
  // the kernel addresses input and weights with 32-bit byte offsets
  if ((unsigned long long)p.NQ * p.in_Cs * 2ull >= 0xffffffffull ||
      (unsigned long long)p.cout_pad * p.Kpad * 2ull >= 0xffffffffull)
    return false;
  const int Wp = p.in_Wp, Hp = p.in_Hp;
  const int np_lin = 2 * (Wp < HALO_BM + 2 ? Wp : HALO_BM + 2) + HALO_BM + 2;
  if (p.pool && ((p.H | p.W) & 1)) return false;
  static const char* force = getenv("POSEHIP_HALO_MODE");  // tests: "1" forces the 16x16-tile form
  const bool tiles2d = p.pool || p.stem || (force && force[0] == '1');
  if (!tiles2d && Wp < HALO_BM + 2 && (np_lin + 15) / 16 <= 32) {
    p.halo_mode = 0;
    p.q_first = p.in_pad * Wp + p.in_pad;
    p.SKH = Wp;
    p.SI = 16;
    p.PR = Wp;
    p.PCLIP = 2;
    p.NI = (np_lin + 15) / 16;
    const long long q_last = ((long long)(p.B - 1) * Hp + p.H - 1 + p.in_pad) * Wp + p.W - 1 + p.in_pad;
    p.n_mtiles = (int)((q_last - p.q_first + HALO_BM) / HALO_BM);
    ph_epilogue_magic(p, (unsigned long long)p.q_first + (unsigned long long)p.n_mtiles * HALO_BM, Hp * Wp, Wp);
  } else {
    p.halo_mode = 1;
    p.tiles_x = (p.W + 15) / 16;
    p.tiles_y = (p.H + 15) / 16;
    // 18 x 18 patch.  Outside the stem kernel its rows are kept 24 pixels apart (columns 18..23 are loaded and
    // never read): a row-block stride that is a multiple of 8 pixels is what the branch-free K loop needs, and
    // 27 pieces fit the same LDS budget as the padded-linear form.  The stem kernel builds a dense 18 x 18 patch.
    const int pr = (p.stem || no_lin_env()) ? 18 : 24;
    p.SKH = pr;
    p.SI = pr;
    p.PR = pr;
    p.PCLIP = 17;
    p.NI = (18 * pr + 15) / 16;
    p.n_mtiles = p.B * p.tiles_x * p.tiles_y;
  }
  p.magic_pr = ph_magic(p.PR);
  p.lin = !no_lin_env() && !p.stem;
  return true;
}

// K-loop buffers, or the epilogue's staged 256 x BN bf16 tile + 256-entry row table, whichever is larger
static int halo_lds(int loop_bytes, int bn) {
  const int epi = HALO_BM * (bn * 2 + 16) + HALO_BM * 8;
  return loop_bytes > epi ? loop_bytes : epi;
}

template <int NREP, int MW, bool DBG>
static int launch_halo(const PhConvParams& p, hipStream_t stream) {
  constexpr int BN = 32 * NREP;
  const int ldsz = halo_lds(2 * p.NI * 1024 + 3 * (BN / 16) * 1024 + 1024, BN);
  static int attr[PH_MAX_DEVICES];
  if (ph_ensure_lds((const void*)conv3x3_halo_kernel<NREP, MW, DBG>, ldsz, attr)) return -1;
  PhConvParams q = p;
  q.n_ntiles = (p.cout + BN - 1) / BN;
  hipLaunchKernelGGL((conv3x3_halo_kernel<NREP, MW, DBG>), dim3(q.n_mtiles * q.n_ntiles), dim3(128 * MW), ldsz, stream,
                     q);
  PH_CHECK(hipGetLastError());
  return 0;
}

template <int NREP, int MW>
static int launch_halo_1x1_mw(const PhConvParams& p, hipStream_t stream) {
  constexpr int BN = 32 * NREP;
  const int ldsz = halo_lds(3 * 16 * 1024 + 3 * (BN / 16) * 1024 + 1024, BN);
  static int attr[PH_MAX_DEVICES];
  if (ph_ensure_lds((const void*)conv3x3_halo_kernel<NREP, MW, false, false, 1>, ldsz, attr)) return -1;
  PhConvParams q = p;
  q.n_ntiles = (p.cout + BN - 1) / BN;
  hipLaunchKernelGGL((conv3x3_halo_kernel<NREP, MW, false, false, 1>), dim3(q.n_mtiles * q.n_ntiles), dim3(128 * MW), ldsz,
                     stream, q);
  PH_CHECK(hipGetLastError());
  return 0;
}

// 1x1 layers: 8-wave blocks (MW = 4) measured 11 % faster than 4-wave ones on 384->512 and 512->52;
// POSEHIP_CONV_1X1_MW=2 brings the 4-wave form back for A/B runs
int ph_conv_1x1_mw() {
  static const char* e = getenv("POSEHIP_CONV_1X1_MW");
  return e && e[0] == '2' ? 2 : 4;
}

template <int NREP>
static int launch_halo_1x1(const PhConvParams& p, hipStream_t stream) {
  if (ph_conv_1x1_mw() == 4) return launch_halo_1x1_mw<NREP, 4>(p, stream);
  return launch_halo_1x1_mw<NREP, 2>(p, stream);
}

static int launch_halo_stem(const PhConvParams& p, hipStream_t stream) {
  const int ldsz = halo_lds(2 * p.NI * 1024 + 6 * 4 * 1024 + 1024 + 7 * 1024, 64);
  static int attr[PH_MAX_DEVICES];
  if (ph_ensure_lds((const void*)conv3x3_halo_kernel<2, 4, false, true>, ldsz, attr)) return -1;
  PhConvParams q = p;
  q.n_ntiles = 1;
  hipLaunchKernelGGL((conv3x3_halo_kernel<2, 4, false, true>), dim3(q.n_mtiles), dim3(512), ldsz, stream, q);
  PH_CHECK(hipGetLastError());
  return 0;
}

template <int NREP>
static int launch_halo_sel(const PhConvParams& p, int mw, hipStream_t stream) {
  if (mw == 4) return p.debug ? launch_halo<NREP, 4, true>(p, stream) : launch_halo<NREP, 4, false>(p, stream);
  return p.debug ? launch_halo<NREP, 2, true>(p, stream) : launch_halo<NREP, 2, false>(p, stream);
}

// variant 0: im2col-gather kernel (any odd kernel size / channel count).  variant 4 / 5: 3x3 halo-patch kernel with
// 4 / 8 waves, 6: 1x1 through the same pipeline (geometry from ph_conv_halo_plan).
bool ph_stem512_applies(const PhConvParams& p);          // conv_stem.hip: the persistent stem kernel
int ph_launch_stem512(const PhConvParams& p, hipStream_t stream);

int ph_launch_conv(const PhConvParams& p, int nrep, int variant, hipStream_t stream) {
  if (p.stem) return ph_stem512_applies(p) ? ph_launch_stem512(p, stream) : launch_halo_stem(p, stream);
  if (variant == 6) {  // 1x1 through the halo pipeline
    switch (nrep) {
      case 1: return launch_halo_1x1<1>(p, stream);
      case 2: return launch_halo_1x1<2>(p, stream);
      case 3: return launch_halo_1x1<3>(p, stream);
      case 4: return launch_halo_1x1<4>(p, stream);
      default: ph_set_error("ph_launch_conv: bad nrep %d", nrep); return -1;
    }
  }
  if (variant == 4 || variant == 5) {  // 5: 8-wave blocks
    const int mw = variant == 5 ? 4 : 2;
    switch (nrep) {
      case 1: return launch_halo_sel<1>(p, mw, stream);
      case 2: return launch_halo_sel<2>(p, mw, stream);
      case 3: return launch_halo_sel<3>(p, mw, stream);
      case 4: return launch_halo_sel<4>(p, mw, stream);
      default: ph_set_error("ph_launch_conv: bad nrep %d", nrep); return -1;
    }
  }
  const int bn = 32 * nrep;
  PhConvParams q = p;
  q.n_mtiles = (p.M + BM - 1) / BM;
  q.n_ntiles = (p.cout + bn - 1) / bn;
  dim3 grid(q.n_mtiles * q.n_ntiles), block(NTHREADS);
  switch (nrep) {
    case 1: hipLaunchKernelGGL(conv_igemm_kernel<1>, grid, block, 0, stream, q); break;
    case 2: hipLaunchKernelGGL(conv_igemm_kernel<2>, grid, block, 0, stream, q); break;
    case 3: hipLaunchKernelGGL(conv_igemm_kernel<3>, grid, block, 0, stream, q); break;
    case 4: hipLaunchKernelGGL(conv_igemm_kernel<4>, grid, block, 0, stream, q); break;
    default: ph_set_error("ph_launch_conv: bad nrep %d", nrep); return -1;
  }
  PH_CHECK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// 2x2 stride-2 max pooling, bf16 NHWC padded -> padded (Caffe Pooling MAX,
// ceil mode: odd trailing row/column pools over what exists).
// ---------------------------------------------------------------------------
__global__ void maxpool2x2_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int B, int H, int W,
                                  int C8, int in_Hp, int in_Wp, int in_Cs, int in_org, int oH, int oW,
                                  int out_Hp, int out_Wp, int out_Cs, int out_org) {
  const size_t total = (size_t)B * oH * oW * C8;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c8 = idx % C8;
    size_t t = idx / C8;
    const int ox = t % oW;
    t /= oW;
    const int oy = t % oH;
    const int b = t / oH;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; e++) m[e] = -3.0e38f;
    for (int dy = 0; dy < 2; dy++)
      for (int dx = 0; dx < 2; dx++) {
        const int y = oy * 2 + dy, x = ox * 2 + dx;
        if (y < H && x < W) {
          const bf16x8 v =
              *(const bf16x8*)(in + (size_t)in_org + ((size_t)(b * in_Hp + y) * in_Wp + x) * in_Cs + c8 * 8);
#pragma unroll
          for (int e = 0; e < 8; e++) m[e] = fmaxf(m[e], (float)v[e]);
        }
      }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (bf16)m[e];
    *(bf16x8*)(out + (size_t)out_org + ((size_t)(b * out_Hp + oy) * out_Wp + ox) * out_Cs + c8 * 8) = o;
  }
}

int ph_launch_maxpool(const bf16* in, bf16* out, int B, int H, int W, int C, int in_Hp, int in_Wp, int in_Cs,
                      int in_org, int oH, int oW, int out_Hp, int out_Wp, int out_Cs, int out_org,
                      hipStream_t stream) {
  const size_t total = (size_t)B * oH * oW * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(blocks), dim3(256), 0, stream, in, out, B, H, W, C / 8, in_Hp, in_Wp,
                     in_Cs, in_org, oH, oW, out_Hp, out_Wp, out_Cs, out_org);
  PH_CHECK(hipGetLastError());
  return 0;
}
