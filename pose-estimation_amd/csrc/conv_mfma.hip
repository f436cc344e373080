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

#define BM 128
#define BK 64
#define NTHREADS 256

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
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_global_load_lds((glb_void*)(p.in + a_off[i] + off), (lds_void*)(sa + i * 4096), 16, 0, 0);
    char* sb = lds + buf * STAGE + A_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < NREP; i++)
      __builtin_amdgcn_global_load_lds((glb_void*)(p.wt + w_off[i] + k), (lds_void*)(sb + i * 4096), 16, 0, 0);
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
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NREP; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: one barrier per K tile -----------------------------------
  const int nk = p.nk;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nk - 1; kt++) {
    stage(kt + 1, buf ^ 1);
    compute(buf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA of tile kt+1 landed
    __syncthreads();
    buf ^= 1;
  }
  compute(buf);

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
}

int ph_launch_conv(const PhConvParams& p, int nrep, hipStream_t stream) {
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
