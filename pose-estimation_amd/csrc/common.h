// Shared device/host definitions for libposehip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PH_MAX_DST 8

// One bf16 NHWC destination of a conv epilogue (a channel window of a padded
// activation buffer).
struct PhDst {
  bf16* base;
  int Hp, Wp, Cs;   // padded height/width, channel stride of the buffer
  int org;          // element offset of interior pixel (0,0) channel window start
};

struct PhConvParams {
  const bf16* in;    // padded NHWC input buffer
  const bf16* wt;    // [CoutPad][Kpad] bf16, k = (kh*KW+kw)*cin_read + c
  const float* bias; // [CoutPad]
  const float* slope;// [CoutPad] PReLU slopes (act == 2)
  int B, H, W, M;    // output pixels M = B*H*W
  int in_Hp, in_Wp, in_Cs;
  int in_org;        // element offset of tap (0,0) of output pixel (0,0,0)
  int cin_read, KW, K, Kpad, nk;
  unsigned magic_cin, magic_kw;
  int cout, cout_pad, act;  // act: 0 none, 1 ReLU, 2 PReLU; cout_pad = rows of wt / bias / slope
  int n_dst;
  PhDst dst[PH_MAX_DST];
  float* f32_out;    // optional NCHW fp32 [B][f32_C][H][W] written at f32_coff
  int f32_C, f32_coff;
  int n_mtiles, n_ntiles;
  // 3x3 halo-patch kernel (conv3x3_halo_kernel) geometry, filled by ph_net_build
  // stem fusion (conv3x3_halo_kernel<.., STEM>): the PRODUCER of this conv's input is a 3x3 conv over an
  // 8-channel buffer (conv1_1); its output patch is computed in LDS instead of being read.  `in`, in_Hp,
  // in_Wp, in_pad, NQ then describe the 8-channel buffer.
  int stem;
  const bf16* wt0;       // [64][Kpad0] producer weights, k = tap*8 + c
  const float* bias0;    // [64]
  int Kpad0, act0;       // act0: 0 none, 1 ReLU
  int cin0_span;         // 1 + the last channel of the 8-channel buffer that the producer's weights use
  int* tile_ctr;         // persistent stem kernel (conv_stem.hip): two tile counters, used alternately per launch
  int tile_par;          //   which of the two this launch counts on (the kernel zeroes the other one)
  int pool;              // 1: 2x2/2 max pool fused into the epilogue (dst = pooled geometry), mode 1 only
  int in_pad, in_coff;   // border width of the input buffer, first channel of the read window
  int NQ;                // pixels of the padded input buffer (B * Hp * Wp)
  int halo_mode;         // 0: tiles are 256-pixel runs of the padded linear domain, 1: 16x16 pixel tiles
  int q_first;           // mode 0: padded linear index of output pixel (0,0,0)
  int tiles_x, tiles_y;  // mode 1
  int SKH, SI;           // patch pixel of (row block i, px, kh, kw) = kh*SKH + i*SI + kw + px
  int PR, PCLIP, NI;     // patch pixel j <-> input pixel Qbase + j + min(j/PR, PCLIP)*(Wp-PR); NI 16-pixel pieces
  unsigned magic_pr;
  unsigned magic_e1, magic_e2;   // epilogue row table: ceil(2^32/d) of its two divisors (image size, row length), 0 = divide
  int lin;    // halo_mode 0 at BN = 128: branch-free K loop (POSEHIP_NO_LIN=1 turns it off for A/B runs)
  int debug;  // timing experiments only (POSEHIP_CONV_DEBUG): 1 skip A loads, 2 skip B loads, 4 skip MFMA
};

// Two consecutive 1x1 convs as one kernel (conv_pair.hip): `a` is the wide first layer (input, weights, bias,
// activation; its destination is never written), the rest describes the narrow second layer (<= 64 channels).
struct PhPairParams {
  PhConvParams a;
  const bf16* wt2;      // [cout_pad2][Kpad2], k = channel of the first layer's output
  const float* bias2;
  const float* slope2;
  int Kpad2, cout2, act2;
  int n_dst2;
  PhDst dst2[PH_MAX_DST];
  float* f32_out2;
  int f32_C2, f32_coff2;
};

__host__ __device__ static inline unsigned ph_magic(unsigned d) {
  // ceil(2^32 / d): __umulhi(n, magic) == n / d whenever n * d < 2^32
  return (unsigned)((0x100000000ull + d - 1) / d);
}
// n / d for n*d < 2^32 (true for every operand pair used here; checked on host)
__device__ static inline unsigned ph_div(unsigned n, unsigned magic, unsigned d) {
  return d == 1 ? n : __umulhi(n, magic);
}

#define PH_CHECK(expr)                                                        \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      ph_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,               \
                   hipGetErrorString(_e));                                    \
      return -1;                                                              \
    }                                                                         \
  } while (0)

void ph_set_error(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember, per kernel and
// device, the largest size already granted (`cache` = a zero-initialised static int[PH_MAX_DEVICES] per kernel).
#define PH_MAX_DEVICES 64
static inline int ph_ensure_lds(const void* fn, int bytes, int* cache) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PH_MAX_DEVICES) dev = 0;
  if (cache[dev] >= bytes) return 0;
  PH_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  cache[dev] = bytes;
  return 0;
}
