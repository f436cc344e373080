// C ABI of libposehip (declared in include/posehip.h): context, network plan
// execution, decode entry points.  Host-side only; kernels live in
// conv_mfma.hip / preprocess.hip / decode_v2.hip.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/posehip.h"
#include "common.h"
#include "decode_v1.h"
#include "decode_v2.h"

int ph_launch_conv(const PhConvParams& p, int nrep, int variant, hipStream_t stream);
bool ph_conv_halo_plan(PhConvParams& p, int kh, bool has_f32);
int ph_launch_torso_bbox(const float* kp, const int* n_people, int B, int maxp, int J, const int* ids, int n_ids,
                         float* out, hipStream_t stream);
int ph_launch_pose_features(const float* net_out, int C, int hs, int ws, const float* bbox, const int* n_people, int B,
                            int maxp, const int* windows, int n_win, int n_heat, int mode, int Ht, int Wt, int img_w,
                            int img_h, int fx, int fy, int capacity, float* feat, int* status, int* total,
                            hipStream_t stream);
int ph_launch_movenet_crop_resize(const uint8_t* frames, int B, int H, int W, const double* crop, int S, int32_t* out,
                                  hipStream_t stream);
int ph_launch_movenet_update(float* kp, const double* crop_in, int B, int ih, int iw, double* crop_out,
                             hipStream_t stream);
int ph_launch_movenet_multipose(const float* center, const float* heat, const float* regr, const float* offs,
                                const float* boxs, int B, int G, int maxp, float min_center, int* c_idx,
                                float* c_score, float* out, hipStream_t stream);
int ph_conv_1x1_mw();
bool ph_stem512_applies(const PhConvParams& p);
bool ph_pair_enabled();                                           // conv_pair.hip: two 1x1 convs as one kernel
int ph_launch_pair(const PhPairParams& q, hipStream_t stream);
int ph_launch_lift3d(const float* kp, const int* n_people, int B, int maxp, int J, const uint16_t* depth, int Hd,
                     int Wd, double fx, double fy, double cx, double cy, double depth_scale, int po, double* out,
                     hipStream_t stream);
int ph_launch_maxpool(const bf16* in, bf16* out, int B, int H, int W, int C, int in_Hp, int in_Wp, int in_Cs,
                      int in_org, int oH, int oW, int out_Hp, int out_Wp, int out_Cs, int out_org,
                      hipStream_t stream);
int ph_launch_preprocess(const uint8_t* frames, int B, int H0, int W0, int Hn, int Wn, int mode, bf16* out,
                         int out_Hp, int out_Wp, int out_Cs, int out_org, hipStream_t stream);

static thread_local char g_err[1024] = "";

// Workspace of ph_decode_v2_noalloc (the one entry point that allocates nothing).  In host code the name `scratch`
// is only a handle for hipGetSymbolAddress: the device address of each GPU's own copy is looked up once per device
// in ph_create (that device current) and kept in g_scratch_ptr; host code uses that pointer everywhere.
#define PH_SCRATCH_BYTES (20u << 20)
static __device__ __align__(256) unsigned char scratch[PH_SCRATCH_BYTES];
static void* g_scratch_ptr[PH_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_init(int device) {
  if (device < 0 || device >= PH_MAX_DEVICES) {
    ph_set_error("device %d beyond the %d the library keeps a scratch pointer for", device, PH_MAX_DEVICES);
    return -1;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[device]) return 0;
  void* p = nullptr;
  PH_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));  // the current device's copy
  g_scratch_ptr[device] = p;
  return 0;
}

void ph_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct Buf {
  ph_buf_desc d;
  bf16* ptr = nullptr;
  size_t elems = 0;
  int Hp() const { return d.height + 2 * d.pad; }
  int Wp() const { return d.width + 2 * d.pad; }
  // element offset of interior pixel (0,0), shifted by (dy,dx) pixels, channel c
  int org(int shift, int c) const { return ((d.pad + shift) * Wp() + d.pad + shift) * d.channels + c; }
};

struct Op {
  ph_op_desc d;
  PhConvParams cp;   // conv: device-ready parameters (B-dependent fields patched per call)
  int nrep = 0;
  int variant = 0;
  bf16* wt = nullptr;
  float* bias = nullptr;
  float* slope = nullptr;
  double flops = 0.0;  // algorithmic, per frame
  bool skip = false;   // computed inside the next op's kernel (stem fusion)
  int cin_span = 0;    // conv: 1 + the last input channel read that carries data (cin_map >= 0)
  int* tile_ctr = nullptr;  // stem op: the persistent kernel's two tile counters
  // 1x1 pair (conv_pair.hip): `pair_head` is computed inside the next op's kernel and `pair` runs that kernel
  // whenever ph_pair_enabled(); both keep their own parameters for the separate launches
  bool pair_head = false, pair = false;
  PhPairParams pp;
};

struct ph_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int batch = 0;
  std::vector<Buf> bufs;
  std::vector<Op> ops;
  int out_c = 0, out_h = 0, out_w = 0;
  float* net_out = nullptr;   // written by the network's fp32 epilogues
  size_t net_out_elems = 0;
  float* inj_out = nullptr;   // ph_set_net_out's own copy (never aliased by conv epilogues)
  size_t inj_elems = 0;
  float* cur_out = nullptr;   // what decode / export read: net_out after a forward, inj_out after an injection
  size_t cur_elems = 0;
  // injected net output geometry (ph_set_net_out) or the built network's
  int cur_c = 0, cur_h = 0, cur_w = 0;
  uint8_t* frames = nullptr;
  size_t frames_bytes = 0;
  // decode workspace
  PhV2Buffers v2 = {};
  PhV2Sizes v2sz = {};
  size_t v2_key = 0;
  V1Buffers v1 = {};
  V1Sizes v1sz = {};
  size_t v1_key = 0;
  int v1_walks[2] = {0, 0};   // of the last ph_decode_v1: irregular blobs walked by a wave / by one group
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev1[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> op_ev;
  int* overflow_host = nullptr;  // pinned: V2 overflow flag of the last ph_decode_v2_results
  // ph_pose_post scratch (grown on demand)
  void* post = nullptr;
  size_t post_bytes = 0;
};

static uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

extern "C" {

const char* ph_last_error(void) { return g_err; }

int ph_device_ok(int device) {
  hipDeviceProp_t prop;
  PH_CHECK(hipGetDeviceProperties(&prop, device));
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

static int create_resources(ph_ctx* c, void* stream) {
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    PH_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  for (int i = 0; i < 4; i++) PH_CHECK(hipEventCreate(&c->ev[i]));
  for (int i = 0; i < 4; i++) PH_CHECK(hipEventCreate(&c->ev1[i]));
  PH_CHECK(hipHostMalloc((void**)&c->overflow_host, sizeof(int), hipHostMallocDefault));
  *c->overflow_host = 0;
  return 0;
}

int ph_create(ph_ctx** out, int device, void* stream) {
  if (!out) {
    ph_set_error("ph_create: null out");
    return -1;
  }
  int n = 0;
  PH_CHECK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) {
    ph_set_error("ph_create: device %d out of range (%d visible)", device, n);
    return -1;
  }
  PH_CHECK(hipSetDevice(device));
  const int ok = ph_device_ok(device);
  if (ok < 0) return -1;
  if (!ok) {
    ph_set_error("ph_create: device %d is not gfx950 (libposehip is built for MI355X only)", device);
    return -1;
  }
  if (scratch_init(device)) return -1;
  ph_ctx* c = new ph_ctx();
  c->device = device;
  const int rc = create_resources(c, stream);
  if (rc) {  // ph_destroy releases whatever was created before the failure
    ph_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

static void free_net(ph_ctx* c) {
  for (auto& b : c->bufs)
    if (b.ptr) (void)hipFree(b.ptr);
  for (auto& o : c->ops) {
    if (o.wt) (void)hipFree(o.wt);
    if (o.bias) (void)hipFree(o.bias);
    if (o.slope) (void)hipFree(o.slope);
    if (o.tile_ctr) (void)hipFree(o.tile_ctr);
  }
  c->bufs.clear();
  c->ops.clear();
  for (auto e : c->op_ev) (void)hipEventDestroy(e);
  c->op_ev.clear();
}

static void free_v2(ph_ctx* c) {
  void* ptrs[] = {c->v2.nms_keys, c->v2.nms_cnt, c->v2.peaks,    c->v2.pair_scores, c->v2.conn_idx, c->v2.conn_score, c->v2.n_conn,
                  c->v2.n_people, c->v2.keypoints,   c->v2.scores,   c->v2.subsets,    c->v2.overflow,
                  c->v2.n_accept, c->v2.mrg_lists};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  c->v2 = PhV2Buffers{};
  c->v2_key = 0;
}

static void free_v1(ph_ctx* c) {
  void* ptrs[] = {c->v1.n_runs, c->v1.runs,    c->v1.kps,     c->v1.n_kps,
                  c->v1.pairs,  c->v1.n_pairs, c->v1.persons, c->v1.n_persons, c->v1.overflow, c->v1.ws_verts};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  c->v1 = V1Buffers{};
  c->v1_key = 0;
}

void ph_destroy(ph_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_net(c);
  free_v2(c);
  free_v1(c);
  for (int i = 0; i < 4; i++)
    if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
  if (c->net_out) (void)hipFree(c->net_out);
  if (c->inj_out) (void)hipFree(c->inj_out);
  if (c->frames) (void)hipFree(c->frames);
  if (c->post) (void)hipFree(c->post);
  if (c->overflow_host) (void)hipHostFree(c->overflow_host);
  for (int i = 0; i < 4; i++)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int ph_sync(ph_ctx* c) {
  PH_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

void* ph_host_alloc(int64_t bytes) {
  void* p = nullptr;
  if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
    ph_set_error("ph_host_alloc: %lld bytes of page-locked memory not available", (long long)bytes);
    return nullptr;
  }
  return p;
}

int ph_host_free(void* ptr) {
  if (ptr) PH_CHECK(hipHostFree(ptr));
  return 0;
}

static int ensure_net_out(ph_ctx* c, size_t elems) {
  if (elems > c->net_out_elems) {
    if (c->net_out) PH_CHECK(hipFree(c->net_out));
    c->net_out = nullptr;
    PH_CHECK(hipMalloc(&c->net_out, elems * sizeof(float)));
    c->net_out_elems = elems;
  }
  return 0;
}

static int net_build(ph_ctx* c, int batch, const ph_buf_desc* bufs, int n_bufs, const ph_op_desc* ops, int n_ops,
                     int out_channels, int out_h, int out_w);

int ph_net_build(ph_ctx* c, int batch, const ph_buf_desc* bufs, int n_bufs, const ph_op_desc* ops, int n_ops,
                 int out_channels, int out_h, int out_w) {
  const int rc = net_build(c, batch, bufs, n_bufs, ops, n_ops, out_channels, out_h, out_w);
  if (rc) {  // never leave a half-built plan behind
    (void)hipStreamSynchronize(c->stream);
    free_net(c);
    c->batch = 0;
  }
  return rc;
}

static int net_build(ph_ctx* c, int batch, const ph_buf_desc* bufs, int n_bufs, const ph_op_desc* ops, int n_ops,
                     int out_channels, int out_h, int out_w) {
  PH_CHECK(hipSetDevice(c->device));
  PH_CHECK(hipStreamSynchronize(c->stream));
  free_net(c);
  if (c->cur_out == c->net_out) {  // the old network's output may be reallocated below
    c->cur_out = nullptr;
    c->cur_elems = 0;
  }
  if (batch < 1 || n_bufs < 1 || n_ops < 1) {
    ph_set_error("ph_net_build: bad sizes (batch %d, %d buffers, %d ops)", batch, n_bufs, n_ops);
    return -1;
  }
  c->batch = batch;
  c->out_c = out_channels;
  c->out_h = out_h;
  c->out_w = out_w;
  c->bufs.resize(n_bufs);
  for (int i = 0; i < n_bufs; i++) {
    Buf& b = c->bufs[i];
    b.d = bufs[i];
    if (b.d.channels % 8 || b.d.channels < 8 || b.d.height < 1 || b.d.width < 1 || b.d.pad < 0) {
      ph_set_error("ph_net_build: buffer %d: channels %d must be a positive multiple of 8", i, b.d.channels);
      return -1;
    }
    b.elems = (size_t)batch * b.Hp() * b.Wp() * b.d.channels;
    if (b.elems >= 0xffffffffull) {
      ph_set_error("ph_net_build: buffer %d has %zu elements; 32-bit offsets need < 2^32", i, b.elems);
      return -1;
    }
    PH_CHECK(hipMalloc(&b.ptr, b.elems * sizeof(bf16)));
    PH_CHECK(hipMemsetAsync(b.ptr, 0, b.elems * sizeof(bf16), c->stream));  // zero border, zero pad channels
  }
  if (ensure_net_out(c, (size_t)batch * out_channels * out_h * out_w)) return -1;
  PH_CHECK(hipMemsetAsync(c->net_out, 0, c->net_out_elems * sizeof(float), c->stream));
  c->ops.resize(n_ops);
  for (int i = 0; i < n_ops; i++) {
    Op& o = c->ops[i];
    o.d = ops[i];
    const ph_op_desc& d = o.d;
    if (d.in_buf < 0 || d.in_buf >= n_bufs || d.n_dst < 0 || d.n_dst > PH_MAX_DST) {
      ph_set_error("ph_net_build: op %d: bad buffer ids", i);
      return -1;
    }
    const Buf& in = c->bufs[d.in_buf];
    for (int k = 0; k < d.n_dst; k++)
      if (d.dst_buf[k] < 0 || d.dst_buf[k] >= n_bufs) {
        ph_set_error("ph_net_build: op %d: bad dst buffer", i);
        return -1;
      }
    if (d.type == PH_OP_POOL) {
      if (d.n_dst != 1) {
        ph_set_error("ph_net_build: op %d: pool needs exactly one dst", i);
        return -1;
      }
      const Buf& ob = c->bufs[d.dst_buf[0]];
      // the pooled map may be one member of a wider (concat) buffer: an 8-aligned channel window of it
      if (d.dst_coff[0] < 0 || (d.dst_coff[0] & 7) || d.dst_coff[0] + in.d.channels > ob.d.channels ||
          ob.d.height != (in.d.height + 1) / 2 || ob.d.width != (in.d.width + 1) / 2) {
        ph_set_error("ph_net_build: op %d: pool geometry mismatch (%dx%dx%d -> window %d of %dx%dx%d)", i, in.d.height,
                     in.d.width, in.d.channels, d.dst_coff[0], ob.d.height, ob.d.width, ob.d.channels);
        return -1;
      }
      continue;
    }
    if (d.type != PH_OP_CONV) {
      ph_set_error("ph_net_build: op %d: unknown type %d", i, d.type);
      return -1;
    }
    const int pad = (d.kh - 1) / 2;
    if (d.kh != d.kw || d.kh % 2 == 0 || pad > in.d.pad) {
      ph_set_error("ph_net_build: op %d: kernel %dx%d needs odd square kernel and input border >= %d (has %d)", i,
                   d.kh, d.kw, pad, in.d.pad);
      return -1;
    }
    if (d.cin_read % 8 || d.in_coff % 8 || d.in_coff + d.cin_read > in.d.channels || d.cin_read < 8 ||
        !d.weight || !d.cin_map || d.cout < 1) {
      ph_set_error("ph_net_build: op %d: bad channel window (coff %d, read %d of %d)", i, d.in_coff, d.cin_read,
                   in.d.channels);
      return -1;
    }
    if (d.pool && (d.kh != 3 || (d.cin_read & 31) || ((in.d.height | in.d.width) & 1) || d.f32_coff >= 0 ||
                   d.n_dst < 1)) {
      ph_set_error("ph_net_build: op %d: fused pooling needs a 3x3 conv, cin_read %% 32 == 0, even H and W", i);
      return -1;
    }
    for (int k = 0; k < d.n_dst; k++) {
      const Buf& ob = c->bufs[d.dst_buf[k]];
      const int oh = d.pool ? in.d.height / 2 : in.d.height, ow = d.pool ? in.d.width / 2 : in.d.width;
      if (ob.d.height != oh || ob.d.width != ow || d.dst_coff[k] % 8 ||
          d.dst_coff[k] + d.cout > ob.d.channels) {
        ph_set_error("ph_net_build: op %d: dst %d geometry/channel window mismatch", i, k);
        return -1;
      }
    }
    if (d.f32_coff >= 0 && (in.d.height != out_h || in.d.width != out_w || d.f32_coff + d.cout > out_channels)) {
      ph_set_error("ph_net_build: op %d: fp32 output window mismatch", i);
      return -1;
    }
    // tile width along Cout
    int tn = d.tile_n;
    if (tn == 0) tn = d.cout <= 32 ? 32 : (d.cout <= 64 ? 64 : (d.cout % 128 == 0 ? 128 : (d.cout % 96 == 0 ? 96 : 128)));
    if (d.tile_n == 0) {
      // few pixels (single frames): narrower N tiles put more blocks on the chip; the K loop is latency-bound
      // per block, so the frame latency falls although every patch is then fetched by several blocks
      static const char* small_env = getenv("POSEHIP_SMALL_BATCH_TILE_N");
      const int small_tn = small_env ? atoi(small_env) : 32;  // narrowest allowed; 0 disables
      const long long m_tiles = ((long long)batch * in.d.height * in.d.width + 255) / 256;
      static const char* blocks_env = getenv("POSEHIP_SMALL_BATCH_BLOCKS");
      const int min_blocks = blocks_env ? atoi(blocks_env) : 128;
      while (small_tn && tn > small_tn && m_tiles * ((d.cout + tn - 1) / tn) < min_blocks)
        tn = (tn == 128 && d.cout % 64 == 0) ? 64 : 32;   // 128 -> 64 -> 32; 96 -> 32 (no half-empty tile)
    }
    if (tn != 32 && tn != 64 && tn != 96 && tn != 128) {
      ph_set_error("ph_net_build: op %d: tile_n %d not in {32,64,96,128}", i, tn);
      return -1;
    }
    o.nrep = tn / 32;
    const int cout_pad = (d.cout + 127) / 128 * 128;
    const int K = d.kh * d.kw * d.cin_read;
    const int Kpad = (K + 63) / 64 * 64;
    if ((unsigned long long)Kpad * d.cin_read >= 0xffffffffull) {
      ph_set_error("ph_net_build: op %d: K too large", i);
      return -1;
    }
    // weights -> bf16 [cout_pad][Kpad], k = (kh*KW + kw)*cin_read + c
    std::vector<uint16_t> w((size_t)cout_pad * Kpad, 0);
    for (int n = 0; n < d.cout; n++)
      for (int t = 0; t < d.kh * d.kw; t++)
        for (int cc = 0; cc < d.cin_read; cc++) {
          const int src = d.cin_map[cc];
          if (src < 0) continue;
          if (src >= d.cin_caffe) {
            ph_set_error("ph_net_build: op %d: cin_map[%d]=%d out of range %d", i, cc, src, d.cin_caffe);
            return -1;
          }
          w[(size_t)n * Kpad + t * d.cin_read + cc] =
              f32_to_bf16(d.weight[((size_t)n * d.cin_caffe + src) * d.kh * d.kw + t]);
        }
    std::vector<float> bias(cout_pad + 128, 0.f), slope(cout_pad + 128, 0.f);  // tiles stage 128 entries from n0
    if (d.bias) memcpy(bias.data(), d.bias, d.cout * sizeof(float));
    if (d.act == PH_ACT_PRELU) {
      if (!d.slope || (d.slope_count != 1 && d.slope_count != d.cout)) {
        ph_set_error("ph_net_build: op %d: PReLU needs 1 or cout slopes", i);
        return -1;
      }
      for (int n = 0; n < d.cout; n++) slope[n] = d.slope[d.slope_count == 1 ? 0 : n];
    }
    PH_CHECK(hipMalloc(&o.wt, w.size() * 2));
    PH_CHECK(hipMalloc(&o.bias, bias.size() * sizeof(float)));
    PH_CHECK(hipMalloc(&o.slope, slope.size() * sizeof(float)));
    PH_CHECK(hipMemcpy(o.wt, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    PH_CHECK(hipMemcpy(o.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    PH_CHECK(hipMemcpy(o.slope, slope.data(), slope.size() * sizeof(float), hipMemcpyHostToDevice));
    PhConvParams& p = o.cp;
    memset(&p, 0, sizeof(p));
    p.in = in.ptr;
    p.wt = o.wt;
    p.bias = o.bias;
    p.slope = o.slope;
    p.H = in.d.height;
    p.W = in.d.width;
    p.in_Hp = in.Hp();
    p.in_Wp = in.Wp();
    p.in_Cs = in.d.channels;
    p.in_org = in.org(-pad, d.in_coff);
    p.cin_read = d.cin_read;
    p.KW = d.kw;
    p.K = K;
    p.Kpad = Kpad;
    p.nk = Kpad / 64;
    p.magic_cin = ph_magic(d.cin_read);
    p.magic_kw = ph_magic(d.kw);
    p.cout = d.cout;
    p.cout_pad = cout_pad;
    {
      const char* ev = getenv("POSEHIP_CONV_DEBUG");
      const char* eo = getenv("POSEHIP_CONV_DEBUG_OP");
      p.debug = ev ? atoi(ev) : 0;
      if (eo && atoi(eo) != i) p.debug = 0;
    }
    p.act = d.act;
    p.n_dst = d.n_dst;
    for (int k = 0; k < d.n_dst; k++) {
      const Buf& ob = c->bufs[d.dst_buf[k]];
      p.dst[k].base = ob.ptr;
      p.dst[k].Hp = ob.Hp();
      p.dst[k].Wp = ob.Wp();
      p.dst[k].Cs = ob.d.channels;
      p.dst[k].org = ob.org(0, d.dst_coff[k]);
    }
    p.f32_out = d.f32_coff >= 0 ? c->net_out : nullptr;
    p.f32_C = out_channels;
    p.f32_coff = d.f32_coff >= 0 ? d.f32_coff : 0;
    p.B = batch;
    p.M = batch * in.d.height * in.d.width;
    p.in_pad = in.d.pad;
    p.in_coff = d.in_coff;
    p.NQ = batch * in.Hp() * in.Wp();
    {
      // kernel choice: POSEHIP_CONV_KERNEL=0 forces the im2col-gather kernel, 4/5 the 3x3 halo-patch kernel
      // (4 / 8 waves) wherever it applies
      const char* ev = getenv("POSEHIP_CONV_KERNEL");
      // auto: halo kernel with 8-wave blocks (64-row wave tiles, 2 blocks per CU = 4 waves per SIMD).  With the
      // branch-free K loop it needs no cross-step fragment prefetch and fits 128 registers at every tile width; it
      // measured 3-8 % faster than the 4-wave / 128-row-wave-tile form on the 46x82 layers and 1-2 % on the others
      const int want = ev ? atoi(ev) : 5;
      o.variant = want;
      p.pool = d.pool;
      if (d.pool && want != 4 && want != 5) o.variant = 4;  // only the halo kernel pools
      if ((want == 4 || want == 5 || want == 6) && d.kh == 1) o.variant = 6;  // 1x1 through the halo pipeline
      if ((o.variant == 4 || o.variant == 5 || o.variant == 6) && !ph_conv_halo_plan(p, d.kh, d.f32_coff >= 0)) {
        if (d.pool) {
          ph_set_error("ph_net_build: op %d: fused pooling not applicable", i);
          return -1;
        }
        o.variant = 0;
      }
    }
    int real_cin = 0;
    for (int cc = 0; cc < d.cin_read; cc++) {
      real_cin += d.cin_map[cc] >= 0;
      if (d.cin_map[cc] >= 0) o.cin_span = cc + 1;
    }
    o.flops = 2.0 * d.kh * d.kw * real_cin * d.cout * in.d.height * in.d.width;
    // descriptor pointers are host memory owned by the caller: do not keep them
    o.d.weight = nullptr;
    o.d.bias = nullptr;
    o.d.slope = nullptr;
    o.d.cin_map = nullptr;
  }
  // Stem fusion: a 3x3 conv over the 8-channel frame buffer (conv1_1) whose 64-channel output is read only
  // by the next 3x3 conv (conv1_2) is computed inside that conv's kernel, straight into its LDS patch; the
  // full-resolution 64-channel blob (1 GB at 32 x 368x656) is then never written or read.
  if (!getenv("POSEHIP_NO_STEM_FUSION")) {
    for (int i = 0; i + 1 < n_ops; i++) {
      Op& a = c->ops[i];
      Op& b = c->ops[i + 1];
      if (a.d.type != PH_OP_CONV || b.d.type != PH_OP_CONV) continue;
      const Buf& fb = c->bufs[a.d.in_buf];
      if (a.d.kh != 3 || a.d.cin_read != 8 || a.d.in_coff != 0 || fb.d.channels != 8 || a.d.cout != 64 ||
          a.d.act > PH_ACT_RELU || a.d.n_dst != 1 || a.d.f32_coff >= 0 || a.d.dst_coff[0] != 0 || a.d.pool)
        continue;
      if (b.d.kh != 3 || b.d.in_buf != a.d.dst_buf[0] || b.d.in_coff != 0 || b.d.cin_read != 64 || b.d.cout > 64 ||
          (b.variant != 4 && b.variant != 5) || b.cp.debug)
        continue;
      bool shared = false;
      for (int j = 0; j < n_ops; j++)
        if (j != i + 1 && c->ops[j].d.in_buf == a.d.dst_buf[0]) shared = true;
      if (shared) continue;
      PhConvParams q = b.cp;
      q.stem = 1;
      q.wt0 = a.wt;
      q.bias0 = a.bias;
      q.Kpad0 = a.cp.Kpad;
      q.act0 = a.d.act;
      q.cin0_span = a.cin_span;
      q.in = fb.ptr;
      q.in_Hp = fb.Hp();
      q.in_Wp = fb.Wp();
      q.in_Cs = 8;
      q.in_pad = fb.d.pad;
      q.in_coff = 0;
      q.NQ = batch * fb.Hp() * fb.Wp();
      if (q.Kpad0 < 96 || !ph_conv_halo_plan(q, 3, false)) continue;
      PH_CHECK(hipMalloc(&b.tile_ctr, 2 * sizeof(int)));
      PH_CHECK(hipMemset(b.tile_ctr, 0, 2 * sizeof(int)));
      q.tile_ctr = b.tile_ctr;
      q.tile_par = 0;
      b.cp = q;
      b.variant = 5;
      b.nrep = 2;  // the stem instantiation is 64 channels wide
      a.skip = true;
    }
  }
  // 1x1 pair: a wide 1x1 conv (Mconv6: -> 256 / 512 channels) whose output is read only by the next, narrow 1x1
  // conv (Mconv7: -> <= 64 channels) runs inside that conv's kernel, slab by slab through LDS; the wide map
  // (123 MB at 32 x 46x82 x 512) is then neither written nor read
  for (int i = 0; i + 1 < n_ops; i++) {
    Op& a = c->ops[i];
    Op& b = c->ops[i + 1];
    if (a.d.type != PH_OP_CONV || b.d.type != PH_OP_CONV || a.skip || b.skip) continue;
    if (a.d.kh != 1 || a.d.kw != 1 || b.d.kh != 1 || b.d.kw != 1 || a.variant != 6 || b.variant != 6) continue;
    if (a.cp.debug || b.cp.debug || a.d.pool || b.d.pool) continue;
    if (a.d.cout % 128 || a.d.cout > 512 || a.d.n_dst != 1 || a.d.f32_coff >= 0 || a.d.dst_coff[0] != 0) continue;
    if (a.d.cin_read % 32 || a.d.cin_read < 96) continue;
    const Buf& mid = c->bufs[a.d.dst_buf[0]];
    if (b.d.in_buf != a.d.dst_buf[0] || b.d.in_coff != 0 || b.d.cin_read != a.d.cout || mid.d.channels != a.d.cout ||
        b.d.cout > 64 || b.cp.Kpad != a.d.cout)
      continue;
    bool shared = false;
    for (int j = 0; j < n_ops; j++)
      if (j != i + 1 && c->ops[j].d.in_buf == a.d.dst_buf[0]) shared = true;
    if (shared) continue;
    PhPairParams& q = b.pp;
    memset(&q, 0, sizeof(q));
    q.a = a.cp;
    q.wt2 = b.wt;
    q.bias2 = b.bias;
    q.slope2 = b.slope;
    q.Kpad2 = b.cp.Kpad;
    q.cout2 = b.d.cout;
    q.act2 = b.d.act;
    q.n_dst2 = b.cp.n_dst;
    for (int k = 0; k < b.cp.n_dst; k++) q.dst2[k] = b.cp.dst[k];
    q.f32_out2 = b.cp.f32_out;
    q.f32_C2 = b.cp.f32_C;
    q.f32_coff2 = b.cp.f32_coff;
    a.pair_head = true;
    b.pair = true;
  }
  c->op_ev.resize(n_ops + 1);
  for (auto& e : c->op_ev) PH_CHECK(hipEventCreate(&e));
  c->cur_out = c->net_out;
  c->cur_elems = c->net_out_elems;
  c->cur_c = out_channels;
  c->cur_h = out_h;
  c->cur_w = out_w;
  PH_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

int ph_preprocess(ph_ctx* c, const uint8_t* frames, int on_device, int B, int H0, int W0, int mode) {
  PH_CHECK(hipSetDevice(c->device));
  if (c->bufs.empty() || B < 1 || B > c->batch || H0 < 1 || W0 < 1 || (mode != 0 && mode != 1)) {
    ph_set_error("ph_preprocess: no network built, or bad args (B %d of %d, %dx%d, mode %d)", B, c->batch, H0, W0,
                 mode);
    return -1;
  }
  const uint8_t* src = frames;
  const size_t bytes = (size_t)B * H0 * W0 * 3;
  if (!on_device) {
    if (bytes > c->frames_bytes) {
      if (c->frames) PH_CHECK(hipFree(c->frames));
      c->frames = nullptr;
      PH_CHECK(hipMalloc(&c->frames, bytes));
      c->frames_bytes = bytes;
    }
    PH_CHECK(hipMemcpyAsync(c->frames, frames, bytes, hipMemcpyHostToDevice, c->stream));
    src = c->frames;
  }
  const Buf& b0 = c->bufs[0];
  return ph_launch_preprocess(src, B, H0, W0, b0.d.height, b0.d.width, mode, b0.ptr, b0.Hp(), b0.Wp(),
                              b0.d.channels, b0.org(0, 0), c->stream);
}

static int run_ops(ph_ctx* c, int B, bool timed) {
  if (c->ops.empty() || B < 1 || B > c->batch) {
    ph_set_error("ph_forward: no network built or batch %d outside [1,%d]", B, c->batch);
    return -1;
  }
  PH_CHECK(hipSetDevice(c->device));
  const bool pairs = ph_pair_enabled();
  for (size_t i = 0; i < c->ops.size(); i++) {
    Op& o = c->ops[i];
    if (timed) PH_CHECK(hipEventRecord(c->op_ev[i], c->stream));
    const Buf& in = c->bufs[o.d.in_buf];
    if (o.skip || (pairs && o.pair_head)) continue;
    if (pairs && o.pair) {
      o.pp.a.B = B;
      o.pp.a.M = B * o.pp.a.H * o.pp.a.W;
      if (ph_launch_pair(o.pp, c->stream)) return -1;
      continue;
    }
    if (o.d.type == PH_OP_POOL) {
      const Buf& ob = c->bufs[o.d.dst_buf[0]];
      if (ph_launch_maxpool(in.ptr, ob.ptr, B, in.d.height, in.d.width, in.d.channels, in.Hp(), in.Wp(),
                            in.d.channels, in.org(0, 0), ob.d.height, ob.d.width, ob.Hp(), ob.Wp(), ob.d.channels,
                            ob.org(0, o.d.dst_coff[0]), c->stream))
        return -1;
    } else {
      o.cp.B = B;
      o.cp.M = B * o.cp.H * o.cp.W;
      if (o.variant >= 4) ph_conv_halo_plan(o.cp, o.d.kh, o.d.f32_coff >= 0);  // tile count follows B
      if (ph_launch_conv(o.cp, o.nrep, o.variant, c->stream)) return -1;
      o.cp.tile_par ^= 1;
    }
  }
  if (timed) PH_CHECK(hipEventRecord(c->op_ev[c->ops.size()], c->stream));
  c->cur_out = c->net_out;
  c->cur_elems = c->net_out_elems;
  c->cur_c = c->out_c;
  c->cur_h = c->out_h;
  c->cur_w = c->out_w;
  return 0;
}

int ph_forward(ph_ctx* c, int B) { return run_ops(c, B, false); }

int ph_forward_times(ph_ctx* c, float* ms);

int ph_forward_timed(ph_ctx* c, int B, float* ms) {
  if (run_ops(c, B, true)) return -1;
  return ms ? ph_forward_times(c, ms) : 0;
}

int ph_forward_times(ph_ctx* c, float* ms) {
  if (c->ops.empty() || !ms) {
    ph_set_error("ph_forward_times: no network built");
    return -1;
  }
  PH_CHECK(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->ops.size(); i++) PH_CHECK(hipEventElapsedTime(&ms[i], c->op_ev[i], c->op_ev[i + 1]));
  return 0;
}

double ph_op_flops(ph_ctx* c, int op) {
  if (op < 0 || op >= (int)c->ops.size()) return -1.0;
  return c->ops[op].flops;
}

int ph_op_kernel(ph_ctx* c, int op, char* name, int cap) {
  if (op < 0 || op >= (int)c->ops.size() || !name || cap < 1) {
    ph_set_error("ph_op_kernel: bad op %d", op);
    return -1;
  }
  const Op& o = c->ops[op];
  if (o.skip || (o.pair_head && ph_pair_enabled())) snprintf(name, cap, "(fused into the next op)");
  else if (o.pair && ph_pair_enabled()) snprintf(name, cap, "conv1x1_pair_kernel(PhPairParams)");
  else if (o.d.type == PH_OP_POOL) snprintf(name, cap, "maxpool2x2_kernel");
  else if (o.cp.stem && ph_stem512_applies(o.cp)) snprintf(name, cap, "conv_stem512_kernel(PhConvParams, int, int, int)");
  else if (o.cp.stem) snprintf(name, cap, "conv3x3_halo_kernel<2, 4, false, true, 9>");
  else if (o.variant == 6) snprintf(name, cap, "conv3x3_halo_kernel<%d, %d, false, false, 1>", o.nrep, ph_conv_1x1_mw());
  else if (o.variant == 4 || o.variant == 5)
    snprintf(name, cap, "conv3x3_halo_kernel<%d, %d, %s, false, 9>", o.nrep, o.variant == 5 ? 4 : 2,
             o.cp.debug ? "true" : "false");
  else snprintf(name, cap, "conv_igemm_kernel<%d>", o.nrep);
  return 0;
}

int ph_get_net_out(ph_ctx* c, int B, float* host_out) {
  PH_CHECK(hipSetDevice(c->device));
  const size_t n = (size_t)B * c->cur_c * c->cur_h * c->cur_w;
  if (!c->cur_out || n > c->cur_elems) {
    ph_set_error("ph_get_net_out: no net output for batch %d", B);
    return -1;
  }
  PH_CHECK(hipMemcpyAsync(host_out, c->cur_out, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  PH_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

float* ph_net_out_dev(ph_ctx* c) { return c->cur_out; }

int ph_set_net_out(ph_ctx* c, const float* src, int on_device, int B, int C, int Hs, int Ws) {
  PH_CHECK(hipSetDevice(c->device));
  if (B < 1 || C < 1 || Hs < 1 || Ws < 1 || !src) {
    ph_set_error("ph_set_net_out: bad shape [%d,%d,%d,%d]", B, C, Hs, Ws);
    return -1;
  }
  const size_t n = (size_t)B * C * Hs * Ws;
  if (n > c->inj_elems) {
    PH_CHECK(hipStreamSynchronize(c->stream));
    if (c->inj_out) PH_CHECK(hipFree(c->inj_out));
    c->inj_out = nullptr;
    c->inj_elems = 0;
    PH_CHECK(hipMalloc(&c->inj_out, n * sizeof(float)));
    c->inj_elems = n;
  }
  PH_CHECK(hipMemcpyAsync(c->inj_out, src, n * sizeof(float),
                          on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  if (!on_device) PH_CHECK(hipStreamSynchronize(c->stream));
  c->cur_out = c->inj_out;
  c->cur_elems = c->inj_elems;
  c->cur_c = C;
  c->cur_h = Hs;
  c->cur_w = Ws;
  return 0;
}

int ph_get_buffer(ph_ctx* c, int buf, uint16_t* host_out, int64_t n_elems) {
  PH_CHECK(hipSetDevice(c->device));
  if (buf < 0 || buf >= (int)c->bufs.size() || n_elems < 0 || (size_t)n_elems > c->bufs[buf].elems) {
    ph_set_error("ph_get_buffer: bad buffer %d / size", buf);
    return -1;
  }
  PH_CHECK(hipMemcpyAsync(host_out, c->bufs[buf].ptr, (size_t)n_elems * 2, hipMemcpyDeviceToHost, c->stream));
  PH_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

int ph_upsample_maps(ph_ctx* c, int frame, int c0, int nc, int Ht, int Wt, int n_heat, int scale_mode, float* host_out) {
  PH_CHECK(hipSetDevice(c->device));
  if (scale_mode < 0 || scale_mode > 3) {
    ph_set_error("ph_upsample_maps: heatmaps_scale %d (0..3)", scale_mode);
    return -1;
  }
  if (!c->cur_out || frame < 0 || c0 < 0 || nc < 1 || c0 + nc > c->cur_c || Ht < 1 || Wt < 1 ||
      (size_t)(frame + 1) * c->cur_c * c->cur_h * c->cur_w > c->cur_elems) {
    ph_set_error("ph_upsample_maps: bad frame/channel window");
    return -1;
  }
  float* tmp = nullptr;
  const size_t n = (size_t)nc * Ht * Wt;
  PH_CHECK(hipMalloc(&tmp, n * sizeof(float)));
  const float* planes = c->cur_out + ((size_t)frame * c->cur_c + c0) * c->cur_h * c->cur_w;
  int rc = ph_launch_upsample_maps(planes, c->cur_h, c->cur_w, Ht, Wt, nc, n_heat - c0, scale_mode, tmp, c->stream);
  if (!rc && hipMemcpyAsync(host_out, tmp, n * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = -1;
  if (hipStreamSynchronize(c->stream) != hipSuccess) rc = -1;
  (void)hipFree(tmp);
  if (rc) ph_set_error("ph_upsample_maps: launch or copy failed");
  return rc;
}

int ph_argmax_maps(ph_ctx* c, int B, float* vals_host, int32_t* locs_host) {
  PH_CHECK(hipSetDevice(c->device));
  const int n_maps = B * c->cur_c, n = c->cur_h * c->cur_w;
  if (!c->cur_out || B < 1 || (size_t)n_maps * n > c->cur_elems) {
    ph_set_error("ph_argmax_maps: no net output for batch %d", B);
    return -1;
  }
  float* dv = nullptr;
  int* dl = nullptr;
  PH_CHECK(hipMalloc(&dv, n_maps * sizeof(float)));
  PH_CHECK(hipMalloc(&dl, n_maps * sizeof(int)));
  int rc = ph_launch_argmax_maps(c->cur_out, n_maps, n, dv, dl, c->stream);
  if (!rc && hipMemcpyAsync(vals_host, dv, n_maps * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = -1;
  if (!rc && hipMemcpyAsync(locs_host, dl, n_maps * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = -1;
  if (hipStreamSynchronize(c->stream) != hipSuccess) rc = -1;
  (void)hipFree(dv);
  (void)hipFree(dl);
  if (rc) ph_set_error("ph_argmax_maps: launch or copy failed");
  return rc;
}

// ---- V2 decode ---------------------------------------------------------------
static int v2_model_to_dev(const ph_v2_model* m, int C, int Ht, int Wt, PhV2Dev* dev) {
  if (m->n_pairs > 64 || m->n_pairs < 1 || m->n_parts < 1 || m->max_peaks < 1 || m->max_peaks > 1023) {
    ph_set_error("ph_decode_v2: bad model constants");
    return -1;
  }
  for (int k = 0; k < 2 * m->n_pairs; k++)
    if (m->pairs[k] < 0 || m->pairs[k] >= m->n_parts || m->mapidx[k] < 0 || m->n_heat + m->mapidx[k] >= C) {
      ph_set_error("ph_decode_v2: pair/mapidx table entry %d out of range", k);
      return -1;
    }
  memset(dev, 0, sizeof(*dev));
  dev->n_parts = m->n_parts;
  dev->n_heat = m->n_heat;
  dev->n_pairs = m->n_pairs;
  dev->max_peaks = m->max_peaks;
  dev->ear_pair0 = m->ear_pair0;
  dev->ear_pair1 = m->ear_pair1;
  dev->min_subset_cnt = m->min_subset_cnt;
  dev->nms_threshold = m->nms_threshold;
  dev->inter_threshold = m->inter_threshold;
  dev->inter_min_above = m->inter_min_above_threshold;
  dev->min_subset_score = m->min_subset_score;
  dev->close_score = (float)((double)m->nms_threshold + 1e-6);
  dev->connector = m->connector;
  // bodyPartConnectorBase.cpp getScoreAB: std::sqrt(area) / 150 in double; .cu process(): sqrtf(w * h) / 150 in float
  dev->close_thr = m->connector == 1 ? (double)(sqrtf((float)(Wt * Ht)) / 150.f) : sqrt((double)(Wt * Ht)) / 150.0;
  memcpy(dev->pairs, m->pairs, sizeof(int) * 2 * m->n_pairs);
  memcpy(dev->mapidx, m->mapidx, sizeof(int) * 2 * m->n_pairs);
  return 0;
}

static int v2_prepare(ph_ctx* c, const ph_v2_model* m, int B, int Ht, int Wt, int max_people, PhV2Dev* dev) {
  PH_CHECK(hipSetDevice(c->device));
  if (!m || B < 1 || max_people < 1 || !c->cur_out) {
    ph_set_error("ph_decode_v2: bad args or no net output");
    return -1;
  }
  if ((size_t)B * c->cur_c * c->cur_h * c->cur_w > c->cur_elems) {
    ph_set_error("ph_decode_v2: batch %d exceeds the resident net output", B);
    return -1;
  }
  if (v2_model_to_dev(m, c->cur_c, Ht, Wt, dev)) return -1;
  const size_t key = ((size_t)B << 40) ^ ((size_t)max_people << 24) ^ ((size_t)m->n_parts << 16) ^
                     ((size_t)m->n_pairs << 8) ^ (size_t)m->max_peaks ^ ((size_t)m->connector << 62);
  if (key != c->v2_key) {
    PH_CHECK(hipStreamSynchronize(c->stream));
    free_v2(c);
    ph_v2_workspace_sizes(*dev, B, max_people, &c->v2sz);
    PH_CHECK(hipMalloc(&c->v2.nms_keys, c->v2sz.nms_keys));
    PH_CHECK(hipMalloc(&c->v2.nms_cnt, c->v2sz.nms_cnt));
    PH_CHECK(hipMalloc(&c->v2.peaks, c->v2sz.peaks));
    PH_CHECK(hipMalloc(&c->v2.pair_scores, c->v2sz.pair_scores));
    PH_CHECK(hipMalloc(&c->v2.conn_idx, c->v2sz.conn_idx));
    PH_CHECK(hipMalloc(&c->v2.conn_score, c->v2sz.conn_score));
    PH_CHECK(hipMalloc(&c->v2.n_conn, c->v2sz.n_conn));
    PH_CHECK(hipMalloc(&c->v2.n_people, c->v2sz.n_people));
    PH_CHECK(hipMalloc(&c->v2.keypoints, c->v2sz.keypoints));
    PH_CHECK(hipMalloc(&c->v2.scores, c->v2sz.scores));
    PH_CHECK(hipMalloc(&c->v2.subsets, c->v2sz.subsets));
    PH_CHECK(hipMalloc(&c->v2.overflow, sizeof(int)));
    PH_CHECK(hipMalloc(&c->v2.n_accept, c->v2sz.n_accept));
    PH_CHECK(hipMalloc(&c->v2.mrg_lists, c->v2sz.mrg_lists));
    c->v2_key = key;
  }
  return 0;
}

int ph_decode_v2_dev(ph_ctx* c, const ph_v2_model* m, int B, int Ht, int Wt, float scale, int max_people,
                     int32_t** n_people_dev, float** keypoints_dev, float** scores_dev) {
  PhV2Dev dev;
  if (v2_prepare(c, m, B, Ht, Wt, max_people, &dev)) return -1;
  if (ph_launch_decode_v2(c->cur_out, c->cur_c, c->cur_h, c->cur_w, Ht, Wt, dev, B, scale, max_people, c->v2,
                          c->stream, c->ev))
    return -1;
  if (n_people_dev) *n_people_dev = c->v2.n_people;
  if (keypoints_dev) *keypoints_dev = c->v2.keypoints;
  if (scores_dev) *scores_dev = c->v2.scores;
  return 0;
}

int ph_decode_v2(ph_ctx* c, const ph_v2_model* m, int B, int Ht, int Wt, float scale, int max_people,
                 int32_t* n_people, float* keypoints, float* scores, float* peaks, int32_t* subsets) {
  if (ph_decode_v2_dev(c, m, B, Ht, Wt, scale, max_people, nullptr, nullptr, nullptr)) return -1;
  hipStream_t s = c->stream;
  if (n_people) PH_CHECK(hipMemcpyAsync(n_people, c->v2.n_people, c->v2sz.n_people, hipMemcpyDeviceToHost, s));
  if (keypoints) PH_CHECK(hipMemcpyAsync(keypoints, c->v2.keypoints, c->v2sz.keypoints, hipMemcpyDeviceToHost, s));
  if (scores) PH_CHECK(hipMemcpyAsync(scores, c->v2.scores, c->v2sz.scores, hipMemcpyDeviceToHost, s));
  if (peaks) PH_CHECK(hipMemcpyAsync(peaks, c->v2.peaks, c->v2sz.peaks, hipMemcpyDeviceToHost, s));
  if (subsets) PH_CHECK(hipMemcpyAsync(subsets, c->v2.subsets, c->v2sz.subsets, hipMemcpyDeviceToHost, s));
  int overflow = 0;
  PH_CHECK(hipMemcpyAsync(&overflow, c->v2.overflow, sizeof(int), hipMemcpyDeviceToHost, s));
  PH_CHECK(hipStreamSynchronize(s));
  if (overflow) {
    ph_set_error("ph_decode_v2: person-candidate table overflowed (more than %d partial skeletons in a frame)",
                 PH_V2_ROW_CAP);
    return -2;
  }
  return 0;
}

int ph_decode_v2_results(ph_ctx* c, int32_t* n_people, float* keypoints, float* scores, int dst_on_device) {
  PH_CHECK(hipSetDevice(c->device));
  if (!c->v2.n_people) {
    ph_set_error("ph_decode_v2_results: no decode has run");
    return -1;
  }
  const hipMemcpyKind k = dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  hipStream_t s = c->stream;
  if (n_people) PH_CHECK(hipMemcpyAsync(n_people, c->v2.n_people, c->v2sz.n_people, k, s));
  if (keypoints) PH_CHECK(hipMemcpyAsync(keypoints, c->v2.keypoints, c->v2sz.keypoints, k, s));
  if (scores) PH_CHECK(hipMemcpyAsync(scores, c->v2.scores, c->v2sz.scores, k, s));
  // the overflow flag travels with the results; ph_decode_v2_status() reads it once the stream has drained
  PH_CHECK(hipMemcpyAsync(c->overflow_host, c->v2.overflow, sizeof(int), hipMemcpyDeviceToHost, s));
  return 0;
}

int ph_decode_v2_status(ph_ctx* c) {
  PH_CHECK(hipSetDevice(c->device));
  if (!c->v2.n_people || !c->overflow_host) {
    ph_set_error("ph_decode_v2_status: no decode has run");
    return -1;
  }
  PH_CHECK(hipStreamSynchronize(c->stream));
  if (*c->overflow_host) {
    ph_set_error("ph_decode_v2: person-candidate table overflowed (more than %d partial skeletons in a frame)",
                 PH_V2_ROW_CAP);
    return -2;
  }
  return 0;
}

// The entry point that allocates nothing: same kernels as ph_decode_v2_dev, workspace carved out of this GPU's
// copy of `scratch` (through g_scratch_ptr, never the symbol itself), inputs and outputs are the caller's.
int ph_decode_v2_noalloc(ph_ctx* c, const ph_v2_model* m, const float* net_out_dev, int B, int C, int Hs, int Ws,
                         int Ht, int Wt, float scale, int max_people, int32_t* n_people_dev, float* keypoints_dev,
                         float* scores_dev, int32_t* overflow_dev) {
  PH_CHECK(hipSetDevice(c->device));
  if (!m || !net_out_dev || !n_people_dev || !keypoints_dev || !scores_dev || B < 1 || C < 1 || Hs < 1 || Ws < 1 ||
      max_people < 1) {
    ph_set_error("ph_decode_v2_noalloc: bad arguments");
    return -1;
  }
  unsigned char* base = (unsigned char*)g_scratch_ptr[c->device];
  if (!base) {
    ph_set_error("ph_decode_v2_noalloc: no scratch address for device %d", c->device);
    return -1;
  }
  PhV2Dev dev;
  if (v2_model_to_dev(m, C, Ht, Wt, &dev)) return -1;
  PhV2Sizes sz;
  ph_v2_workspace_sizes(dev, B, max_people, &sz);
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    unsigned char* p = base + off;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  };
  PhV2Buffers w = {};
  w.overflow = (int*)carve(sizeof(int));
  w.nms_keys = (int*)carve(sz.nms_keys);
  w.nms_cnt = (int*)carve(sz.nms_cnt);
  w.peaks = (float*)carve(sz.peaks);
  w.pair_scores = (float*)carve(sz.pair_scores);
  w.conn_idx = (int*)carve(sz.conn_idx);
  w.conn_score = (float*)carve(sz.conn_score);
  w.n_conn = (int*)carve(sz.n_conn);
  w.n_accept = (int*)carve(sz.n_accept);
  w.mrg_lists = carve(sz.mrg_lists);
  w.subsets = nullptr;  // optional output of the assembly kernel, not offered here
  if (B > PH_NOALLOC_MAX_FRAMES || off > PH_SCRATCH_BYTES) {
    ph_set_error("ph_decode_v2_noalloc: %d frames need %zu B of workspace; the static scratch holds %u B and a call "
                 "at most %d frames", B, off, PH_SCRATCH_BYTES, PH_NOALLOC_MAX_FRAMES);
    return -3;
  }
  w.n_people = n_people_dev;
  w.keypoints = keypoints_dev;
  w.scores = scores_dev;
  // ph_launch_decode_v2 clears what must start at zero (peak table, overflow flag); every other part of the
  // workspace is fully rewritten before it is read, so the previous call's contents do not matter
  if (ph_launch_decode_v2(net_out_dev, C, Hs, Ws, Ht, Wt, dev, B, scale, max_people, w, c->stream, nullptr)) return -1;
  if (overflow_dev)
    PH_CHECK(hipMemcpyAsync(overflow_dev, w.overflow, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int ph_pose_post(ph_ctx* c, int B, int max_people, int n_parts, const int32_t* torso_ids, int n_torso, float* bbox,
                 const uint16_t* depth, int depth_on_device, int Hd, int Wd, double fx, double fy, double cx, double cy,
                 double depth_scale, int patch_offset, double* kp3d) {
  PH_CHECK(hipSetDevice(c->device));
  if (!c->v2.n_people || B < 1 || max_people < 1 || n_parts < 1 ||
      (size_t)B * max_people * n_parts * 3 * sizeof(float) > c->v2sz.keypoints) {
    ph_set_error("ph_pose_post: no V2 decode of %d x %d x %d keypoints has run", B, max_people, n_parts);
    return -1;
  }
  if (bbox && (!torso_ids || n_torso < 1 || n_torso > 64)) {
    ph_set_error("ph_pose_post: bbox needs 1..64 torso ids");
    return -1;
  }
  for (int k = 0; bbox && k < n_torso; k++)
    if (torso_ids[k] < 0 || torso_ids[k] >= n_parts) {
      ph_set_error("ph_pose_post: torso id %d out of range", torso_ids[k]);
      return -1;
    }
  if (kp3d && (!depth || Hd < 1 || Wd < 1 || patch_offset < 0 || fx == 0.0 || fy == 0.0)) {
    ph_set_error("ph_pose_post: the 3-D lift needs a depth image and non-zero focal lengths");
    return -1;
  }
  const size_t n_bb = (size_t)B * max_people * 4 * sizeof(float);
  const size_t n_3d = (size_t)B * max_people * n_parts * 3 * sizeof(double);
  const size_t n_dp = (kp3d && !depth_on_device) ? (size_t)B * Hd * Wd * sizeof(uint16_t) : 0;
  const size_t off_bb = 256, off_3d = off_bb + ((n_bb + 255) & ~(size_t)255), off_dp = off_3d + ((n_3d + 255) & ~(size_t)255);
  const size_t need = off_dp + n_dp;
  if (need > c->post_bytes) {
    if (c->post) PH_CHECK(hipFree(c->post));
    c->post = nullptr;
    c->post_bytes = 0;
    PH_CHECK(hipMalloc(&c->post, need));
    c->post_bytes = need;
  }
  char* base = (char*)c->post;
  hipStream_t s = c->stream;
  if (bbox) {
    PH_CHECK(hipMemcpyAsync(base, torso_ids, n_torso * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (ph_launch_torso_bbox(c->v2.keypoints, c->v2.n_people, B, max_people, n_parts, (const int*)base, n_torso,
                             (float*)(base + off_bb), s))
      return -1;
    PH_CHECK(hipMemcpyAsync(bbox, base + off_bb, n_bb, hipMemcpyDeviceToHost, s));
  }
  if (kp3d) {
    const uint16_t* dd = depth;
    if (!depth_on_device) {
      PH_CHECK(hipMemcpyAsync(base + off_dp, depth, n_dp, hipMemcpyHostToDevice, s));
      dd = (const uint16_t*)(base + off_dp);
    }
    if (ph_launch_lift3d(c->v2.keypoints, c->v2.n_people, B, max_people, n_parts, dd, Hd, Wd, fx, fy, cx, cy,
                         depth_scale, patch_offset, (double*)(base + off_3d), s))
      return -1;
    PH_CHECK(hipMemcpyAsync(kp3d, base + off_3d, n_3d, hipMemcpyDeviceToHost, s));
  }
  PH_CHECK(hipStreamSynchronize(s));
  return 0;
}

int ph_pose_features(ph_ctx* c, int B, int max_people, int n_parts, const int32_t* torso_ids, int n_torso,
                     const int32_t* windows, int n_win, int n_heat, int scale_mode, int Ht, int Wt, int img_w, int img_h,
                     int fx, int fy, int capacity, float* feat, int32_t* status, int32_t* n_rows) {
  PH_CHECK(hipSetDevice(c->device));
  if (!c->v2.n_people || !c->cur_out || B < 1 || max_people < 1 || n_parts < 1 ||
      (size_t)B * max_people * n_parts * 3 * sizeof(float) > c->v2sz.keypoints ||
      (size_t)B * c->cur_c * c->cur_h * c->cur_w > c->cur_elems) {
    ph_set_error("ph_pose_features: no V2 decode of %d x %d x %d keypoints / net output has run", B, max_people, n_parts);
    return -1;
  }
  if (!torso_ids || n_torso < 1 || n_torso > 64 || !windows || n_win < 1 || n_win > 4 || !feat || !status || !n_rows ||
      capacity < 1 || fx < 1 || fy < 1 || Ht < 1 || Wt < 1 || img_w < 1 || img_h < 1 || scale_mode < 0 || scale_mode > 3) {
    ph_set_error("ph_pose_features: bad arguments");
    return -1;
  }
  for (int k = 0; k < n_torso; k++)
    if (torso_ids[k] < 0 || torso_ids[k] >= n_parts) {
      ph_set_error("ph_pose_features: torso id %d out of range", torso_ids[k]);
      return -1;
    }
  int nch = 0;
  for (int k = 0; k < n_win; k++) {
    if (windows[2 * k] < 0 || windows[2 * k + 1] < 1 || windows[2 * k] + windows[2 * k + 1] > c->cur_c) {
      ph_set_error("ph_pose_features: channel window %d out of range", k);
      return -1;
    }
    nch += windows[2 * k + 1];
  }
  const size_t row = (size_t)fy * fx * nch * sizeof(float);
  const size_t n_bb = (size_t)B * max_people * 4 * sizeof(float);
  const size_t n_st = (size_t)capacity * sizeof(int32_t);
  const size_t off_bb = 512, off_st = off_bb + ((n_bb + 255) & ~(size_t)255), off_ft = off_st + ((n_st + 255) & ~(size_t)255);
  const size_t need = off_ft + (size_t)capacity * row;
  hipStream_t s = c->stream;
  if (need > c->post_bytes) {
    PH_CHECK(hipStreamSynchronize(s));
    if (c->post) PH_CHECK(hipFree(c->post));
    c->post = nullptr;
    c->post_bytes = 0;
    PH_CHECK(hipMalloc(&c->post, need));
    c->post_bytes = need;
  }
  char* base = (char*)c->post;  // [0,256) torso ids, [256,260) row total
  PH_CHECK(hipMemcpyAsync(base, torso_ids, n_torso * sizeof(int32_t), hipMemcpyHostToDevice, s));
  PH_CHECK(hipMemsetAsync(base + off_st, 0, n_st, s));
  if (ph_launch_torso_bbox(c->v2.keypoints, c->v2.n_people, B, max_people, n_parts, (const int*)base, n_torso,
                           (float*)(base + off_bb), s))
    return -1;
  if (ph_launch_pose_features(c->cur_out, c->cur_c, c->cur_h, c->cur_w, (const float*)(base + off_bb), c->v2.n_people, B,
                              max_people, windows, n_win, n_heat, scale_mode, Ht, Wt, img_w, img_h, fx, fy, capacity,
                              (float*)(base + off_ft), (int*)(base + off_st), (int*)(base + 256), s))
    return -1;
  int32_t total = 0;
  PH_CHECK(hipMemcpyAsync(&total, base + 256, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PH_CHECK(hipStreamSynchronize(s));
  *n_rows = total;
  if (total > capacity) {
    ph_set_error("ph_pose_features: %d people, feature buffer holds %d", total, capacity);
    return -3;
  }
  if (total > 0) {
    PH_CHECK(hipMemcpyAsync(feat, base + off_ft, (size_t)total * row, hipMemcpyDeviceToHost, s));
    PH_CHECK(hipMemcpyAsync(status, base + off_st, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PH_CHECK(hipStreamSynchronize(s));
  }
  return 0;
}

// ---- MoveNet pre-/post-model steps -------------------------------------------------------------
namespace {
// host or device operand: a device copy for the duration of one call when the caller's pointer is host memory
struct MvBuf {
  void* dev = nullptr;
  void* host = nullptr;
  size_t bytes = 0;
  bool owned = false;
  int get(const void* p, size_t n, int on_device, bool copy_in, hipStream_t s) {
    bytes = n;
    if (on_device) {
      dev = const_cast<void*>(p);
      return 0;
    }
    host = const_cast<void*>(p);
    if (hipMalloc(&dev, n ? n : 1) != hipSuccess) return -1;
    owned = true;
    if (copy_in && hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return 0;
  }
  int put(hipStream_t s) { return owned && hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s) != hipSuccess ? -1 : 0; }
  ~MvBuf() {
    if (owned && dev) (void)hipFree(dev);
  }
};
}  // namespace

int ph_movenet_crop_resize(ph_ctx* c, const uint8_t* frames, int B, int H, int W, const double* crop, int size,
                           int32_t* out, int on_device) {
  PH_CHECK(hipSetDevice(c->device));
  if (!frames || !crop || !out || B < 1 || H < 1 || W < 1 || size < 1) {
    ph_set_error("ph_movenet_crop_resize: bad arguments");
    return -1;
  }
  hipStream_t s = c->stream;
  MvBuf f, r, o;
  int rc = f.get(frames, (size_t)B * H * W * 3, on_device, true, s) || r.get(crop, (size_t)B * 6 * sizeof(double), on_device, true, s) ||
           o.get(out, (size_t)B * size * size * 3 * sizeof(int32_t), on_device, false, s);
  if (!rc) rc = ph_launch_movenet_crop_resize((const uint8_t*)f.dev, B, H, W, (const double*)r.dev, size, (int32_t*)o.dev, s);
  if (!rc) rc = o.put(s);
  if (hipStreamSynchronize(s) != hipSuccess) rc = -1;
  if (rc) ph_set_error("ph_movenet_crop_resize: allocation, launch or copy failed");
  return rc;
}

int ph_movenet_update(ph_ctx* c, float* keypoints, const double* crop_in, int B, int image_height, int image_width,
                      double* crop_out, int on_device) {
  PH_CHECK(hipSetDevice(c->device));
  if (!keypoints || !crop_in || !crop_out || B < 1 || image_height < 1 || image_width < 1) {
    ph_set_error("ph_movenet_update: bad arguments");
    return -1;
  }
  hipStream_t s = c->stream;
  MvBuf k, ci, co;
  int rc = k.get(keypoints, (size_t)B * 51 * sizeof(float), on_device, true, s) ||
           ci.get(crop_in, (size_t)B * 6 * sizeof(double), on_device, true, s) ||
           co.get(crop_out, (size_t)B * 6 * sizeof(double), on_device, false, s);
  if (!rc) rc = ph_launch_movenet_update((float*)k.dev, (const double*)ci.dev, B, image_height, image_width, (double*)co.dev, s);
  if (!rc) rc = k.put(s) || co.put(s);
  if (hipStreamSynchronize(s) != hipSuccess) rc = -1;
  if (rc) ph_set_error("ph_movenet_update: allocation, launch or copy failed");
  return rc;
}

int ph_movenet_multipose_decode(ph_ctx* c, const float* center, const float* kpt_heat, const float* kpt_regr,
                                const float* kpt_offs, const float* box_size, int B, int G, int max_people,
                                float min_center, float* out, int on_device) {
  PH_CHECK(hipSetDevice(c->device));
  if (!center || !kpt_heat || !kpt_regr || !kpt_offs || !box_size || !out || B < 1 || G < 3 || G > 128 || max_people < 1 ||
      max_people > 64) {
    ph_set_error("ph_movenet_multipose_decode: bad arguments (grid 3..128, people 1..64)");
    return -1;
  }
  hipStream_t s = c->stream;
  const size_t n = (size_t)B * G * G * sizeof(float);
  // centre list (index, score per person slot) in the ctx scratch, grown on demand
  const size_t n_c = ((size_t)B * max_people * sizeof(int) + 255) & ~(size_t)255;
  if (2 * n_c > c->post_bytes) {
    PH_CHECK(hipStreamSynchronize(s));
    if (c->post) PH_CHECK(hipFree(c->post));
    c->post = nullptr;
    c->post_bytes = 0;
    PH_CHECK(hipMalloc(&c->post, 2 * n_c));
    c->post_bytes = 2 * n_c;
  }
  MvBuf ce, he, re, of, bx, o;
  int rc = ce.get(center, n, on_device, true, s) || he.get(kpt_heat, n * 17, on_device, true, s) ||
           re.get(kpt_regr, n * 34, on_device, true, s) || of.get(kpt_offs, n * 34, on_device, true, s) ||
           bx.get(box_size, n * 2, on_device, true, s) ||
           o.get(out, (size_t)B * max_people * 56 * sizeof(float), on_device, false, s);
  if (!rc)
    rc = ph_launch_movenet_multipose((const float*)ce.dev, (const float*)he.dev, (const float*)re.dev, (const float*)of.dev,
                                     (const float*)bx.dev, B, G, max_people, min_center, (int*)c->post,
                                     (float*)((char*)c->post + n_c), (float*)o.dev, s);
  if (!rc) rc = o.put(s);
  if (hipStreamSynchronize(s) != hipSuccess) rc = -1;
  if (rc) ph_set_error("ph_movenet_multipose_decode: allocation, launch or copy failed");
  return rc;
}

int ph_decode_v2_connections(ph_ctx* c, int32_t* n_conn, int32_t* conn_idx, float* conn_score) {
  PH_CHECK(hipSetDevice(c->device));
  if (!c->v2.n_conn) {
    ph_set_error("ph_decode_v2_connections: no decode has run");
    return -1;
  }
  hipStream_t s = c->stream;
  if (n_conn) PH_CHECK(hipMemcpyAsync(n_conn, c->v2.n_conn, c->v2sz.n_conn, hipMemcpyDeviceToHost, s));
  if (conn_idx) PH_CHECK(hipMemcpyAsync(conn_idx, c->v2.conn_idx, c->v2sz.conn_idx, hipMemcpyDeviceToHost, s));
  if (conn_score) PH_CHECK(hipMemcpyAsync(conn_score, c->v2.conn_score, c->v2sz.conn_score, hipMemcpyDeviceToHost, s));
  PH_CHECK(hipStreamSynchronize(s));
  return 0;
}

// ---- V1 decode ---------------------------------------------------------------
int ph_decode_v1(ph_ctx* c, const ph_v1_model* m, int B, int W0, int H0, float conf_thres, int kp_cap,
                 int person_cap, int32_t* n_kps, float* kps, int32_t* n_pairs, double* pairs, int32_t* n_persons,
                 double* persons) {
  PH_CHECK(hipSetDevice(c->device));
  if (!m || B < 1 || kp_cap < 1 || person_cap < 1 || !c->cur_out ||
      (size_t)B * c->cur_c * c->cur_h * c->cur_w > c->cur_elems) {
    ph_set_error("ph_decode_v1: bad args or no net output for batch %d", B);
    return -1;
  }
  if (m->n_parts < 1 || m->n_parts > V1_MAX_PARTS || m->n_limbs < 1 || m->n_limbs > V1_MAX_LIMBS) {
    ph_set_error("ph_decode_v1: bad model constants");
    return -1;
  }
  V1Model dev;
  memset(&dev, 0, sizeof(dev));
  dev.n_parts = m->n_parts;
  dev.n_limbs = m->n_limbs;
  memcpy(dev.pair_parts, m->pair_parts, sizeof(int) * 2 * m->n_limbs);
  memcpy(dev.mapidx, m->mapidx, sizeof(int) * 2 * m->n_limbs);
  V1Caps caps;
  caps.run_cap = 8192;
  caps.kp_cap = kp_cap;
  caps.person_cap = person_cap;
  const size_t key = ((size_t)B << 44) ^ ((size_t)kp_cap << 28) ^ ((size_t)person_cap << 12) ^
                     ((size_t)m->n_parts << 6) ^ (size_t)m->n_limbs;
  if (key != c->v1_key) {
    PH_CHECK(hipStreamSynchronize(c->stream));
    free_v1(c);
    ph_v1_sizes(dev, B, W0, H0, caps, &c->v1sz);
    PH_CHECK(hipMalloc(&c->v1.n_runs, c->v1sz.n_runs));
    PH_CHECK(hipMalloc(&c->v1.runs, c->v1sz.runs));
    PH_CHECK(hipMalloc(&c->v1.ws_verts, c->v1sz.ws_verts));
    PH_CHECK(hipMalloc(&c->v1.kps, c->v1sz.kps));
    PH_CHECK(hipMalloc(&c->v1.n_kps, c->v1sz.n_kps));
    PH_CHECK(hipMalloc(&c->v1.pairs, c->v1sz.pairs));
    PH_CHECK(hipMalloc(&c->v1.n_pairs, c->v1sz.n_pairs));
    PH_CHECK(hipMalloc(&c->v1.persons, c->v1sz.persons));
    PH_CHECK(hipMalloc(&c->v1.n_persons, c->v1sz.n_persons));
    PH_CHECK(hipMalloc(&c->v1.overflow, 4 * sizeof(int)));   // flag, blobs walked by a wave, by one group, -
    c->v1_key = key;
  }
  if (ph_launch_decode_v1(c->cur_out, c->cur_c, c->cur_h, c->cur_w, W0, H0, conf_thres, dev, B, caps, c->v1,
                          c->stream, c->ev1))
    return -1;
  hipStream_t s = c->stream;
  if (n_kps) PH_CHECK(hipMemcpyAsync(n_kps, c->v1.n_kps, c->v1sz.n_kps, hipMemcpyDeviceToHost, s));
  if (kps) PH_CHECK(hipMemcpyAsync(kps, c->v1.kps, c->v1sz.kps, hipMemcpyDeviceToHost, s));
  if (n_pairs) PH_CHECK(hipMemcpyAsync(n_pairs, c->v1.n_pairs, c->v1sz.n_pairs, hipMemcpyDeviceToHost, s));
  if (pairs) PH_CHECK(hipMemcpyAsync(pairs, c->v1.pairs, c->v1sz.pairs, hipMemcpyDeviceToHost, s));
  if (n_persons) PH_CHECK(hipMemcpyAsync(n_persons, c->v1.n_persons, c->v1sz.n_persons, hipMemcpyDeviceToHost, s));
  if (persons) PH_CHECK(hipMemcpyAsync(persons, c->v1.persons, c->v1sz.persons, hipMemcpyDeviceToHost, s));
  int flags[4] = {0, 0, 0, 0};
  PH_CHECK(hipMemcpyAsync(flags, c->v1.overflow, sizeof(flags), hipMemcpyDeviceToHost, s));
  PH_CHECK(hipStreamSynchronize(s));
  const int overflow = flags[0];
  c->v1_walks[0] = flags[1];
  c->v1_walks[1] = flags[2];
  if (overflow) {
    ph_set_error("ph_decode_v1: capacity exceeded (more than %d mask runs or %d contours per part, the contour vertex pool, or %d people)",
                 caps.run_cap, kp_cap, person_cap);
    return -2;
  }
  return 0;
}

int ph_decode_v1_times(ph_ctx* c, float* ms3) {
  PH_CHECK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 3; i++) PH_CHECK(hipEventElapsedTime(&ms3[i], c->ev1[i], c->ev1[i + 1]));
  return 0;
}

int ph_decode_v1_walks(ph_ctx* c, int32_t* wave, int32_t* group) {
  if (wave) *wave = c->v1_walks[0];
  if (group) *group = c->v1_walks[1];
  return 0;
}

int ph_decode_v2_times(ph_ctx* c, float* ms3) {
  PH_CHECK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 3; i++) PH_CHECK(hipEventElapsedTime(&ms3[i], c->ev[i], c->ev[i + 1]));
  return 0;
}

}  // extern "C"
