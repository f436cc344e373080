// Micro-benchmark (diagnostic, not part of the library): LDS-DMA fill rate of one CU from an L2-resident matrix as a
// function of the piece shape -- how many bytes of each row one 1-KiB global_load_lds_dwordx4 wave-instruction takes.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// rows of `row_bytes` bytes; a piece takes `seg` bytes (64, 128, 256, 1024) from each of 1024 / seg rows
template <int SEG, int WAVES, int INFLIGHT>
__global__ __launch_bounds__(64 * WAVES) void fill(const char* src, int rows, int row_bytes, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = SEG / 16;          // lanes per row
  constexpr int RPP = 64 / LPR;          // rows per piece
  const int segs_per_row = row_bytes / SEG;
  const unsigned lane_off = (unsigned)(lane / LPR) * (unsigned)row_bytes + (unsigned)(lane % LPR) * 16u;
  const int pieces_per_col = rows / RPP;
  int piece = wave + blockIdx.x * 3;     // walk the matrix: piece -> (row group, segment)
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) {
      const int pg = piece % pieces_per_col, sg = (piece / pieces_per_col) % segs_per_row;
      const unsigned off = (unsigned)(pg * RPP) * (unsigned)row_bytes + (unsigned)(sg * SEG) + lane_off;
      __builtin_amdgcn_global_load_lds((glb_void*)(src + off), (lds_void*)(lds + ((it & 1) * INFLIGHT + u) * WAVES * 1024 + wave * 1024),
                                       16, 0, 0);
      piece += WAVES;
    }
    if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = ((int*)lds)[3];
}

template <int SEG, int WAVES, int INFLIGHT>
static void run(const char* name, const char* d, int rows, int row_bytes, int* sink) {
  const int iters = 2000, grid = 256;
  const int ldsb = 2 * INFLIGHT * WAVES * 1024;
  hipFuncSetAttribute((const void*)fill<SEG, WAVES, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL((fill<SEG, WAVES, INFLIGHT>), dim3(grid), dim3(64 * WAVES), ldsb, 0, d, rows, row_bytes, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * iters * INFLIGHT * WAVES * 1024.0;
  printf("%-44s seg %4d B  waves %2d  in flight %2d KB/CU  %7.1f GB/s per CU  %6.2f TB/s chip\n", name, SEG, WAVES,
         2 * INFLIGHT * WAVES, bytes / grid / (ms * 1e-3) * 1e-9, bytes / (ms * 1e-3) * 1e-12);
}

int main() {
  const int rows = 512, row_bytes = 768;   // the 384 -> 512 weight matrix, bf16
  char* d;
  int* sink;
  hipMalloc(&d, (size_t)rows * row_bytes + 4096);
  hipMemset(d, 1, (size_t)rows * row_bytes + 4096);
  hipMalloc(&sink, 256 * sizeof(int));
  run<64, 8, 2>("16 rows x 64 B (the conv kernels' pieces)", d, rows, row_bytes, sink);
  run<128, 8, 2>("8 rows x 128 B", d, rows, row_bytes, sink);
  run<256, 8, 2>("4 rows x 256 B", d, rows, row_bytes, sink);
  run<64, 8, 4>("16 rows x 64 B", d, rows, row_bytes, sink);
  run<128, 8, 4>("8 rows x 128 B", d, rows, row_bytes, sink);
  run<256, 8, 4>("4 rows x 256 B", d, rows, row_bytes, sink);
  run<64, 8, 8>("16 rows x 64 B", d, rows, row_bytes, sink);
  run<128, 8, 8>("8 rows x 128 B", d, rows, row_bytes, sink);
  run<256, 8, 8>("4 rows x 256 B", d, rows, row_bytes, sink);
  run<64, 16, 4>("16 rows x 64 B", d, rows, row_bytes, sink);
  run<128, 16, 4>("8 rows x 128 B", d, rows, row_bytes, sink);
  // a big matrix (96 MB: beyond L2, within the Infinity Cache): the activations' side
  char* big;
  const int brows = 120704;
  hipMalloc(&big, (size_t)brows * row_bytes + 4096);
  hipMemset(big, 1, (size_t)brows * row_bytes + 4096);
  run<64, 8, 4>("93 MB matrix, 16 rows x 64 B", big, brows, row_bytes, sink);
  run<128, 8, 4>("93 MB matrix, 8 rows x 128 B", big, brows, row_bytes, sink);
  run<256, 8, 4>("93 MB matrix, 4 rows x 256 B", big, brows, row_bytes, sink);
  run<64, 8, 8>("93 MB matrix, 16 rows x 64 B", big, brows, row_bytes, sink);
  run<256, 8, 8>("93 MB matrix, 4 rows x 256 B", big, brows, row_bytes, sink);
  return 0;
}
