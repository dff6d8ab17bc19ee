// How fast can a CU bring bytes from L2 into LDS?  (developer tool, not shipped)
//   a) global_load_lds_dwordx4  (LDS-DMA: 1 KiB per wave instruction, no registers)
//   b) global_load_dwordx4 to registers + ds_write_b128
// 16 waves per workgroup, one workgroup per CU, every wave moves PIECES x 1 KiB per
// iteration from an L2-resident source (the same 64 KiB per CU over and over).
//
//   hipcc --offload-arch=gfx950 -O2 tools/ldsdma_bench.hip -o tools/bin/ldsdma_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

constexpr int kPieces = 4;

__global__ __launch_bounds__(1024) void k_dma(const float* src, int iters, float* sink) {
  __shared__ float lds[32768];   // 128 KiB
  const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
  const float* base = src + (blockIdx.x % 64) * 16384;   // 64 KiB per block, L2 resident
  const unsigned off = lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const int piece = wave + 16 * p;
      const unsigned lds_addr = (it & 1) * 65536 + piece * 1024;
      const float* row = base + piece * 256;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   : : "s"(lds_addr), "v"(off), "s"(row) : "memory", "m0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (lds[threadIdx.x] == -1.f) sink[0] = 1.f;
}

__global__ __launch_bounds__(1024) void k_reg(const float* src, int iters, float* sink) {
  __shared__ float lds[32768];
  const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
  const float* base = src + (blockIdx.x % 64) * 16384;
  for (int it = 0; it < iters; ++it) {
    float4 v[kPieces];
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const int piece = wave + 16 * p;
      v[p] = *reinterpret_cast<const float4*>(base + piece * 256 + lane * 4);
    }
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const int piece = wave + 16 * p;
      *reinterpret_cast<float4*>(&lds[(it & 1) * 16384 + piece * 256 + lane * 4]) = v[p];
    }
    __syncthreads();
  }
  if (lds[threadIdx.x] == -1.f) sink[0] = 1.f;
}

int main() {
  float *src, *sink;
  CHECK(hipMalloc(&src, 64 * 65536));
  CHECK(hipMemset(src, 0, 64 * 65536));
  CHECK(hipMalloc(&sink, 64));
  const int iters = 2000;
  for (int which = 0; which < 2; ++which) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_dma, dim3(256), dim3(1024), 0, 0, src, iters, sink);
      else hipLaunchKernelGGL(k_reg, dim3(256), dim3(1024), 0, 0, src, iters, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_cu = double(iters) * 65536;
    printf("%s: %.3f ms  %.1f GB/s per CU  %.2f TB/s chip  (%.1f B/clk/CU at 2.1 GHz)\n",
           which == 0 ? "lds-dma dwordx4       " : "load dwordx4 + ds_write", ms,
           bytes_per_cu / ms / 1e6, bytes_per_cu * 256 / ms / 1e9, bytes_per_cu / (ms * 1e-3) / 2.1e9);
  }
  return 0;
}
