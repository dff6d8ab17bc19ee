// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the
// tiled SpMM kernel (developer tool; MI355X_MICROARCH.md, section HBM: "other
// access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Every kernel reads a 512 MiB buffer exactly once:
//   wide_dma     global_load_lds_dwordx4, 1 KiB per wave instruction (the B tiles)
//   wide_vgpr    global_load_dwordx4 into registers
//   narrow       global_load_dword, 256 B per wave instruction
//   window       global_load_dword, lane l reads dword l % 16: 64 B per wave
//                instruction, replicated four times (the entry windows)
// Run:  hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//       rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_calib
// and compare FETCH_SIZE (KiB) per kernel with the 524288 KiB each one reads.
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr size_t kBytes = size_t{512} << 20;

__global__ __launch_bounds__(256) void wide_dma(const float* __restrict__ src, float* out, size_t pieces) {
  __shared__ float tile[4][256];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x % 64;
  const unsigned lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) void*)(&tile[wave][0])));
  for (size_t p = static_cast<size_t>(blockIdx.x) * 4 + wave; p < pieces; p += static_cast<size_t>(gridDim.x) * 4) {
    const float* base = src + p * 256;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds), "v"(static_cast<unsigned>(lane * 16)), "s"(base) : "memory", "m0");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && tile[0][0] == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void wide_vgpr(const float4* __restrict__ src, float* out, size_t n4) {
  float acc = 0.f;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.f) out[0] = acc;
}

__global__ __launch_bounds__(256) void narrow(const float* __restrict__ src, float* out, size_t n) {
  float acc = 0.f;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
    acc += src[i];
  if (acc == 12345.f) out[0] = acc;
}

__global__ __launch_bounds__(256) void window(const float* __restrict__ src, float* out, size_t groups) {
  float acc = 0.f;
  const int lane16 = threadIdx.x % 16;
  for (size_t g = static_cast<size_t>(blockIdx.x) * 4 + threadIdx.x / 64; g < groups; g += static_cast<size_t>(gridDim.x) * 4)
    acc += src[g * 16 + lane16];
  if (acc == 12345.f) out[0] = acc;
}

int main() {
  float *src, *out;
  if (hipMalloc(&src, kBytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) return 1;
  (void)hipMemset(src, 0, kBytes);
  const int blocks = 256 * 8;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(wide_dma, dim3(blocks), dim3(256), 0, 0, src, out, kBytes / 1024);
    hipLaunchKernelGGL(wide_vgpr, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const float4*>(src), out, kBytes / 16);
    hipLaunchKernelGGL(narrow, dim3(blocks), dim3(256), 0, 0, src, out, kBytes / 4);
    hipLaunchKernelGGL(window, dim3(blocks), dim3(256), 0, 0, src, out, kBytes / 64);
  }
  (void)hipDeviceSynchronize();
  printf("each kernel read %zu KiB\n", kBytes / 1024);
  return 0;
}
