// Developer probe: does a workgroup with all 160 KiB of a CU's LDS launch on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(float* out) {
  __shared__ float tile[2][40 * 512];
  for (int i = threadIdx.x; i < 2 * 40 * 512; i += 1024) (&tile[0][0])[i] = i;
  __syncthreads();
  out[threadIdx.x + blockIdx.x * 1024] = tile[1][40 * 512 - 1 - threadIdx.x];
}
int main() {
  float* d; hipMalloc(&d, 4 * 1024 * 512);
  hipLaunchKernelGGL(k, dim3(512), dim3(1024), 0, 0, d);
  hipError_t e = hipDeviceSynchronize();
  float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("launch: %s, out[0]=%.0f (expect %d), sharedMemPerBlock=%zu maxSharedMemoryPerMultiProcessor=%zu\n",
         hipGetErrorString(e), h[0], 2 * 40 * 512 - 1, (size_t)p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor);
  return e != hipSuccess;
}
