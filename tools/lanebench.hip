// Feasibility microbenchmark (developer tool, not shipped): "one lane = one
// output row" SpMM inner loop.  Every lane walks its own list of (column,
// value) pairs; per pair it reads the 64-float row `column` of a B tile in LDS
// as 16 ds_read_b128 in a per-lane rotated piece order (piece (kk + lane) % 16,
// conflict-free for any set of rows) and does 64 FMAs into private
// accumulators.  Reports FMA rate and LDS bandwidth for several waves per CU.
//
//   hipcc --offload-arch=gfx950 -O3 tools/lanebench.hip -o /tmp/lanebench && /tmp/lanebench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
      exit(1);                                                                            \
    }                                                                                     \
  } while (0)

constexpr int kBK = 256;  // rows of the B tile (64 floats each): 64 KiB

template <int WAVES, bool ROTATE>
__global__ __launch_bounds__(WAVES * 64) void lane_kernel(const int* __restrict__ cols,
                                                          const float* __restrict__ vals,
                                                          int iters, float* __restrict__ out) {
  __shared__ float tile[kBK * 64];
  for (int i = threadIdx.x; i < kBK * 64; i += WAVES * 64) tile[i] = 1.0f + (i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rot = ROTATE ? (lane & 15) : 0;
  float4 acc[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int* c = cols + (static_cast<size_t>(blockIdx.x) * WAVES + wave) * iters * 64 + lane;
  const float* v = vals + (static_cast<size_t>(blockIdx.x) * WAVES + wave) * iters * 64 + lane;
  int col = c[0];
  float a = v[0];
  for (int t = 0; t < iters; ++t) {
    const int tn = min(t + 1, iters - 1);
    const int col_next = c[tn * 64];
    const float a_next = v[tn * 64];
    const float* row = tile + col * 64;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 b = *reinterpret_cast<const float4*>(row + (((kk + rot) & 15) << 2));
      acc[kk].x = fmaf(a, b.x, acc[kk].x);
      acc[kk].y = fmaf(a, b.y, acc[kk].y);
      acc[kk].z = fmaf(a, b.z, acc[kk].z);
      acc[kk].w = fmaf(a, b.w, acc[kk].w);
    }
    col = col_next;
    a = a_next;
  }
  float s = 0.f;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) s += acc[kk].x + acc[kk].y + acc[kk].z + acc[kk].w;
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

// Same work, software pipelined by hand at half-pair granularity: two sets of
// eight float4 registers; the reads of the next half are issued before the 32
// FMAs of the current half.  Pairs arrive four at a time per lane, three
// groups ahead, in a statically indexed ring of three register sets.
template <int WAVES, bool STREAM>
__global__ __launch_bounds__(WAVES * 64) void lane_kernel_pipelined(
    const int* __restrict__ cols, const float* __restrict__ vals, int iters,
    float* __restrict__ out) {
  __shared__ float tile[kBK * 64];
  for (int i = threadIdx.x; i < kBK * 64; i += WAVES * 64) tile[i] = 1.0f + (i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rot = lane & 15;
  float4 acc[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ha[8], hb[8];
  // piece offsets (in floats) of this lane's rotated order
  auto issue = [&](float4 (&h)[8], int col, int half) {
    const float* row = tile + col * 64;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      h[kk] = *reinterpret_cast<const float4*>(row + (((half * 8 + kk + rot) & 15) << 2));
  };
  auto consume = [&](const float4 (&h)[8], float a, int half) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float4& c = acc[half * 8 + kk];
      c.x = fmaf(a, h[kk].x, c.x);
      c.y = fmaf(a, h[kk].y, c.y);
      c.z = fmaf(a, h[kk].z, c.z);
      c.w = fmaf(a, h[kk].w, c.w);
    }
  };
  // STREAM: every lane reads its own contiguous run of pairs (a CSR row), so one
  // load instruction touches 64 different cache lines; otherwise [group][lane].
  const int groups = iters / 4;
  const int gstride = STREAM ? 1 : 64;
  const size_t wave_base = (static_cast<size_t>(blockIdx.x) * WAVES + wave) * groups * 64;
  const int4* c4 = reinterpret_cast<const int4*>(cols) + wave_base +
                   (STREAM ? static_cast<size_t>(lane) * groups : lane);
  const float4* v4 = reinterpret_cast<const float4*>(vals) + wave_base +
                     (STREAM ? static_cast<size_t>(lane) * groups : lane);
  int4 c0 = c4[0], c1 = c4[min(1, groups - 1) * gstride], c2 = c4[min(2, groups - 1) * gstride];
  float4 v0 = v4[0], v1 = v4[min(1, groups - 1) * gstride], v2 = v4[min(2, groups - 1) * gstride];
  issue(ha, c0.x, 0);
  auto pair = [&](int col, float a, int next_col) {
    issue(hb, col, 1);
    consume(ha, a, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue(ha, next_col, 0);
    consume(hb, a, 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto group = [&](int4& cc_slot, float4& vv_slot, const int4& next_slot, int g) {
    const int4 cc = cc_slot;
    const float4 vv = vv_slot;
    const int gn = min(g + 3, groups - 1);
    cc_slot = c4[gn * gstride];
    vv_slot = v4[gn * gstride];
    pair(cc.x, vv.x, cc.y);
    pair(cc.y, vv.y, cc.z);
    pair(cc.z, vv.z, cc.w);
    pair(cc.w, vv.w, next_slot.x);
  };
#pragma unroll 1
  for (int g = 0; g + 2 < groups; g += 3) {
    group(c0, v0, c1, g);
    group(c1, v1, c2, g + 1);
    group(c2, v2, c0, g + 2);
  }
  float s = 0.f;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) s += acc[kk].x + acc[kk].y + acc[kk].z + acc[kk].w + ha[kk & 7].x;
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

template <int WAVES, bool ROTATE, bool PIPE = false, bool STREAM = false>
void run(const char* name, int blocks, int iters) {
  const size_t n = static_cast<size_t>(blocks) * WAVES * iters * 64;
  std::vector<int> h_cols(n);
  std::vector<float> h_vals(n, 0.5f);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h_cols[i] = (s >> 8) % kBK;
  }
  int* cols;
  float *vals, *out;
  CHECK(hipMalloc(&cols, n * 4));
  CHECK(hipMalloc(&vals, n * 4));
  CHECK(hipMalloc(&out, static_cast<size_t>(blocks) * WAVES * 64 * 4));
  CHECK(hipMemcpy(cols, h_cols.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(vals, h_vals.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    if (PIPE)
      hipLaunchKernelGGL((lane_kernel_pipelined<WAVES, STREAM>), dim3(blocks), dim3(WAVES * 64), 0, 0, cols,
                         vals, iters, out);
    else
      hipLaunchKernelGGL((lane_kernel<WAVES, ROTATE>), dim3(blocks), dim3(WAVES * 64), 0, 0, cols,
                         vals, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) {
      const double fma = static_cast<double>(n) * 64;
      printf("%-28s blocks=%d waves/block=%d iters=%d  %.3f ms  %.1f TFLOP/s  LDS %.1f TB/s\n",
             name, blocks, WAVES, iters, ms, 2 * fma / ms * 1e-9, fma * 4 / ms * 1e-9);
    }
  }
  CHECK(hipFree(cols));
  CHECK(hipFree(vals));
  CHECK(hipFree(out));
}

int main() {
  run<4, true>("rotated, 4 waves", 256, 2000);
  run<8, true>("rotated, 8 waves", 256, 1000);
  run<16, true>("rotated, 16 waves", 256, 500);
  run<4, false>("unrotated, 4 waves", 256, 2000);
  run<4, true, true>("pipelined, 4 waves", 256, 2040);
  run<8, true, true>("pipelined, 8 waves", 256, 1020);
  run<4, true, true, true>("pipelined+stream, 4 waves", 256, 2040);
  run<8, true, true, true>("pipelined+stream, 8 waves", 256, 1020);
  return 0;
}
