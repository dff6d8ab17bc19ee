// MEASUREMENT ONLY -- not part of the product, which uses no MFMA
// (north star: the five operators are irregular gathers).  Question: for HALF-storage
// operands at attention density, does a DENSE bf16 MFMA tile product followed by
// sampling at the mask beat the sparse kernels?  At density 0.1 every 128 x 128 tile of a
// 1024 x 1024 mask is occupied, so the dense product does 10 x the flops of the sparse
// one -- on a unit that is 16 x faster than the packed-fp32 vector pipe.
//
// Config 3's attention scores: out[r][p] = <Q_r[i_p, :], K_r[j_p, :]>, S = 1024, D = 64,
// 64 replicas, mask density 0.1, bf16 operands, float32 output (exact products, float32
// accumulation: the same arithmetic contract as v_dot2_f32_bf16 in sddmm_quad_kernel /
// sddmm_flat_kernel, other summation order).
//
// One workgroup (4 waves) = one 128 x 128 tile of one replica's Q K^T:
//   * every wave computes a 64 x 64 quarter as 2 x 2 tiles of v_mfma_f32_32x32x16_bf16,
//     4 k-steps; the operand fragments are 16-byte loads straight from global memory
//     (lane l: row l & 31, k = 8 (l >> 5) + j: rows of Q and of K are both k-contiguous);
//   * the 128 x 128 float tile goes to LDS (C/D layout: col = lane & 31, row = (reg & 3) +
//     8 (reg >> 2) + 4 (lane >> 5));
//   * a 16-lane group per mask row copies the row's entries inside the tile's column
//     range from LDS to out (their CSR positions: a per-(row, column tile) start table).
//
// Build and run (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_sddmm_bench.hip -o tools/bin/mfma_sddmm_bench
//   tools/bin/mfma_sddmm_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)

using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

constexpr int kT = 128, kPitch = kT + 4;

// kS x kS mask, inner dimension kD (a multiple of 64); sum_reps: the tile accumulates over
// `sum_reps` replicas too (the gradient of weights shared by a batch: ONE output vector)
__global__ __launch_bounds__(256) void mfma_sddmm_kernel(const __bf16* __restrict__ q,
                                                        const __bf16* __restrict__ k,
                                                        const int* __restrict__ column_indices,
                                                        const int* __restrict__ tile_start /* [S][S/T + 1] */,
                                                        float* __restrict__ out, int nnz, int sample,
                                                        int kS, int kD, int sum_reps) {
  __shared__ float tile[kT * kPitch];
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
  const int wr = wave >> 1, wc = wave & 1;
  const int ct = blockIdx.x, rt = blockIdx.y, rep = blockIdx.z;
  const int r0 = rt * kT, c0 = ct * kT;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{};
  for (int sr = 0; sr < sum_reps; ++sr) {
    const __bf16* qr = q + static_cast<int64_t>(rep * sum_reps + sr) * kS * kD;
    const __bf16* kr = k + static_cast<int64_t>(rep * sum_reps + sr) * kS * kD;
    for (int k0 = 0; k0 < kD; k0 += 64) {
      bf16x8 a[2][4], b[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kk = k0 + ks * 16 + 8 * (lane >> 5);
          a[t][ks] = *reinterpret_cast<const bf16x8*>(
              qr + static_cast<int64_t>(r0 + wr * 64 + t * 32 + (lane & 31)) * kD + kk);
          b[t][ks] = *reinterpret_cast<const bf16x8*>(
              kr + static_cast<int64_t>(c0 + wc * 64 + t * 32 + (lane & 31)) * kD + kk);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b[j][ks], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int col = wc * 64 + j * 32 + (lane & 31);
        tile[row * kPitch + col] = acc[i][j][reg];
      }
  __syncthreads();
  if (!sample) return;
  // a 16-lane group per mask row of the tile
  const int g = threadIdx.x >> 4, i16 = threadIdx.x & 15;
  float* o = out + static_cast<int64_t>(rep) * nnz;
  for (int r = g; r < kT; r += 16) {
    const int row = r0 + r;
    const int p0 = tile_start[row * (kS / kT + 1) + ct], p1 = tile_start[row * (kS / kT + 1) + ct + 1];
    for (int p = p0 + i16; p < p1; p += 16) o[p] = tile[r * kPitch + (column_indices[p] - c0)];
  }
}

static float bf16_round(float v) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float r;
  std::memcpy(&r, &u, 4);
  return r;
}
static __bf16 to_bf16(float v) {
  const float r = bf16_round(v);
  uint32_t u;
  std::memcpy(&u, &r, 4);
  const uint16_t h = static_cast<uint16_t>(u >> 16);
  __bf16 out;
  std::memcpy(&out, &h, 2);
  return out;
}

static int run(const char* name, int kS, int kD, int reps, int one_in, int sum_reps) {
  uint64_t state = 88172645463325252ull;
  auto rnd = [&]() {
    state ^= state << 13;
    state ^= state >> 7;
    state ^= state << 17;
    return static_cast<uint32_t>(state >> 32);
  };
  std::vector<int> ro(kS + 1, 0), ci;
  for (int r = 0; r < kS; ++r) {
    for (int c = 0; c < kS; ++c)
      if (rnd() % one_in == 0) ci.push_back(c);
    ro[r + 1] = static_cast<int>(ci.size());
  }
  const int nnz = static_cast<int>(ci.size());
  const int tiles = kS / kT;
  std::vector<int> tstart(static_cast<size_t>(kS) * (tiles + 1));
  for (int r = 0; r < kS; ++r) {
    int p = ro[r];
    for (int t = 0; t <= tiles; ++t) {
      while (p < ro[r + 1] && ci[p] < t * kT) ++p;
      tstart[static_cast<size_t>(r) * (tiles + 1) + t] = p;
    }
  }
  std::vector<float> qf(static_cast<size_t>(reps) * kS * kD), kf(qf.size());
  std::vector<__bf16> qh(qf.size()), kh(qf.size());
  for (size_t i = 0; i < qf.size(); ++i) {
    qf[i] = bf16_round((rnd() % 2001) / 1000.f - 1.f);
    kf[i] = bf16_round((rnd() % 2001) / 1000.f - 1.f);
    qh[i] = to_bf16(qf[i]);
    kh[i] = to_bf16(kf[i]);
  }
  const int outs = reps / sum_reps;
  __bf16 *dq, *dk;
  int *dci, *dts;
  float* dout;
  CHECK(hipMalloc(&dq, qh.size() * 2));
  CHECK(hipMalloc(&dk, kh.size() * 2));
  CHECK(hipMalloc(&dci, ci.size() * 4));
  CHECK(hipMalloc(&dts, tstart.size() * 4));
  CHECK(hipMalloc(&dout, static_cast<size_t>(outs) * nnz * 4));
  CHECK(hipMemcpy(dq, qh.data(), qh.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dk, kh.data(), kh.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dts, tstart.data(), tstart.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(dout, 0xff, static_cast<size_t>(outs) * nnz * 4));

  const dim3 grid(tiles, tiles, outs);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int sample = 1; sample >= 0; --sample) {
    for (int i = 0; i < 5; ++i)
      hipLaunchKernelGGL(mfma_sddmm_kernel, grid, dim3(256), 0, 0, dq, dk, dci, dts, dout, nnz, sample, kS, kD,
                         sum_reps);
    CHECK(hipEventRecord(e0));
    const int iters = 50;
    for (int i = 0; i < iters; ++i)
      hipLaunchKernelGGL(mfma_sddmm_kernel, grid, dim3(256), 0, 0, dq, dk, dci, dts, dout, nnz, sample, kS, kD,
                         sum_reps);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    std::printf("{\"config\": \"%s\", \"variant\": \"%s\", \"us\": %.2f, \"sparse_tflops\": %.2f, "
                "\"dense_tflops\": %.1f}\n", name,
                sample ? "dense bf16 MFMA tiles + sampling" : "dense tiles only (no sampling, no output)", us,
                2.0 * nnz * kD * reps / us / 1e6, 2.0 * kS * static_cast<double>(kS) * kD * reps / us / 1e6);
  }
  // check: every entry of the first and the last output against float64 on the same operands
  std::vector<float> got(static_cast<size_t>(outs) * nnz);
  CHECK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (int o : {0, outs - 1})
    for (int r = 0; r < kS; r += 7)
      for (int p = ro[r]; p < ro[r + 1]; ++p) {
        double want = 0;
        for (int sr = 0; sr < sum_reps; ++sr)
          for (int d = 0; d < kD; ++d)
            want += static_cast<double>(qf[(static_cast<size_t>(o * sum_reps + sr) * kS + r) * kD + d]) *
                    kf[(static_cast<size_t>(o * sum_reps + sr) * kS + ci[p]) * kD + d];
        worst = std::fmax(worst, std::fabs(got[static_cast<size_t>(o) * nnz + p] - want));
        scale = std::fmax(scale, std::fabs(want));
      }
  std::printf("{\"config\": \"%s\", \"nnz\": %d, \"max_abs_error_vs_float64\": %.3g, \"largest_value\": %.3g}\n",
              name, nnz, worst, scale);
  CHECK(hipFree(dq));
  CHECK(hipFree(dk));
  CHECK(hipFree(dci));
  CHECK(hipFree(dts));
  CHECK(hipFree(dout));
  return worst <= 1e-5 * (1.0 + scale) ? 0 : 1;
}

int main() {
  int bad = 0;
  // config 3's attention scores: 1024^2 mask at density 0.1, k = 64, 64 replicas
  bad |= run("c3 scores 1024^2 d0.1 k64 x64", 1024, 64, 64, 10, 1);
  // config 5's weight gradient: 2048^2 mask at density 0.2, k = seq 512, summed over batch 8
  bad |= run("c5 weight gradient 2048^2 d0.2 k512 sum of 8", 2048, 512, 8, 5, 8);
  return bad;
}
