// Layout pass for the reference's module design, for gfx950: the batched 2-D
// transpose  out[b][c][r] = in[b][r][c].
//
// The reference's SparseLinear computes y[B, out, S] = W @ x[B, S, in]^T and gets
// the k-major dense operand with `x.transpose(1, 2).contiguous()`
// (modules/sparse_linear.py:89); SparseAttention then moves every projection
// back (`.transpose(1, 2).contiguous()`), splits the heads with a second copy
// (`four_d_to_three_d` on a transposed view) and merges them again on the way
// out (modules/sparse_attention.py:108-126): eleven passes over the activations
// per forward, each an elementwise strided copy at ~1.5 TB/s.  All of them are
// instances of this one operation (the head split [B, H*D, S] -> [B*H, S, D] is
// a transpose of B*H matrices of D x S), which moves 64 x 64 tiles through LDS so
// that both the reads and the writes are 16-byte accesses along the contiguous
// dimension.  HBM-bound: 8 bytes per element.
#include "common.h"
#include "wave_utils.h"

namespace sputnik_hip {
namespace {

constexpr int kTile = 64;
constexpr int kThreads = 256;

// VEC = 4: rows, cols multiples of 4 and 16-byte aligned operands; VEC = 1: anything.
template <int VEC>
__global__ __launch_bounds__(kThreads) void transpose_tiles_kernel(
    int rows, int cols, int tiles_c, const float* __restrict__ in, int64_t in_batch_stride,
    float* __restrict__ out, int64_t out_batch_stride) {
  __shared__ float tile[kTile][kTile + 1];
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const int r0 = tr * kTile, c0 = tc * kTile;
  in += blockIdx.y * in_batch_stride;
  out += blockIdx.y * out_batch_stride;
  const int t = threadIdx.x;
  if constexpr (VEC == 4) {
    const int tx = t % 16, ty = t / 16;   // 16 float4 across, 16 rows per pass
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + ty + 16 * j, c = c0 + 4 * tx;
      if (r < rows && c < cols) {
        const float4 v = *reinterpret_cast<const float4*>(in + static_cast<int64_t>(r) * cols + c);
        tile[ty + 16 * j][4 * tx + 0] = v.x;
        tile[ty + 16 * j][4 * tx + 1] = v.y;
        tile[ty + 16 * j][4 * tx + 2] = v.z;
        tile[ty + 16 * j][4 * tx + 3] = v.w;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + ty + 16 * j, r = r0 + 4 * tx;   // output row c, columns r .. r+3
      if (c < cols && r < rows) {
        const float4 v = make_float4(tile[4 * tx + 0][ty + 16 * j], tile[4 * tx + 1][ty + 16 * j],
                                     tile[4 * tx + 2][ty + 16 * j], tile[4 * tx + 3][ty + 16 * j]);
        *reinterpret_cast<float4*>(out + static_cast<int64_t>(c) * rows + r) = v;
      }
    }
  } else {
    const int tx = t % kTile, ty = t / kTile;   // 64 across, 4 rows per pass
    for (int j = 0; j < kTile / 4; ++j) {
      const int r = r0 + ty + 4 * j, c = c0 + tx;
      if (r < rows && c < cols) tile[ty + 4 * j][tx] = in[static_cast<int64_t>(r) * cols + c];
    }
    __syncthreads();
    for (int j = 0; j < kTile / 4; ++j) {
      const int c = c0 + ty + 4 * j, r = r0 + tx;
      if (c < cols && r < rows) out[static_cast<int64_t>(c) * rows + r] = tile[tx][ty + 4 * j];
    }
  }
}

}  // namespace
}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

int sputnik_hip_transpose_batched(int batches, int rows, int cols, const float* in,
                                  int64_t in_batch_stride, float* out, int64_t out_batch_stride,
                                  sputnik_hip_stream_t stream) {
  if (batches < 0 || rows < 0 || cols < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (batches == 0 || rows == 0 || cols == 0) return 0;
  const int tiles_r = ceil_div(rows, kTile), tiles_c = ceil_div(cols, kTile);
  if (static_cast<int64_t>(tiles_r) * tiles_c > 0x7fffffff) return SPUTNIK_HIP_INVALID_ARGUMENT;
  const bool vec = rows % 4 == 0 && cols % 4 == 0 && aligned_to(in, 16) && aligned_to(out, 16) &&
                   in_batch_stride % 4 == 0 && out_batch_stride % 4 == 0;
  for (int b0 = 0; b0 < batches; b0 += kMaxGridYZ) {
    const int by = min(batches - b0, kMaxGridYZ);
    const dim3 grid(tiles_r * tiles_c, by);
    const float* in_b = in + b0 * in_batch_stride;
    float* out_b = out + b0 * out_batch_stride;
    if (vec)
      hipLaunchKernelGGL(transpose_tiles_kernel<4>, grid, dim3(kThreads), 0, stream, rows, cols,
                         tiles_c, in_b, in_batch_stride, out_b, out_batch_stride);
    else
      hipLaunchKernelGGL(transpose_tiles_kernel<1>, grid, dim3(kThreads), 0, stream, rows, cols,
                         tiles_c, in_b, in_batch_stride, out_b, out_batch_stride);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

}  // extern "C"
