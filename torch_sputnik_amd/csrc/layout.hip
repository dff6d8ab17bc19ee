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
// dimension.  HBM-bound: 8 bytes per fp32 element.  The pass may also change
// the storage type (fp16 / bf16 <-> fp32): BASELINE config 5 stores activations
// in half precision, and widening them here costs no pass of its own.
#include "common.h"
#include "wave_utils.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace sputnik_hip {
namespace {

constexpr int kTile = 64;
constexpr int kThreads = 256;

// Element types: float, __half, __hip_bfloat16 -- the pass may change the storage
// type on the way (half-precision activations are widened to float HERE, on
// their way into left_spmm, instead of in a pass of their own; gradients are
// narrowed back the same way).  Arithmetic type of the tile: float.
template <typename T>
__device__ __forceinline__ float to_float(T v) { return static_cast<float>(v); }
template <>
__device__ __forceinline__ float to_float<__half>(__half v) { return __half2float(v); }
template <>
__device__ __forceinline__ float to_float<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T>
__device__ __forceinline__ T from_float(float v);
template <>
__device__ __forceinline__ float from_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ __half from_float<__half>(float v) { return __float2half(v); }
template <>
__device__ __forceinline__ __hip_bfloat16 from_float<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename T>
struct alignas(4 * sizeof(T)) Quad {
  T v[4];
};

// VEC = 4: rows, cols multiples of 4 and operands aligned to four elements;
// VEC = 1: anything.
template <typename TIn, typename TOut, int VEC>
__global__ __launch_bounds__(kThreads) void transpose_tiles_kernel(
    int rows, int cols, int tiles_c, const TIn* __restrict__ in, int64_t in_batch_stride,
    TOut* __restrict__ out, int64_t out_batch_stride) {
  __shared__ float tile[kTile][kTile + 1];
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const int r0 = tr * kTile, c0 = tc * kTile;
  in += blockIdx.y * in_batch_stride;
  out += blockIdx.y * out_batch_stride;
  const int t = threadIdx.x;
  if constexpr (VEC == 4) {
    const int tx = t % 16, ty = t / 16;   // 16 quads across, 16 rows per pass
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + ty + 16 * j, c = c0 + 4 * tx;
      if (r < rows && c < cols) {
        const Quad<TIn> q = *reinterpret_cast<const Quad<TIn>*>(in + static_cast<int64_t>(r) * cols + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ty + 16 * j][4 * tx + e] = to_float(q.v[e]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + ty + 16 * j, r = r0 + 4 * tx;   // output row c, columns r .. r+3
      if (c < cols && r < rows) {
        Quad<TOut> q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q.v[e] = from_float<TOut>(tile[4 * tx + e][ty + 16 * j]);
        *reinterpret_cast<Quad<TOut>*>(out + static_cast<int64_t>(c) * rows + r) = q;
      }
    }
  } else {
    const int tx = t % kTile, ty = t / kTile;   // 64 across, 4 rows per pass
    for (int j = 0; j < kTile / 4; ++j) {
      const int r = r0 + ty + 4 * j, c = c0 + tx;
      if (r < rows && c < cols) tile[ty + 4 * j][tx] = to_float(in[static_cast<int64_t>(r) * cols + c]);
    }
    __syncthreads();
    for (int j = 0; j < kTile / 4; ++j) {
      const int c = c0 + ty + 4 * j, r = r0 + tx;
      if (c < cols && r < rows)
        out[static_cast<int64_t>(c) * rows + r] = from_float<TOut>(tile[tx][ty + 4 * j]);
    }
  }
}

// out[r][i] = in[r][perm[i]] for R value rows sharing one permutation: the
// transposed-topology values of a static pattern (functional.TransposeCache keeps
// the permutation; the reference re-runs csr_transpose per backward,
// modules/spmm.py:59-62).  A thread owns one output slot i for all rows: the
// permutation entry is read once, the writes are coalesced; the reads are 4-byte
// gathers inside one row of `in` at a time (a row of a few hundred KB: L2 hits).
constexpr int kGatherRows = 8;   // value rows in flight per thread
__global__ __launch_bounds__(kThreads) void permute_last_kernel(
    int n, int rows, const float* __restrict__ in, int64_t in_stride,
    const int* __restrict__ perm, float* __restrict__ out, int64_t out_stride) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int p = perm[i];
  const int r0 = blockIdx.y * kGatherRows;
  float v[kGatherRows];
#pragma unroll
  for (int j = 0; j < kGatherRows; ++j)
    v[j] = r0 + j < rows ? in[(r0 + j) * in_stride + p] : 0.f;
#pragma unroll
  for (int j = 0; j < kGatherRows; ++j)
    if (r0 + j < rows) out[(r0 + j) * out_stride + i] = v[j];
}

// The same permutation through LDS, for MANY rows of values (attention weights:
// one row per batch x head).  The plain kernel's 4-byte gathers each pull a whole
// 64-byte sector out of L2 (16 x the useful bytes: 41 us for 64 rows of 105 k
// entries, the L2 -> L1 rate).  Here the positions are processed grouped by the
// BAND of kPermuteBand consecutive source entries their value comes from
// (`dest_list`: output positions, band after band, ascending inside a band;
// `source_in_band`: the source's offset inside its band).  A workgroup copies one
// band of one row into LDS with coalesced reads, gathers from LDS, and writes
// runs of consecutive output positions (a band of a transposed CSR pattern holds
// a run of every column's entries).  The lists depend on the permutation alone
// and are reused for kRowsPerGroup rows per workgroup.
constexpr int kPermuteBand = 16384;       // 64 KiB of LDS
constexpr int kBandThreads = 1024;
constexpr int kBandPerThread = kPermuteBand / kBandThreads;
constexpr int kRowsPerGroup = 2;

template <bool VEC>
__global__ __launch_bounds__(kBandThreads) void permute_banded_kernel(
    int n, int rows, const float* __restrict__ in, int64_t in_stride,
    const int* __restrict__ dest_list, const int* __restrict__ source_in_band,
    float* __restrict__ out, int64_t out_stride) {
  __shared__ float band[kPermuteBand];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * kPermuteBand;
  const int count = min(kPermuteBand, n - base);
  const int r0 = blockIdx.y * kRowsPerGroup;
  int dest[kBandPerThread], src[kBandPerThread];
#pragma unroll
  for (int j = 0; j < kBandPerThread; ++j) {
    const int t = j * kBandThreads + tid;
    dest[j] = t < count ? dest_list[base + t] : -1;
    src[j] = t < count ? source_in_band[base + t] : 0;
  }
  for (int r = r0; r < min(r0 + kRowsPerGroup, rows); ++r) {
    const float* __restrict__ row_in = in + r * in_stride + base;
    if (r > r0) __syncthreads();   // the previous row's gathers are done
    if (VEC) {
#pragma unroll
      for (int j = 0; j < kBandPerThread / 4; ++j) {
        const int t = (j * kBandThreads + tid) * 4;
        if (t + 3 < count) {
          *reinterpret_cast<float4*>(band + t) = *reinterpret_cast<const float4*>(row_in + t);
        } else {
          for (int e = 0; e < 4; ++e)
            if (t + e < count) band[t + e] = row_in[t + e];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < kBandPerThread; ++j) {
        const int t = j * kBandThreads + tid;
        if (t < count) band[t] = row_in[t];
      }
    }
    __syncthreads();
    float* __restrict__ row_out = out + r * out_stride;
#pragma unroll
    for (int j = 0; j < kBandPerThread; ++j)
      if (dest[j] >= 0) row_out[dest[j]] = band[src[j]];
  }
}

template <typename TIn, typename TOut>
int launch_transpose(int batches, int rows, int cols, const void* in_v, int64_t in_batch_stride,
                     void* out_v, int64_t out_batch_stride, hipStream_t stream) {
  const TIn* in = static_cast<const TIn*>(in_v);
  TOut* out = static_cast<TOut*>(out_v);
  const int tiles_r = ceil_div(rows, kTile), tiles_c = ceil_div(cols, kTile);
  if (static_cast<int64_t>(tiles_r) * tiles_c > 0x7fffffff) return SPUTNIK_HIP_INVALID_ARGUMENT;
  const bool vec = rows % 4 == 0 && cols % 4 == 0 && aligned_to(in, 4 * sizeof(TIn)) &&
                   aligned_to(out, 4 * sizeof(TOut)) && in_batch_stride % 4 == 0 &&
                   out_batch_stride % 4 == 0;
  for (int b0 = 0; b0 < batches; b0 += kMaxGridYZ) {
    const int by = min(batches - b0, kMaxGridYZ);
    const dim3 grid(tiles_r * tiles_c, by);
    const TIn* in_b = in + b0 * in_batch_stride;
    TOut* out_b = out + b0 * out_batch_stride;
    if (vec)
      hipLaunchKernelGGL((transpose_tiles_kernel<TIn, TOut, 4>), grid, dim3(kThreads), 0, stream,
                         rows, cols, tiles_c, in_b, in_batch_stride, out_b, out_batch_stride);
    else
      hipLaunchKernelGGL((transpose_tiles_kernel<TIn, TOut, 1>), grid, dim3(kThreads), 0, stream,
                         rows, cols, tiles_c, in_b, in_batch_stride, out_b, out_batch_stride);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

}  // namespace
}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

int sputnik_hip_transpose_batched(int batches, int rows, int cols, const float* in,
                                  int64_t in_batch_stride, float* out, int64_t out_batch_stride,
                                  sputnik_hip_stream_t stream) {
  return sputnik_hip_transpose_cast_batched(batches, rows, cols, in, SPUTNIK_HIP_F32,
                                            in_batch_stride, out, SPUTNIK_HIP_F32,
                                            out_batch_stride, stream);
}

int sputnik_hip_permute_last_batched(int n, int rows, const float* in, int64_t in_stride,
                                     const int* permutation, float* out, int64_t out_stride,
                                     sputnik_hip_stream_t stream) {
  if (n < 0 || rows < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (n == 0 || rows == 0) return 0;
  const int by = ceil_div(rows, kGatherRows);
  if (by > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  hipLaunchKernelGGL(permute_last_kernel, dim3(ceil_div(n, kThreads), by), dim3(kThreads), 0,
                     stream, n, rows, in, in_stride, permutation, out, out_stride);
  return launch_status();
}

int sputnik_hip_permute_band_size(void) { return kPermuteBand; }

int sputnik_hip_permute_banded_batched(int n, int rows, const float* in, int64_t in_stride,
                                       const int* dest_list, const int* source_in_band,
                                       float* out, int64_t out_stride,
                                       sputnik_hip_stream_t stream) {
  if (n < 0 || rows < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (n == 0 || rows == 0) return 0;
  const int by = ceil_div(rows, kRowsPerGroup);
  if (by > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  const dim3 grid(ceil_div(n, kPermuteBand), by);
  // (a band starts at a multiple of 16384 entries: rows aligned alike are enough)
  if (aligned_to(in, 16) && in_stride % 4 == 0)
    hipLaunchKernelGGL(permute_banded_kernel<true>, grid, dim3(kBandThreads), 0, stream, n, rows,
                       in, in_stride, dest_list, source_in_band, out, out_stride);
  else
    hipLaunchKernelGGL(permute_banded_kernel<false>, grid, dim3(kBandThreads), 0, stream, n, rows,
                       in, in_stride, dest_list, source_in_band, out, out_stride);
  return launch_status();
}

int sputnik_hip_transpose_cast_batched(int batches, int rows, int cols, const void* in,
                                       int in_type, int64_t in_batch_stride, void* out,
                                       int out_type, int64_t out_batch_stride,
                                       sputnik_hip_stream_t stream) {
  if (batches < 0 || rows < 0 || cols < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (batches == 0 || rows == 0 || cols == 0) return 0;
#define SPUTNIK_HIP_TRANSPOSE_CASE(A, TA, B, TB)                                                \
  if (in_type == A && out_type == B)                                                            \
    return launch_transpose<TA, TB>(batches, rows, cols, in, in_batch_stride, out,              \
                                    out_batch_stride, stream)
  SPUTNIK_HIP_TRANSPOSE_CASE(SPUTNIK_HIP_F32, float, SPUTNIK_HIP_F32, float);
  SPUTNIK_HIP_TRANSPOSE_CASE(SPUTNIK_HIP_F16, __half, SPUTNIK_HIP_F32, float);
  SPUTNIK_HIP_TRANSPOSE_CASE(SPUTNIK_HIP_BF16, __hip_bfloat16, SPUTNIK_HIP_F32, float);
  SPUTNIK_HIP_TRANSPOSE_CASE(SPUTNIK_HIP_F32, float, SPUTNIK_HIP_F16, __half);
  SPUTNIK_HIP_TRANSPOSE_CASE(SPUTNIK_HIP_F32, float, SPUTNIK_HIP_BF16, __hip_bfloat16);
  SPUTNIK_HIP_TRANSPOSE_CASE(SPUTNIK_HIP_F16, __half, SPUTNIK_HIP_F16, __half);
  SPUTNIK_HIP_TRANSPOSE_CASE(SPUTNIK_HIP_BF16, __hip_bfloat16, SPUTNIK_HIP_BF16, __hip_bfloat16);
#undef SPUTNIK_HIP_TRANSPOSE_CASE
  return SPUTNIK_HIP_INVALID_ARGUMENT;
}

}  // extern "C"
