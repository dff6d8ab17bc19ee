// CSR transpose  CSR(m x n) -> CSR(n x m)  for gfx950: a stable counting sort
// on the column index, written by hand (no rocSPARSE / rocPRIM).
//
// Replaces cusparseCsr2cscEx2(..., CSR2CSC_ALG1, ...) as driven by
// src/transpose_cuda.cu:22-31,90-99.  "Stable" = inside every output row the
// source row ids ascend, which is the order ALG1 produces and what makes
// values_t line up with column_indices_t deterministically.
//
// The slot of a nonzero (row r, column c) in the output is
//     row_offsets_t[c] + #{nonzeros (r', c) with r' < r}.
// The second term is split by chunks of R (8..32) consecutive source rows:
// whole earlier chunks come from a [chunks][n] count table (scanned down the
// chunks), and inside a chunk the rank is a popcount: a workgroup builds, in
// LDS, one 32-bit mask per column with bit r-r0 set iff row r holds that
// column (a CSR row holds a column at most once), so
//     rank inside chunk = popc(mask[c] & ((1 << (r - r0)) - 1)).
// All nonzeros of a chunk are therefore independent: no ordered walk, no
// returning atomics -- every phase is flat data-parallel work.
//   1. count    masks in LDS (ds_or_b32), table[chunk][c] = popc(mask[c])
//   2. scan     table[:, c] made exclusive down the chunks; column totals
//               scanned into out_row_offsets
//   3. scatter  masks rebuilt in LDS, every nonzero computes its slot and
//               moves (value(s), row id, optionally its source index)
// Matrices with more than 8192 columns are processed in column ranges of 8192
// (64 KiB of masks + slot bases) by separate workgroups.  R is chosen so that
// a few hundred workgroups exist (m/R >= 256 where m allows).
// HBM traffic: 16 B per nonzero (values and indices read and written once)
// + 12 B per nonzero of repeated column-index reads + the small table.
#include "common.h"
#include "wave_utils.h"

namespace sputnik_hip {
namespace {

constexpr int kMaxRowsPerChunk = 32;  // one mask bit per row
constexpr int kColsPerRange = 8192;   // 32 KiB of masks + 32 KiB of slot bases
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kScanGroups = 16;  // chunk groups per column in the table scan

typedef unsigned int mask_t;

// Builds mask[c - c0] for the columns [c0, c1) of the rows [row0, row1).
// One wave per row at a time, lanes over the row's nonzeros (coalesced).
__device__ __forceinline__ void build_masks(mask_t* __restrict__ masks, int row0, int row1, int c0,
                                            int c1, const int* __restrict__ row_offsets,
                                            const int* __restrict__ column_indices) {
  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;
  for (int c = threadIdx.x; c < c1 - c0; c += kBlock) masks[c] = 0;
  __syncthreads();
  for (int row = row0 + wave; row < row1; row += kWaves) {
    const mask_t bit = mask_t{1} << (row - row0);
    const int p1 = row_offsets[row + 1];
    for (int p = row_offsets[row] + lane; p < p1; p += kWave) {
      const int c = column_indices[p];
      if (c >= c0 && c < c1) atomicOr(&masks[c - c0], bit);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void transpose_count_kernel(
    int m, int n, int rows_per_chunk, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, int* __restrict__ table) {
  extern __shared__ mask_t masks[];
  const int chunk = blockIdx.x;
  const int c0 = blockIdx.y * kColsPerRange, c1 = min(n, c0 + kColsPerRange);
  const int row0 = chunk * rows_per_chunk, row1 = min(m, row0 + rows_per_chunk);
  build_masks(masks, row0, row1, c0, c1, row_offsets, column_indices);
  int* __restrict__ my_table = table + static_cast<int64_t>(chunk) * n + c0;
  for (int c = threadIdx.x; c < c1 - c0; c += kBlock) my_table[c] = __popc(masks[c]);
}

// table[:, c] -> exclusive prefix down the chunks; totals[c] = column count.
// Block = 64 columns x kScanGroups chunk groups.
__global__ __launch_bounds__(kWave* kScanGroups) void transpose_scan_table_kernel(
    int n, int chunks, int* __restrict__ table, int* __restrict__ totals) {
  __shared__ int group_sum[kScanGroups][kWave];
  const int lane = threadIdx.x % kWave;
  const int group = threadIdx.x / kWave;
  const int c = blockIdx.x * kWave + lane;
  const int per_group = (chunks + kScanGroups - 1) / kScanGroups;
  const int ch0 = min(chunks, group * per_group);
  const int ch1 = min(chunks, ch0 + per_group);

  int sum = 0;
  if (c < n) {
    for (int ch = ch0; ch < ch1; ++ch) sum += table[static_cast<int64_t>(ch) * n + c];
  }
  group_sum[group][lane] = sum;
  __syncthreads();
  int running = 0;
  for (int g = 0; g < group; ++g) running += group_sum[g][lane];
  if (c < n) {
    for (int ch = ch0; ch < ch1; ++ch) {
      const int64_t idx = static_cast<int64_t>(ch) * n + c;
      const int t = table[idx];
      table[idx] = running;
      running += t;
    }
    if (group == kScanGroups - 1) totals[c] = running;
  }
}

// Exclusive scan of totals[0..n) into offsets[0..n], single workgroup.
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void transpose_scan_totals_kernel(
    int n, const int* __restrict__ totals, int* __restrict__ offsets) {
  __shared__ int wave_sums[kScanBlock / kWave];
  const int tid = threadIdx.x;
  const int lane = tid % kWave;
  const int wave = tid / kWave;
  const int per = (n + kScanBlock - 1) / kScanBlock;
  const int c0 = min(n, tid * per);
  const int c1 = min(n, c0 + per);
  int mine = 0;
  for (int c = c0; c < c1; ++c) mine += totals[c];
  int incl = mine;  // inclusive scan of `mine` over the wave
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += up;
  }
  if (lane == kWave - 1) wave_sums[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_sums[w];
  int running = base + incl - mine;
  for (int c = c0; c < c1; ++c) {
    offsets[c] = running;
    running += totals[c];
  }
  if (tid == kScanBlock - 1) offsets[n] = running;
}

__global__ __launch_bounds__(kBlock) void transpose_scatter_kernel(
    int m, int n, int rows_per_chunk, int replicas, const float* __restrict__ values,
    int64_t values_stride,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const int* __restrict__ table, const int* __restrict__ out_row_offsets,
    float* __restrict__ out_values, int64_t out_values_stride,
    int* __restrict__ out_column_indices, int* __restrict__ out_permutation) {
  extern __shared__ mask_t masks[];
  const int chunk = blockIdx.x;
  const int c0 = blockIdx.y * kColsPerRange, c1 = min(n, c0 + kColsPerRange);
  const int row0 = chunk * rows_per_chunk, row1 = min(m, row0 + rows_per_chunk);
  build_masks(masks, row0, row1, c0, c1, row_offsets, column_indices);

  // Slot base of every column of the range for this chunk (coalesced reads,
  // instead of two dependent gathers per nonzero).
  int* __restrict__ base = reinterpret_cast<int*>(masks + (c1 - c0));
  const int* __restrict__ my_table = table + static_cast<int64_t>(chunk) * n + c0;
  for (int c = threadIdx.x; c < c1 - c0; c += kBlock) base[c] = out_row_offsets[c0 + c] + my_table[c];
  __syncthreads();

  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;
  for (int row = row0 + wave; row < row1; row += kWaves) {
    const mask_t below = (mask_t{1} << (row - row0)) - 1;
    const int p1 = row_offsets[row + 1];
    for (int p = row_offsets[row] + lane; p < p1; p += kWave) {
      const int c = column_indices[p];
      if (c < c0 || c >= c1) continue;
      const int pos = base[c - c0] + __popc(masks[c - c0] & below);
      out_column_indices[pos] = row;
      if (out_permutation != nullptr) out_permutation[pos] = p;
      for (int r = 0; r < replicas; ++r)
        out_values[r * out_values_stride + pos] = values[r * values_stride + p];
    }
  }
}

// Rows per chunk: 8, 16 or 32, the largest that still leaves >= 256 chunks.
inline int rows_per_chunk_of(int m) {
  int r = kMaxRowsPerChunk;
  while (r > 8 && m / r < 256) r /= 2;
  return r;
}
inline int chunks_of(int m) { return ceil_div(m, rows_per_chunk_of(m)); }

}  // namespace
}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

size_t sputnik_hip_csr_transpose_workspace_bytes(int m, int n, int nonzeros) {
  (void)nonzeros;
  if (m <= 0 || n <= 0) return 0;
  return sizeof(int) * (static_cast<size_t>(chunks_of(m)) * n + n);
}

int sputnik_hip_csr_transpose(int m, int n, int nonzeros, int replicas, const float* values,
                              int64_t values_stride, const int* row_offsets,
                              const int* column_indices, float* out_values,
                              int64_t out_values_stride, int* out_row_offsets,
                              int* out_column_indices, int* out_permutation, void* workspace,
                              size_t workspace_bytes, sputnik_hip_stream_t stream) {
  if (m < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (n == 0) {
    return static_cast<int>(hipMemsetAsync(out_row_offsets, 0, sizeof(int), stream));
  }
  if (m == 0 || nonzeros == 0) {
    return static_cast<int>(
        hipMemsetAsync(out_row_offsets, 0, sizeof(int) * (static_cast<size_t>(n) + 1), stream));
  }
  if (workspace == nullptr ||
      workspace_bytes < sputnik_hip_csr_transpose_workspace_bytes(m, n, nonzeros))
    return SPUTNIK_HIP_INVALID_ARGUMENT;

  const int rows_per_chunk = rows_per_chunk_of(m);
  const int chunks = chunks_of(m);
  const int ranges = ceil_div(n, kColsPerRange);
  if (ranges > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  int* table = static_cast<int*>(workspace);
  int* totals = table + static_cast<size_t>(chunks) * n;
  const size_t range_cols = static_cast<size_t>(min(n, kColsPerRange));
  const size_t lds_bytes = sizeof(mask_t) * range_cols;

  hipLaunchKernelGGL(transpose_count_kernel, dim3(chunks, ranges), dim3(kBlock), lds_bytes, stream,
                     m, n, rows_per_chunk, row_offsets, column_indices, table);
  int st = launch_status();
  if (st != 0) return st;
  hipLaunchKernelGGL(transpose_scan_table_kernel, dim3(ceil_div(n, kWave)),
                     dim3(kWave * kScanGroups), 0, stream, n, chunks, table, totals);
  st = launch_status();
  if (st != 0) return st;
  hipLaunchKernelGGL(transpose_scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, stream, n,
                     totals, out_row_offsets);
  st = launch_status();
  if (st != 0) return st;
  hipLaunchKernelGGL(transpose_scatter_kernel, dim3(chunks, ranges), dim3(kBlock),
                     lds_bytes + sizeof(int) * range_cols, stream, m, n, rows_per_chunk, replicas,
                     values, values_stride, row_offsets, column_indices,
                     table, out_row_offsets, out_values, out_values_stride, out_column_indices,
                     out_permutation);
  return launch_status();
}

}  // extern "C"
