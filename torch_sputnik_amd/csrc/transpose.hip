// CSR transpose  CSR(m x n) -> CSR(n x m)  for gfx950: a stable counting sort
// on the column index, written by hand (no rocSPARSE / rocPRIM).
//
// Replaces cusparseCsr2cscEx2(..., CSR2CSC_ALG1, ...) as driven by
// src/transpose_cuda.cu:22-31,90-99.  "Stable" = inside every output row the
// source row ids ascend, which is the order ALG1 produces and what makes
// values_t line up with column_indices_t deterministically.
//
// The source rows are cut into `chunks` contiguous row ranges; one wave owns
// one chunk and walks its rows IN ORDER, 64 nonzeros of one row at a time (a
// valid CSR row holds each column at most once, so the lanes of one step never
// meet on a counter and the order of two entries of one column is the order
// of their rows):
//   1. count    per-chunk histogram of column ids (counters in LDS)
//                 -> table[chunk][column]
//   2. scan     table[:, column] made exclusive down the chunks, column
//               totals scanned into out_row_offsets
//   3. scatter  every chunk reloads its base offsets into LDS and walks its
//               rows again, handing out slots with returning LDS adds.
// Workspace: chunks*n + n int32 (table + column totals).  HBM traffic is
// 16 B per nonzero (values and indices read and written once) + 8 B per
// nonzero for the second read of the column ids + the table passes.
#include "common.h"
#include "wave_utils.h"

namespace sputnik_hip {
namespace {

constexpr int kMaxChunks = 1024;
constexpr int kLdsColumns = 16384;  // 64 KiB of counters per wave
constexpr int kScanGroups = 16;     // chunk groups per column in the table scan

struct ChunkPlan {
  int chunks;
  int rows_per_chunk;
};

inline ChunkPlan plan_chunks(int m) {
  ChunkPlan p;
  p.rows_per_chunk = max(1, ceil_div(m, kMaxChunks));
  p.chunks = max(1, ceil_div(m, p.rows_per_chunk));
  return p;
}

// One wave per chunk.  LDS_COUNTERS: counters live in LDS (n <= kLdsColumns),
// otherwise in this chunk's row of the global table.
template <bool LDS_COUNTERS>
__global__ __launch_bounds__(kWave) void transpose_count_kernel(
    int m, int n, int rows_per_chunk, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, int* __restrict__ table) {
  extern __shared__ int lds_counters[];
  const int chunk = blockIdx.x;
  const int lane = threadIdx.x;
  int* __restrict__ my_table = table + static_cast<int64_t>(chunk) * n;
  int* counters = LDS_COUNTERS ? lds_counters : my_table;
  for (int c = lane; c < n; c += kWave) counters[c] = 0;
  if constexpr (LDS_COUNTERS) {
    __syncthreads();
  } else {
    __threadfence();
  }

  const int row0 = chunk * rows_per_chunk;
  const int row1 = min(m, row0 + rows_per_chunk);
  // Counting does not need the row order: sweep the chunk's nonzeros flat.
  const int p0 = row_offsets[row0];
  const int p1 = row_offsets[row1];
  for (int p = p0 + lane; p < p1; p += kWave) atomicAdd(&counters[column_indices[p]], 1);

  if constexpr (LDS_COUNTERS) {
    __syncthreads();
    for (int c = lane; c < n; c += kWave) my_table[c] = lds_counters[c];
  }
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
  // inclusive scan of `mine` over the wave
  int incl = mine;
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

template <bool LDS_COUNTERS>
__global__ __launch_bounds__(kWave) void transpose_scatter_kernel(
    int m, int n, int nonzeros, int replicas, int rows_per_chunk,
    const float* __restrict__ values, int64_t values_stride,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ table, const int* __restrict__ out_row_offsets,
    float* __restrict__ out_values, int64_t out_values_stride,
    int* __restrict__ out_column_indices, int* __restrict__ out_permutation) {
  extern __shared__ int lds_counters[];
  const int chunk = blockIdx.x;
  const int lane = threadIdx.x;
  int* __restrict__ my_table = table + static_cast<int64_t>(chunk) * n;
  int* counters = LDS_COUNTERS ? lds_counters : my_table;
  for (int c = lane; c < n; c += kWave) counters[c] = my_table[c] + out_row_offsets[c];
  if constexpr (LDS_COUNTERS) {
    __syncthreads();
  } else {
    __threadfence();
  }

  const int row0 = chunk * rows_per_chunk;
  const int row1 = min(m, row0 + rows_per_chunk);
  for (int row = row0; row < row1; ++row) {
    const int p0 = row_offsets[row];
    const int p1 = row_offsets[row + 1];
    for (int p = p0 + lane; p < p1; p += kWave) {
      const int pos = atomicAdd(&counters[column_indices[p]], 1);
      out_column_indices[pos] = row;
      if (out_permutation != nullptr) out_permutation[pos] = p;
      for (int r = 0; r < replicas; ++r)
        out_values[r * out_values_stride + pos] = values[r * values_stride + p];
    }
    // The next row's adds must come after this row's.  One wave issues its
    // LDS instructions in order and the LDS executes them in order, so the
    // LDS path needs nothing; the global-counter path drains its returning
    // atomics before it starts the next row.
    if constexpr (!LDS_COUNTERS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  (void)nonzeros;
}

}  // namespace
}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

size_t sputnik_hip_csr_transpose_workspace_bytes(int m, int n, int nonzeros) {
  (void)nonzeros;
  if (m <= 0 || n <= 0) return 0;
  const ChunkPlan plan = plan_chunks(m);
  return sizeof(int) * (static_cast<size_t>(plan.chunks) * n + n);
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

  const ChunkPlan plan = plan_chunks(m);
  int* table = static_cast<int*>(workspace);
  int* totals = table + static_cast<size_t>(plan.chunks) * n;
  const bool lds = n <= kLdsColumns;
  const size_t lds_bytes = lds ? sizeof(int) * static_cast<size_t>(n) : 0;

  if (lds) {
    hipLaunchKernelGGL(transpose_count_kernel<true>, dim3(plan.chunks), dim3(kWave), lds_bytes,
                       stream, m, n, plan.rows_per_chunk, row_offsets, column_indices, table);
  } else {
    hipLaunchKernelGGL(transpose_count_kernel<false>, dim3(plan.chunks), dim3(kWave), 0, stream,
                       m, n, plan.rows_per_chunk, row_offsets, column_indices, table);
  }
  int st = launch_status();
  if (st != 0) return st;

  hipLaunchKernelGGL(transpose_scan_table_kernel, dim3(ceil_div(n, kWave)),
                     dim3(kWave * kScanGroups), 0, stream, n, plan.chunks, table, totals);
  st = launch_status();
  if (st != 0) return st;

  hipLaunchKernelGGL(transpose_scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, stream, n,
                     totals, out_row_offsets);
  st = launch_status();
  if (st != 0) return st;

  if (lds) {
    hipLaunchKernelGGL(transpose_scatter_kernel<true>, dim3(plan.chunks), dim3(kWave), lds_bytes,
                       stream, m, n, nonzeros, replicas, plan.rows_per_chunk, values,
                       values_stride, row_offsets, column_indices, table, out_row_offsets,
                       out_values, out_values_stride, out_column_indices, out_permutation);
  } else {
    hipLaunchKernelGGL(transpose_scatter_kernel<false>, dim3(plan.chunks), dim3(kWave), 0,
                       stream, m, n, nonzeros, replicas, plan.rows_per_chunk, values,
                       values_stride, row_offsets, column_indices, table, out_row_offsets,
                       out_values, out_values_stride, out_column_indices, out_permutation);
  }
  return launch_status();
}

}  // extern "C"
