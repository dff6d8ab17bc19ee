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
// The second term is split by chunks of 32 consecutive source rows: whole
// earlier chunks come from a [chunks][n] count table (scanned down the chunks),
// and inside a chunk the rank is a popcount: one 32-bit mask per column with
// bit r-r0 set iff row r holds that column (a CSR row holds a column at most
// once -- a PRECONDITION, as for every CSR consumer of the reference: a row that
// stores a column twice would have both entries land on one slot), so
//     rank inside chunk = popc(mask[c] & ((1 << (r - r0)) - 1)).
// All nonzeros of a chunk are therefore independent: no ordered walk, no
// returning atomics -- every phase is flat data-parallel work.
//   1. masks    one workgroup per chunk builds the masks in LDS (ds_or_b32) and
//               writes them out: gmask[chunk][c]
//   2. scan     table[chunk][c] = number of entries of column c in earlier
//               chunks (popcounts scanned down the chunks), totals[c]
//   3. scatter  one workgroup per (chunk, group of 256 columns): masks and slot
//               bases of its columns in LDS, every nonzero of the chunk's rows
//               that falls into the group computes its slot; the entries are
//               first put in OUTPUT order in LDS and then written as runs, so
//               that a wave's store instruction covers a few contiguous runs
//               instead of 64 different cache lines (see the kernel).  Up to
//               8192 columns every workgroup derives the slot bases of its
//               columns from the column totals itself instead of waiting for a
//               fourth launch; wider matrices take that launch and a plain
//               scatter over column ranges of 8192.
// Three launches (four above 8192 columns).  HBM traffic: 16 B per nonzero
// (values and indices read and written once) + the mask / count tables (8 B
// per chunk and column).
// Workspace: two int tables of [ceil(m / 32)][n] + [n] totals.
//
// VERY SPARSE, VERY LARGE matrices (the tables would dwarf the nonzeros: 1 GiB at
// 65536^2 whatever nnz is) take a second path with O(n + nnz) work and workspace,
// as cusparseCsr2cscEx2 has (src/transpose_cuda.cu:22-31): column histogram
// (integer atomics) -> scan -> unordered scatter of (source row, source index)
// through per-column cursors -> every output row RANKS its few entries by source
// row (the stable order, whatever order the atomics came in: deterministic).
//
// Both paths DETECT a row that stores a column twice (the mask kernel: the set
// bits of a chunk's masks must number the chunk's entries; the ranking kernel: two
// equal source rows in one output row) and set a status word at the end of the
// workspace; sputnik_hip_csr_transpose_checked waits for it and returns
// SPUTNIK_HIP_INVALID_ARGUMENT, the asynchronous entry leaves it for the caller.
#include "common.h"
#include "wave_utils.h"

namespace sputnik_hip {
namespace {

constexpr int kRowsPerChunk = 32;    // one mask bit per row
constexpr int kColsPerRange = 8192;  // 32 KiB of masks + 32 KiB of slot bases
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kScanGroups = 16;  // chunk groups per column in the table scan
constexpr int kGroupCols = 256;  // columns per scatter workgroup (<= kColsPerRange columns)
constexpr int kLongColumn = 2048; // O(n + nnz) path: output rows beyond this are ranked through a bitmap


typedef unsigned int mask_t;

// "Many mask" batches in ONE launch per phase (round 4): the grid's last dimension is
// the mask.  Mask i reads the i-th topology of the concatenated arrays (common.h:
// row_offsets [masks][m + 1], column_indices back to back), moves the values of its
// `heads` replicas, and keeps its tables `region` ints behind those of mask i - 1.
// mask 0 -- and every single-topology call -- moves nothing.
struct ManyPlace {
  int64_t offsets;   // ints to add to row_offsets
  int first;         // entries in front of this mask's in column_indices / the outputs
};
__device__ __forceinline__ ManyPlace many_place(int mask, int m, const int* __restrict__ row_offsets) {
  int first = 0;
  for (int j = 0; j < mask; ++j) first += row_offsets[static_cast<int64_t>(j) * (m + 1) + m];
  return ManyPlace{static_cast<int64_t>(mask) * (m + 1), first};
}

// Column groups of a matrix with n <= kColsPerRange columns: a multiple of 8, so
// that group g of every chunk has the same `workgroup id % 8` (see the scatter).
inline int column_groups(int n) { return ceil_div(ceil_div(n, kGroupCols), 8) * 8; }

// A wave walks one row: lanes over the row's nonzeros (coalesced), kUnroll
// requests per lane in flight before the first one is used -- a row of a few
// hundred entries costs ONE memory latency, not one per 64 entries.
constexpr int kUnroll = 8;

// gmask[chunk][c] for the columns [c0, c1) of one chunk of rows.  16 waves, two
// rows each; a wave requests the bounds of both rows, then the first kUnroll x 64
// entries of both, before it touches the first answer: the kernel is a chain of
// memory latencies, not of bytes.
constexpr int kMaskBlock = 1024;
constexpr int kRowsPerWave = kRowsPerChunk / (kMaskBlock / kWave);  // 2
__global__ __launch_bounds__(kMaskBlock) void transpose_mask_kernel(
    int m, int n, const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    mask_t* __restrict__ gmask, int* __restrict__ balances, int64_t region) {
  extern __shared__ mask_t masks[];
  if (blockIdx.z > 0) {
    const ManyPlace place = many_place(blockIdx.z, m, row_offsets);
    row_offsets += place.offsets;
    column_indices += place.first;
    gmask += blockIdx.z * region;
    balances += blockIdx.z * region;
  }
  __shared__ int balance;   // entries seen minus bits found set: 0 unless a row repeats a column or holds one outside [0, n)
  if (threadIdx.x == 0) balance = 0;
  int entries = 0;
  const int chunk = blockIdx.x;
  const int c0 = blockIdx.y * kColsPerRange, c1 = min(n, c0 + kColsPerRange);
  const int row0 = chunk * kRowsPerChunk;
  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;
  int p0[kRowsPerWave], p1[kRowsPerWave];
#pragma unroll
  for (int j = 0; j < kRowsPerWave; ++j) {
    const int row = row0 + wave + j * (kMaskBlock / kWave);
    p0[j] = row < m ? row_offsets[row] : 0;
    p1[j] = row < m ? row_offsets[row + 1] : 0;
  }
  int c[kRowsPerWave][kUnroll];
#pragma unroll
  for (int j = 0; j < kRowsPerWave; ++j)
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int p = p0[j] + lane + u * kWave;
      c[j][u] = p < p1[j] ? column_indices[p] : -1;
    }
  for (int i = threadIdx.x; i < c1 - c0; i += kMaskBlock) masks[i] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRowsPerWave; ++j) {
    const mask_t bit = mask_t{1} << (wave + j * (kMaskBlock / kWave));
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
      if (c[j][u] >= c0 && c[j][u] < c1) {
        atomicOr(&masks[c[j][u] - c0], bit);
        ++entries;
      } else if (blockIdx.y == 0 && (c[j][u] < 0 || c[j][u] >= n) &&
                 p0[j] + lane + u * kWave < p1[j]) {
        ++entries;   // a column outside [0, n): no mask takes it, so the chunk does not balance
      }
    // rows longer than kUnroll x 64 entries: the rest, one batch at a time
    for (int first = p0[j] + lane + kUnroll * kWave; first < p1[j]; first += kUnroll * kWave) {
      int more[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int p = first + u * kWave;
        more[u] = p < p1[j] ? column_indices[p] : -1;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
        if (more[u] >= c0 && more[u] < c1) {
          atomicOr(&masks[more[u] - c0], bit);
          ++entries;
        } else if (blockIdx.y == 0 && (more[u] < 0 || more[u] >= n) && first + u * kWave < p1[j]) {
          ++entries;
        }
    }
  }
  __syncthreads();
  mask_t* __restrict__ mine = gmask + static_cast<int64_t>(chunk) * n + c0;
  int bits = 0;
  for (int i = threadIdx.x; i < c1 - c0; i += kMaskBlock) {
    mine[i] = masks[i];
    bits += __popc(masks[i]);
  }
  int diff = entries - bits;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) diff += __shfl_xor(diff, o);
  if (lane == 0 && diff != 0) atomicAdd(&balance, diff);
  __syncthreads();
  // (every workgroup writes its own slot: the table needs no initialisation; the scan
  // kernel's first workgroup folds the slots into the status word)
  if (threadIdx.x == 0) balances[blockIdx.y * gridDim.x + blockIdx.x] = balance;
}

// ---------------------------------------------------------------------------
// O(n + nnz) path (see the file header).
// ---------------------------------------------------------------------------
// one wave per row: counts[c] += 1 for each of its entries (unordered = false), or
// (unordered = true) the entry takes the next slot of its column's cursor
template <bool SCATTER>
__global__ __launch_bounds__(kBlock) void transpose_sparse_rows_kernel(
    int m, int n, const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ counts_or_cursor, int* __restrict__ tmp_row, int* __restrict__ tmp_src,
    int* __restrict__ status) {
  const int lane = threadIdx.x % kWave;
  const int row = blockIdx.x * kWaves + threadIdx.x / kWave;
  if (row >= m) return;
  const int p1 = row_offsets[row + 1];
  for (int p = row_offsets[row] + lane; p < p1; p += kWave) {
    const int c = column_indices[p];
    if (c < 0 || c >= n) {
      atomicOr(status, 2);   // column out of range
      continue;
    }
    if constexpr (SCATTER) {
      const int slot = atomicAdd(&counts_or_cursor[c], 1);
      tmp_row[slot] = row;
      tmp_src[slot] = p;
    } else {
      atomicAdd(&counts_or_cursor[c], 1);
    }
  }
}

// out_row_offsets = exclusive scan of counts (n + 1 values); cursor = the same
__global__ __launch_bounds__(1024) void transpose_sparse_scan_kernel(
    int n, const int* __restrict__ counts, int* __restrict__ out_row_offsets,
    int* __restrict__ cursor) {
  __shared__ int partial[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int i0 = min(t * per, n), i1 = min(i0 + per, n);
  int sum = 0;
  for (int i = i0; i < i1; ++i) sum += counts[i];
  partial[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off *= 2) {
    const int v = t >= off ? partial[t - off] : 0;
    __syncthreads();
    partial[t] += v;
    __syncthreads();
  }
  int run = partial[t] - sum;
  for (int i = i0; i < i1; ++i) {
    out_row_offsets[i] = run;
    cursor[i] = run;
    run += counts[i];
  }
  if (t == 1023) out_row_offsets[n] = partial[1023];
}

// one wave per output row: every entry of the row's (unordered) segment finds its
// rank = number of entries with a smaller source row, and goes there
__global__ __launch_bounds__(kBlock) void transpose_sparse_rank_kernel(
    int n, const int* __restrict__ out_row_offsets, const int* __restrict__ tmp_row,
    const int* __restrict__ tmp_src, int* __restrict__ out_column_indices,
    int* __restrict__ permutation, int* __restrict__ status) {
  const int lane = threadIdx.x % kWave;
  const int c = blockIdx.x * kWaves + threadIdx.x / kWave;
  if (c >= n) return;
  const int o0 = out_row_offsets[c], o1 = out_row_offsets[c + 1];
  if (o1 - o0 > kLongColumn) return;   // transpose_sparse_rank_long_kernel's
  for (int i = o0 + lane; i < o1; i += kWave) {
    const int mine = tmp_row[i];
    int rank = 0, same = 0;
    for (int j = o0; j < o1; ++j) {
      const int other = tmp_row[j];
      rank += other < mine;
      same += other == mine;
    }
    if (same != 1) {
      atomicOr(status, 1);   // a row holds this column twice: no stable order exists
      continue;
    }
    out_column_indices[o0 + rank] = mine;
    permutation[o0 + rank] = tmp_src[i];
  }
}

// Output rows with more than kLongColumn entries (a "global token" column of a
// very sparse 65536^2 attention pattern holds 65536 -- all-pairs ranking by one wave would
// be 6.7e7 serial steps).  One workgroup per such row: a bitmap of the source rows in LDS
// (kBitmapWords x 32 rows per pass), its prefix popcounts, and rank = set bits below the
// entry's own -- O(m / 32 + entries) per pass.  A bit found set already = the row holds
// the column twice.
constexpr int kBitmapWords = 8192;   // 32 KiB bitmap + 32 KiB prefix: 262 144 source rows per pass
__global__ __launch_bounds__(kBlock) void transpose_sparse_rank_long_kernel(
    int m, const int* __restrict__ out_row_offsets, const int* __restrict__ tmp_row,
    const int* __restrict__ tmp_src, int* __restrict__ out_column_indices,
    int* __restrict__ permutation, int* __restrict__ status) {
  __shared__ unsigned bitmap[kBitmapWords];
  __shared__ int prefix[kBitmapWords];
  __shared__ int partial[kBlock];
  const int c = blockIdx.x, t = threadIdx.x;
  const int o0 = out_row_offsets[c], o1 = out_row_offsets[c + 1];
  if (o1 - o0 <= kLongColumn) return;
  constexpr int kPer = kBitmapWords / kBlock;   // words per thread in the scan
  int ranked = 0;                               // entries of earlier passes
  for (int r0 = 0; r0 < m; r0 += kBitmapWords * 32) {
    for (int w = t; w < kBitmapWords; w += kBlock) bitmap[w] = 0u;
    __syncthreads();
    for (int i = o0 + t; i < o1; i += kBlock) {
      const unsigned rel = static_cast<unsigned>(tmp_row[i] - r0);
      if (rel < static_cast<unsigned>(kBitmapWords * 32)) {
        const unsigned bit = 1u << (rel & 31);
        if (atomicOr(&bitmap[rel >> 5], bit) & bit) atomicOr(status, 1);
      }
    }
    __syncthreads();
    int sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) sum += __popc(bitmap[t * kPer + j]);
    partial[t] = sum;
    __syncthreads();
    for (int off = 1; off < kBlock; off *= 2) {
      const int v = t >= off ? partial[t - off] : 0;
      __syncthreads();
      partial[t] += v;
      __syncthreads();
    }
    int run = partial[t] - sum;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      prefix[t * kPer + j] = run;
      run += __popc(bitmap[t * kPer + j]);
    }
    const int pass_total = partial[kBlock - 1];
    __syncthreads();
    for (int i = o0 + t; i < o1; i += kBlock) {
      const int row = tmp_row[i];
      const unsigned rel = static_cast<unsigned>(row - r0);
      if (rel < static_cast<unsigned>(kBitmapWords * 32)) {
        const int rank = ranked + prefix[rel >> 5] + __popc(bitmap[rel >> 5] & ((1u << (rel & 31)) - 1u));
        out_column_indices[o0 + rank] = row;
        permutation[o0 + rank] = tmp_src[i];
      }
    }
    ranked += pass_total;
    __syncthreads();
  }
}

// out_values[r][i] = values[r][permutation[i]], widened to float if T is a half type.
template <typename T = float>
__global__ __launch_bounds__(kBlock) void transpose_sparse_values_kernel(
    int nonzeros, int replicas, const T* __restrict__ values, int64_t values_stride,
    const int* __restrict__ permutation, float* __restrict__ out_values,
    int64_t out_values_stride) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nonzeros) return;
  const int p = permutation[i];
  for (int r = 0; r < replicas; ++r)
    out_values[r * out_values_stride + i] = static_cast<float>(values[r * values_stride + p]);
}

// The same for a "many mask" batch: grid (pieces of 4 x kBlock entries, heads, masks);
// replica mask * heads + head reads its row through the mask's part of the permutation.
constexpr int kManyValuesUnroll = 4;
__global__ __launch_bounds__(kBlock) void transpose_many_values_kernel(
    int m, int heads, const float* __restrict__ values, int64_t values_stride,
    const int* __restrict__ row_offsets, const int* __restrict__ permutation,
    float* __restrict__ out_values, int64_t out_values_stride) {
  // Workgroups are dealt to the 8 XCDs round-robin in launch order and every XCD has its
  // own L2: the pieces of one replica's row -- 4-byte reads all over up to 2 MiB -- go to
  // ONE XCD.  In launch order
  // every XCD fetched every row for itself: 88 us for 57 MB at 8 masks x 8 heads of 1024^2.
  const unsigned total = gridDim.x * gridDim.y * gridDim.z;
  const unsigned launch = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const unsigned work = total % 8 == 0 ? (launch % 8) * (total / 8) + launch / 8 : launch;
  // (masks before heads: the masks differ in size, an XCD that took whole masks would take
  // 10 x the work of another -- 142 us)
  const int piece = work % gridDim.x;
  const int mask = (work / gridDim.x) % gridDim.z;
  const int head = work / (gridDim.x * gridDim.z);
  const ManyPlace place = many_place(mask, m, row_offsets);
  const int count = row_offsets[place.offsets + m];
  const int64_t replica = static_cast<int64_t>(mask) * heads + head;
  values += replica * values_stride;
  out_values += replica * out_values_stride;
  permutation += place.first;
  const int i0 = piece * (kBlock * kManyValuesUnroll) + threadIdx.x;
  int p[kManyValuesUnroll];
#pragma unroll
  for (int u = 0; u < kManyValuesUnroll; ++u) p[u] = i0 + u * kBlock < count ? permutation[i0 + u * kBlock] : 0;
  float v[kManyValuesUnroll];
#pragma unroll
  for (int u = 0; u < kManyValuesUnroll; ++u) v[u] = i0 + u * kBlock < count ? values[p[u]] : 0.f;
#pragma unroll
  for (int u = 0; u < kManyValuesUnroll; ++u)
    if (i0 + u * kBlock < count) out_values[i0 + u * kBlock] = v[u];
}

// table[chunk][c] = sum of popc(gmask[chunk'][c]) over chunk' < chunk;
// totals[c] = column count.  Block = 64 columns x kScanGroups chunk groups.
__global__ __launch_bounds__(kWave* kScanGroups) void transpose_scan_table_kernel(
    int n, int chunks, const mask_t* __restrict__ gmask, int* __restrict__ table,
    int* __restrict__ totals, const int* __restrict__ balances, int nbalances,
    int* __restrict__ status, int64_t region) {
  __shared__ int group_sum[kScanGroups][kWave];
  if (blockIdx.y > 0) {
    const int64_t move = blockIdx.y * region;
    gmask += move;
    table += move;
    totals += move;
    balances += move;
    status += move;
  }
  if (blockIdx.x == 0) {   // status word: a non-zero balance = a row holds a column twice
    int bad = 0;
    for (int i = threadIdx.x; i < nbalances; i += kWave * kScanGroups) bad |= balances[i] != 0;
    const int any = __syncthreads_or(bad);
    if (threadIdx.x == 0) *status = any ? 1 : 0;
  }
  const int lane = threadIdx.x % kWave;
  const int group = threadIdx.x / kWave;
  const int c = blockIdx.x * kWave + lane;
  const int per_group = (chunks + kScanGroups - 1) / kScanGroups;
  const int ch0 = min(chunks, group * per_group);
  const int ch1 = min(chunks, ch0 + per_group);

  // up to kKeep chunks per group (m <= 4096 rows) stay in registers between the
  // two passes: one memory round trip instead of two
  constexpr int kKeep = 8;
  int kept[kKeep];
  int sum = 0;
  if (c < n) {
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
      const int ch = ch0 + i;
      kept[i] = ch < ch1 ? __popc(gmask[static_cast<int64_t>(ch) * n + c]) : 0;
      sum += kept[i];
    }
    for (int ch = ch0 + kKeep; ch < ch1; ++ch) sum += __popc(gmask[static_cast<int64_t>(ch) * n + c]);
  }
  group_sum[group][lane] = sum;
  __syncthreads();
  int running = 0;
  for (int g = 0; g < group; ++g) running += group_sum[g][lane];
  if (c < n) {
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
      const int ch = ch0 + i;
      if (ch < ch1) table[static_cast<int64_t>(ch) * n + c] = running;
      running += kept[i];
    }
    for (int ch = ch0 + kKeep; ch < ch1; ++ch) {
      const int64_t idx = static_cast<int64_t>(ch) * n + c;
      table[idx] = running;
      running += __popc(gmask[idx]);
    }
    if (group == kScanGroups - 1) totals[c] = running;
  }
}

// Exclusive scan of totals[0..n) into offsets[0..n], single workgroup
// (matrices wider than kColsPerRange only).
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

// Plain scatter over column ranges (matrices wider than kColsPerRange):
// `parts` workgroups per chunk, a few rows each; out_row_offsets comes from the
// scan kernel above.
__global__ __launch_bounds__(kBlock) void transpose_scatter_kernel(
    int m, int n, int parts, int rows_per_part, int replicas, const float* __restrict__ values,
    int64_t values_stride, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, const mask_t* __restrict__ gmask,
    const int* __restrict__ table, const int* __restrict__ out_row_offsets,
    float* __restrict__ out_values, int64_t out_values_stride,
    int* __restrict__ out_column_indices, int* __restrict__ out_permutation) {
  extern __shared__ mask_t masks[];
  const int chunk = blockIdx.x / parts, part = blockIdx.x % parts;
  const int c0 = blockIdx.y * kColsPerRange, c1 = min(n, c0 + kColsPerRange);
  const int width = c1 - c0;
  const int chunk_row0 = chunk * kRowsPerChunk;
  const int row0 = chunk_row0 + part * rows_per_part;
  const int row1 = min(min(m, chunk_row0 + kRowsPerChunk), row0 + rows_per_part);
  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;

  int* __restrict__ base = reinterpret_cast<int*>(masks + width);
  const mask_t* __restrict__ my_mask = gmask + static_cast<int64_t>(chunk) * n + c0;
  const int* __restrict__ my_table = table + static_cast<int64_t>(chunk) * n + c0;
  for (int c = threadIdx.x; c < width; c += kBlock) {
    masks[c] = my_mask[c];
    base[c] = out_row_offsets[c0 + c] + my_table[c];
  }
  __syncthreads();

  for (int row = row0 + wave; row < row1; row += kWaves) {
    const mask_t below = (mask_t{1} << (row - chunk_row0)) - 1;
    const int p1 = row_offsets[row + 1];
    for (int first = row_offsets[row] + lane; first < p1; first += kWave * kUnroll) {
      int c[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int p = first + u * kWave;
        c[u] = p < p1 ? column_indices[p] : -1;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (c[u] < c0 || c[u] >= c1) continue;
        const int p = first + u * kWave;
        const int pos = base[c[u] - c0] + __popc(masks[c[u] - c0] & below);
        out_column_indices[pos] = row;
        if (out_permutation != nullptr) out_permutation[pos] = p;
        for (int r = 0; r < replicas; ++r)
          out_values[r * out_values_stride + pos] = values[r * values_stride + p];
      }
    }
  }
}

// Exclusive scan over the first kGroupCols threads' values (one per thread; the
// other threads pass 0); returns the prefix of the caller, *total = the sum.
// Contains two workgroup barriers.
constexpr int kGroupBlock = 512;  // 8 waves: four rows of the chunk each
__device__ __forceinline__ int scan_group_columns(int value, int* __restrict__ wave_sums,
                                                  int* __restrict__ total) {
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  int incl = value;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += up;
  }
  __syncthreads();   // wave_sums may still be read from a previous scan
  if (lane == kWave - 1) wave_sums[wave] = incl;
  __syncthreads();
  int before = 0, all = 0;
  for (int w = 0; w < kGroupCols / kWave; ++w) {
    if (w < wave) before += wave_sums[w];
    all += wave_sums[w];
  }
  *total = all;
  return before + incl - value;
}

// Scatter for matrices of up to kColsPerRange columns: workgroup b handles chunk
// b / groups and the kGroupCols columns of group b % groups.  It reads all of its
// chunk's entries (the reads of the `groups` workgroups of a chunk overlap in L2)
// and moves those that fall into its columns.
//
// Why by column group: an output cache line (32 consecutive entries of one
// output row) collects entries from many source rows.  With the writers of a
// line spread over the chip, every XCD's L2 ends up holding a few bytes of
// every line and writes them back one partial line at a time (measured at
// 2048^2, density 0.2: 20.7 us for 6.7 MB of stores).  Workgroups are dealt
// round-robin to the 8 XCDs and `groups` is a multiple of 8, so all writers of an
// output line sit behind ONE L2, which merges them into whole lines (16 us;
// 12.7 us with the staging below).
// (Placement is a speed matter only: nothing depends on it for correctness.)
//
// Why staged: in source order the 64 lanes of a store instruction hit 64
// different output rows -- 64 requests of 4 bytes to L2.  The workgroup
// therefore first puts its entries in OUTPUT order in LDS (local slot = the
// same rank arithmetic with local bases) and then writes them flat: consecutive
// lanes then hold consecutive slots of an output row (a run of up to 32
// entries per column and chunk), and an instruction covers a few contiguous
// runs.  The staging area holds kStageCap entries (a chunk x group that is more
// than half full writes the overflow directly).
//
// Latency: everything the workgroup needs that does not depend on another load
// (column totals, masks, counts, the bounds of its rows) is requested at the
// top; a wave then walks its four rows with kUnroll x 64 entries in flight.
constexpr int kStageCap = 4096;
__global__ __launch_bounds__(kGroupBlock) void transpose_scatter_grouped_kernel(
    int m, int n, int groups, int replicas, const float* __restrict__ values,
    int64_t values_stride, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, const mask_t* __restrict__ gmask,
    const int* __restrict__ table, const int* __restrict__ totals,
    int* __restrict__ out_row_offsets, float* __restrict__ out_values,
    int64_t out_values_stride, int* __restrict__ out_column_indices,
    int* __restrict__ out_permutation, int64_t region) {
  if (blockIdx.y > 0) {   // (many masks: `replicas` is the heads of one mask)
    const ManyPlace place = many_place(blockIdx.y, m, row_offsets);
    const int64_t move = blockIdx.y * region;
    row_offsets += place.offsets;
    column_indices += place.first;
    gmask += move;
    table += move;
    totals += move;
    out_row_offsets += static_cast<int64_t>(blockIdx.y) * (n + 1);
    out_column_indices += place.first;
    if (out_permutation != nullptr) out_permutation += place.first;
    if (values != nullptr) values += static_cast<int64_t>(blockIdx.y) * replicas * values_stride;
    if (out_values != nullptr) out_values += static_cast<int64_t>(blockIdx.y) * replicas * out_values_stride;
  }
  __shared__ mask_t masks[kGroupCols];
  __shared__ int base[kGroupCols];       // global slot of the first entry (this chunk, column)
  __shared__ int lbase[kGroupCols];      // the same in the staging area
  __shared__ int wave_sums[kGroupBlock / kWave];
  __shared__ int st_pos[kStageCap], st_src[kStageCap];
  __shared__ float st_val[kStageCap];
  __shared__ unsigned char st_row[kStageCap];

  const int chunk = blockIdx.x / groups, cg = blockIdx.x % groups;
  const int c0 = cg * kGroupCols, c1 = min(n, c0 + kGroupCols);
  const int width = c1 - c0;
  if (width <= 0) return;
  const int row0 = chunk * kRowsPerChunk;
  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;
  const int t = threadIdx.x;
  constexpr int kRows = kRowsPerChunk / (kGroupBlock / kWave);  // rows per wave: 2
  const bool stage_values = replicas == 1;

  // ---- independent requests first
  int p0[kRows], p1[kRows];
#pragma unroll
  for (int j = 0; j < kRows; ++j) {
    const int row = row0 + wave + j * (kGroupBlock / kWave);
    p0[j] = row < m ? row_offsets[row] : 0;
    p1[j] = row < m ? row_offsets[row + 1] : 0;
  }
  int before = 0;
  for (int c = t; c < c0; c += kGroupBlock) before += totals[c];
  const bool mine = t < width;
  const int my_total = mine ? totals[c0 + t] : 0;
  const mask_t my_mask = mine ? gmask[static_cast<int64_t>(chunk) * n + c0 + t] : 0;
  const int my_count = mine ? table[static_cast<int64_t>(chunk) * n + c0 + t] : 0;
  // slot base of column c = sum of totals[0..c) + entries of c in earlier chunks
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) before += __shfl_xor(before, off, kWave);
  if (lane == 0) wave_sums[wave] = before;
  __syncthreads();
  before = 0;
  for (int w = 0; w < kGroupBlock / kWave; ++w) before += wave_sums[w];
  int sum_totals, sum_local;
  const int prefix = scan_group_columns(my_total, wave_sums, &sum_totals);
  const int lprefix = scan_group_columns(__popc(my_mask), wave_sums, &sum_local);
  if (mine) {
    masks[t] = my_mask;
    base[t] = before + prefix + my_count;
    lbase[t] = lprefix;
    if (chunk == 0) out_row_offsets[c0 + t] = before + prefix;
  }
  if (chunk == 0 && c1 == n && t == 0) out_row_offsets[n] = before + sum_totals;
  __syncthreads();

  auto place = [&](int row_in_chunk, int p, int col, float val) {
    const mask_t below = (mask_t{1} << row_in_chunk) - 1;
    const int rank = __popc(masks[col - c0] & below);
    const int pos = base[col - c0] + rank;
    const int slot = lbase[col - c0] + rank;
    if (slot < kStageCap) {
      st_pos[slot] = pos;
      st_src[slot] = p;
      st_row[slot] = static_cast<unsigned char>(row_in_chunk);
      if (stage_values) st_val[slot] = val;
    } else {
      out_column_indices[pos] = row0 + row_in_chunk;
      if (out_permutation != nullptr) out_permutation[pos] = p;
      for (int r = 0; r < replicas; ++r)
        out_values[r * out_values_stride + pos] = values[r * values_stride + p];
    }
  };
#pragma unroll
  for (int j = 0; j < kRows; ++j) {
    const int row_in_chunk = wave + j * (kGroupBlock / kWave);
    for (int first = p0[j] + lane; first < p1[j]; first += kUnroll * kWave) {
      int c[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int p = first + u * kWave;
        c[u] = p < p1[j] ? column_indices[p] : -1;
      }
      // (the value is fetched for the entries that pass only: requesting all of
      // them with the columns saves a round trip but reads 8x the bytes -- 15.0
      // against 12.7 us)
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
        if (c[u] >= c0 && c[u] < c1)
          place(row_in_chunk, first + u * kWave, c[u],
                stage_values ? values[first + u * kWave] : 0.f);
    }
  }
  __syncthreads();
  const int staged = min(sum_local, kStageCap);
  for (int i = t; i < staged; i += kGroupBlock) {
    const int pos = st_pos[i];
    out_column_indices[pos] = row0 + st_row[i];
    if (out_permutation != nullptr) out_permutation[pos] = st_src[i];
    if (stage_values) out_values[pos] = st_val[i];
  }
  if (stage_values || replicas == 0) return;
  // the values of several replicas: a thread's (up to) 8 entries per replica in flight
  // together -- the loads of a replica do not wait for the stores of the one before
  constexpr int kPer = kStageCap / kGroupBlock;
  int pos[kPer], src[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int i = t + u * kGroupBlock;
    pos[u] = i < staged ? st_pos[i] : -1;
    src[u] = i < staged ? st_src[i] : 0;
  }
  for (int r = 0; r < replicas; ++r) {
    const float* __restrict__ in = values + r * values_stride;
    float* __restrict__ out = out_values + r * out_values_stride;
    float v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) v[u] = pos[u] >= 0 ? in[src[u]] : 0.f;
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (pos[u] >= 0) out[pos[u]] = v[u];
  }
}

inline int chunks_of(int m) { return ceil_div(m, kRowsPerChunk); }

// The table path costs 2 * chunks * n table entries of work and workspace whatever
// nnz is; beyond eight per nonzero the O(n + nnz) path takes over (2048^2 at density
// 0.2 and 4096^2 at 0.1 stay on the tables; 65536^2 at 1e-4 does not).
inline bool sparse_path(int m, int n, int nonzeros) {
  return static_cast<int64_t>(chunks_of(m)) * n > 8 * static_cast<int64_t>(nonzeros);
}
inline size_t table_bytes(int m, int n) {   // masks, counts, totals, one balance per mask workgroup
  return sizeof(int) * (2 * static_cast<size_t>(chunks_of(m)) * n + n +
                        static_cast<size_t>(chunks_of(m)) * ceil_div(n, kColsPerRange));
}
inline size_t sparse_bytes(int n, int nonzeros) {   // counts + cursor + (row, source, permutation) per nonzero
  return sizeof(int) * (2 * static_cast<size_t>(n) + 3 * static_cast<size_t>(nonzeros));
}
inline size_t status_offset(int m, int n, int nonzeros) {
  const size_t body = sparse_path(m, n, nonzeros) ? sparse_bytes(n, nonzeros) : table_bytes(m, n);
  return (body + 15) / 16 * 16;
}

}  // namespace

// All masks of a "many mask" batch in three launches (many_mask.hip).  Taken when the
// LARGEST mask is a table-path matrix of one column range (every attention mask is) and
// the workspace holds one region of tables per mask; returns -1 otherwise (the caller
// then transposes mask after mask).  With several heads per mask the scatter moves the
// topology and the permutation only and ONE gather moves the values of all replicas: the
// permutation then needs room behind the regions if the caller takes none.  Measured at 8
// masks x 8 heads of 1024^2 (densities 0.05 .. 0.5, 14.2 M values): 238 us mask after mask,
// 127 us here = masks 7 + scan 5 + scatter 38 + gather 77; with the values in the scatter
// (eight replicas per staged entry, loads of a replica batched) 133.  Either way a value
// costs one 4-byte request to L2 on one side -- the gather's reads, the scatter's runs of
// 32 x density entries per column and chunk -- and 14.2 M of them take 77 us; full lines on
// both sides need chunks of 32 / density rows (not built).
size_t csr_transpose_many_region_bytes(int m, int n, int largest_nonzeros) {
  return (status_offset(m, n, largest_nonzeros) + 16 + 255) / 256 * 256;
}
size_t csr_transpose_many_bytes(int masks, int m, int n, int largest_nonzeros) {
  return static_cast<size_t>(masks) * (csr_transpose_many_region_bytes(m, n, largest_nonzeros) +
                                       sizeof(int) * static_cast<size_t>(largest_nonzeros));
}
int csr_transpose_many_launch(int masks, int m, int n, int largest_nonzeros, int64_t total_nonzeros,
                              int heads, const float* values, int64_t values_stride,
                              const int* row_offsets, const int* column_indices, float* out_values,
                              int64_t out_values_stride, int* out_row_offsets,
                              int* out_column_indices, int* out_permutation, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
  const int chunks = chunks_of(m);
  const int groups = column_groups(n);
  const size_t region_bytes = csr_transpose_many_region_bytes(m, n, largest_nonzeros);
  const bool gather = heads > 1;
  const size_t tables_bytes = region_bytes * static_cast<size_t>(masks);
  const size_t need = tables_bytes + (gather && out_permutation == nullptr
                                          ? sizeof(int) * static_cast<size_t>(total_nonzeros) : 0);
  if (masks < 2 || masks > kMaxGridYZ || heads > kMaxGridYZ || largest_nonzeros <= 0 ||
      static_cast<int64_t>(ceil_div(largest_nonzeros, kBlock * kManyValuesUnroll)) * max(heads, 1) * masks >=
          (int64_t{1} << 32) ||
      n > kColsPerRange || sparse_path(m, n, largest_nonzeros) || chunks > 0x7fffffff / groups ||
      workspace == nullptr || workspace_bytes < need)
    return -1;
  const int64_t region = static_cast<int64_t>(region_bytes / sizeof(int));
  mask_t* gmask = static_cast<mask_t*>(workspace);
  int* table = reinterpret_cast<int*>(gmask + static_cast<size_t>(chunks) * n);
  int* totals = table + static_cast<size_t>(chunks) * n;
  int* balances = totals + n;
  int* status = reinterpret_cast<int*>(static_cast<char*>(workspace) + status_offset(m, n, largest_nonzeros));
  int* permutation = out_permutation;
  if (gather && permutation == nullptr)
    permutation = reinterpret_cast<int*>(static_cast<char*>(workspace) + tables_bytes);
  hipLaunchKernelGGL(transpose_mask_kernel, dim3(chunks, 1, masks), dim3(kMaskBlock),
                     sizeof(mask_t) * static_cast<size_t>(n), stream, m, n, row_offsets,
                     column_indices, gmask, balances, region);
  hipLaunchKernelGGL(transpose_scan_table_kernel, dim3(ceil_div(n, kWave), masks),
                     dim3(kWave * kScanGroups), 0, stream, n, chunks, gmask, table, totals, balances,
                     chunks, status, region);
  hipLaunchKernelGGL(transpose_scatter_grouped_kernel, dim3(chunks * groups, masks),
                     dim3(kGroupBlock), 0, stream, m, n, groups, gather ? 0 : heads,
                     gather ? nullptr : values, values_stride, row_offsets, column_indices, gmask,
                     table, totals, out_row_offsets, gather ? nullptr : out_values,
                     out_values_stride, out_column_indices, permutation, region);
  if (gather)
    hipLaunchKernelGGL(transpose_many_values_kernel,
                       dim3(ceil_div(largest_nonzeros, kBlock * kManyValuesUnroll), heads, masks),
                       dim3(kBlock), 0, stream, m, heads, values, values_stride, row_offsets,
                       permutation, out_values, out_values_stride);
  return launch_status();
}

}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

size_t sputnik_hip_csr_transpose_workspace_bytes(int m, int n, int nonzeros) {
  if (m <= 0 || n <= 0) return 0;
  // tables: masks [chunks][n] + counts [chunks][n] + totals [n]; or the O(n + nnz)
  // path's arrays; + the status word (16 bytes)
  return status_offset(m, n, max(nonzeros, 0)) + 16;
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

  int* status = reinterpret_cast<int*>(static_cast<char*>(workspace) + status_offset(m, n, nonzeros));
  if (sparse_path(m, n, nonzeros)) {
    int* counts = static_cast<int*>(workspace);
    int* cursor = counts + n;
    int* tmp_row = cursor + n;
    int* tmp_src = tmp_row + nonzeros;
    int* perm = out_permutation != nullptr ? out_permutation : tmp_src + nonzeros;
    // counts and the status word in one memset (the status word sits behind the arrays)
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * static_cast<size_t>(n), stream);
    if (e == hipSuccess) e = hipMemsetAsync(status, 0, sizeof(int), stream);
    if (e != hipSuccess) return static_cast<int>(e);
    const int row_blocks = ceil_div(m, kWaves);
    hipLaunchKernelGGL(transpose_sparse_rows_kernel<false>, dim3(row_blocks), dim3(kBlock), 0,
                       stream, m, n, row_offsets, column_indices, counts, nullptr, nullptr, status);
    hipLaunchKernelGGL(transpose_sparse_scan_kernel, dim3(1), dim3(1024), 0, stream, n, counts,
                       out_row_offsets, cursor);
    hipLaunchKernelGGL(transpose_sparse_rows_kernel<true>, dim3(row_blocks), dim3(kBlock), 0,
                       stream, m, n, row_offsets, column_indices, cursor, tmp_row, tmp_src, status);
    hipLaunchKernelGGL(transpose_sparse_rank_kernel, dim3(ceil_div(n, kWaves)), dim3(kBlock), 0,
                       stream, n, out_row_offsets, tmp_row, tmp_src, out_column_indices, perm,
                       status);
    if (nonzeros > kLongColumn)   // (a workgroup per output row; all but the long rows leave at once)
      hipLaunchKernelGGL(transpose_sparse_rank_long_kernel, dim3(n), dim3(kBlock), 0, stream, m,
                         out_row_offsets, tmp_row, tmp_src, out_column_indices, perm, status);
    hipLaunchKernelGGL(transpose_sparse_values_kernel<float>, dim3(ceil_div(nonzeros, kBlock)),
                       dim3(kBlock), 0, stream, nonzeros, replicas, values, values_stride, perm,
                       out_values, out_values_stride);
    return launch_status();
  }
  const int chunks = chunks_of(m);
  const int ranges = ceil_div(n, kColsPerRange);
  if (ranges > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  mask_t* gmask = static_cast<mask_t*>(workspace);
  int* table = reinterpret_cast<int*>(gmask + static_cast<size_t>(chunks) * n);
  int* totals = table + static_cast<size_t>(chunks) * n;
  int* balances = totals + n;
  const size_t range_cols = static_cast<size_t>(min(n, kColsPerRange));
  const size_t lds_bytes = sizeof(mask_t) * range_cols;
  const int groups = column_groups(n);
  const bool grouped = ranges == 1 && chunks <= 0x7fffffff / groups;

  hipLaunchKernelGGL(transpose_mask_kernel, dim3(chunks, ranges), dim3(kMaskBlock), lds_bytes,
                     stream, m, n, row_offsets, column_indices, gmask, balances, int64_t{0});
  int st = launch_status();
  if (st != 0) return st;
  hipLaunchKernelGGL(transpose_scan_table_kernel, dim3(ceil_div(n, kWave)),
                     dim3(kWave * kScanGroups), 0, stream, n, chunks, gmask, table, totals, balances,
                     chunks * ranges, status, int64_t{0});
  st = launch_status();
  if (st != 0) return st;
  if (grouped) {
    hipLaunchKernelGGL(transpose_scatter_grouped_kernel, dim3(chunks * groups), dim3(kGroupBlock),
                       0, stream, m, n, groups, replicas, values, values_stride, row_offsets,
                       column_indices, gmask, table, totals, out_row_offsets,
                       out_values, out_values_stride, out_column_indices, out_permutation,
                       int64_t{0});
    return launch_status();
  }
  hipLaunchKernelGGL(transpose_scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, stream, n,
                     totals, out_row_offsets);
  st = launch_status();
  if (st != 0) return st;
  // Workgroups per chunk in the plain scatter: enough for about two per CU (the
  // chunk's 32 rows split into parts of 16 / 8 / 4).
  int parts = 1;
  while (parts < 8 && static_cast<int64_t>(chunks) * ranges * parts < 512) parts *= 2;
  if (chunks > 0x7fffffff / parts) return SPUTNIK_HIP_INVALID_ARGUMENT;
  hipLaunchKernelGGL(transpose_scatter_kernel, dim3(chunks * parts, ranges), dim3(kBlock),
                     lds_bytes + sizeof(int) * range_cols, stream, m, n, parts,
                     kRowsPerChunk / parts, replicas, values, values_stride, row_offsets,
                     column_indices, gmask, table, out_row_offsets, out_values, out_values_stride,
                     out_column_indices, out_permutation);
  return launch_status();
}

int sputnik_hip_csr_transpose_checked(int m, int n, int nonzeros, int replicas,
                                      const float* values, int64_t values_stride,
                                      const int* row_offsets, const int* column_indices,
                                      float* out_values, int64_t out_values_stride,
                                      int* out_row_offsets, int* out_column_indices,
                                      int* out_permutation, void* workspace,
                                      size_t workspace_bytes, sputnik_hip_stream_t stream) {
  const int st = sputnik_hip_csr_transpose(m, n, nonzeros, replicas, values, values_stride,
                                           row_offsets, column_indices, out_values,
                                           out_values_stride, out_row_offsets, out_column_indices,
                                           out_permutation, workspace, workspace_bytes, stream);
  if (st != 0 || m <= 0 || n <= 0 || nonzeros <= 0) return st;
  int status = 0;
  const int* device_status =
      reinterpret_cast<const int*>(static_cast<const char*>(workspace) + status_offset(m, n, nonzeros));
  hipError_t e = hipMemcpyAsync(&status, device_status, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return static_cast<int>(e);
  return status != 0 ? SPUTNIK_HIP_INVALID_ARGUMENT : 0;
}

// values stored as float16 / bfloat16: the transposition (topology and permutation) runs
// without touching them, then ONE gather reads the half values through the permutation
// and writes float32 -- the widening rides on the move every value makes anyway.
int sputnik_hip_csr_transpose_typed(int m, int n, int nonzeros, int replicas, const void* values,
                                    int values_type, int64_t values_stride,
                                    const int* row_offsets, const int* column_indices,
                                    float* out_values, int64_t out_values_stride,
                                    int* out_row_offsets, int* out_column_indices,
                                    int* out_permutation, void* workspace, size_t workspace_bytes,
                                    int checked, sputnik_hip_stream_t stream) {
  const auto base = checked ? sputnik_hip_csr_transpose_checked : sputnik_hip_csr_transpose;
  if (values_type == SPUTNIK_HIP_F32)
    return base(m, n, nonzeros, replicas, static_cast<const float*>(values), values_stride,
                row_offsets, column_indices, out_values, out_values_stride, out_row_offsets,
                out_column_indices, out_permutation, workspace, workspace_bytes, stream);
  if ((values_type != SPUTNIK_HIP_F16 && values_type != SPUTNIK_HIP_BF16) || replicas < 0 ||
      (out_permutation == nullptr && m > 0 && n > 0 && nonzeros > 0) || !aligned_to(values, 2))
    return SPUTNIK_HIP_INVALID_ARGUMENT;   // (the half forms need the permutation array)
  const int st = base(m, n, nonzeros, /*replicas=*/0, nullptr, 0, row_offsets, column_indices,
                      nullptr, 0, out_row_offsets, out_column_indices, out_permutation, workspace,
                      workspace_bytes, stream);
  if (st != 0 || m <= 0 || n <= 0 || nonzeros <= 0 || replicas == 0) return st;
  if (values_type == SPUTNIK_HIP_F16)
    hipLaunchKernelGGL(transpose_sparse_values_kernel<_Float16>, dim3(ceil_div(nonzeros, kBlock)),
                       dim3(kBlock), 0, stream, nonzeros, replicas,
                       static_cast<const _Float16*>(values), values_stride, out_permutation,
                       out_values, out_values_stride);
  else
    hipLaunchKernelGGL(transpose_sparse_values_kernel<__bf16>, dim3(ceil_div(nonzeros, kBlock)),
                       dim3(kBlock), 0, stream, nonzeros, replicas,
                       static_cast<const __bf16*>(values), values_stride, out_permutation,
                       out_values, out_values_stride);
  return launch_status();
}

}  // extern "C"
