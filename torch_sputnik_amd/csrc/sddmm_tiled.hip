// LDS-tiled SDDMM for short inner dimensions (k = 64 or 128: attention heads).
//
//   out[p] = < lhs[i_p, 0:k], rhs[j_p, 0:k] >   for every stored (i_p, j_p)
//
// The row-wave kernel of sddmm.hip gathers one rhs row per nonzero from L2
// (nnz*k*4 bytes of cache traffic: 1.7 GB for config 3, which is what its
// 119 us were).  Here a workgroup owns 128 rows of the mask and walks the
// COLUMNS in chunks of 128: the matching 128 rows of rhs are staged once per
// workgroup into LDS (direct global->LDS copies, double buffered) and gathered
// from there.  The structure is the 64-column SpMM kernel's (spmm_tiled64.hip)
// with the multiply-add turned around:
//   * each 16-lane row group of a wave owns one mask row; its lhs row lives in
//     registers (k/16 floats per lane);
//   * the group's next 32 column indices per chunk are prefetched one chunk
//     ahead (lane = entry) and handed out with DPP row_newbcast;
//   * per nonzero: one ds_read_b128 per 64 inner elements, 4 FMAs, a 4-step DPP
//     sum over the group; lane u keeps result u, so 16 results leave as one
//     64-byte store.
// Needs ascending columns inside rows (checked per row by the shared pre-pass;
// other row blocks take an order-independent path in the same launch).
#include <stdlib.h>

#include <type_traits>

#include "spmm_tiled_common.h"

namespace sputnik_hip {
namespace {

using namespace tiled;

constexpr int kWaves = 8;  // waves per workgroup
constexpr int kRQ = 4;     // row quads per wave (4 rows each)
constexpr int kBK = 128;   // rhs rows (mask columns) per LDS stage
constexpr int kBM = kWaves * kRQ * 4;
constexpr int kThreads = kWaves * kWave;
constexpr int kWin = 2;    // 16-entry windows prefetched per row and chunk

// One wave instruction copies 1 KiB = 4/KV tile rows of 64*KV floats.
template <int KV>
__device__ __forceinline__ void stage_rhs(float* __restrict__ tile, const float* __restrict__ rhs,
                                          int kdim, int n, int jc, int wave, int lane) {
  constexpr int kLanesPerRow = 16 * KV;
  constexpr int kRowsPerCopy = kWave / kLanesPerRow;
  constexpr int kCopies = kBK / kRowsPerCopy / kWaves;
  const int lr = lane / kLanesPerRow, lc = lane % kLanesPerRow;
#pragma unroll
  for (int j = 0; j < kCopies; ++j) {
    const int r0 = (wave + j * kWaves) * kRowsPerCopy;
    const int src_row = min(jc + r0 + lr, n - 1);  // past the last column: re-read the last row
    const unsigned off =
        (static_cast<unsigned>(src_row) * static_cast<unsigned>(kdim) + lc * 4u) * 4u;
    lds_dma_row(rhs, off, tile + r0 * (64 * KV));
  }
}

template <int KV>
__global__ __launch_bounds__(kThreads) void sddmm_tiled_kernel(
    int m, int n, int nonzeros, int slots, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const int* __restrict__ table, const int* __restrict__ row_ok,
    const float* __restrict__ lhs, int64_t lhs_stride, const float* __restrict__ rhs,
    int64_t rhs_stride, float* __restrict__ out, int64_t out_stride) {
  constexpr int kdim = 64 * KV;
  constexpr int kTileFloats = kBK * kdim;
  __shared__ float tile[2][kTileFloats];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const int mblock = blockIdx.x;
  const int replica = blockIdx.y;
  lhs += replica * lhs_stride;
  rhs += replica * rhs_stride;
  out += replica * out_stride;
  const int slot0 = mblock * kBM + wave * (kRQ * 4);
  const int last = nonzeros - 1;

  // This group's lhs rows (lane i holds elements 64v + 4i .. +3 of each).
  float4 lf[kRQ][KV];
  int my_row[kRQ];
#pragma unroll
  for (int t = 0; t < kRQ; ++t) {
    const int slot = slot0 + 4 * t + g;
    my_row[t] = slot < m ? row_indices[slot] : -1;
#pragma unroll
    for (int v = 0; v < KV; ++v)
      lf[t][v] = my_row[t] >= 0
                     ? *reinterpret_cast<const float4*>(lhs + static_cast<int64_t>(my_row[t]) * kdim +
                                                        64 * v + 4 * i)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  auto dot = [&](int t, const float4 (&b)[KV]) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      s = fmaf(lf[t][v].x, b[v].x, s);
      s = fmaf(lf[t][v].y, b[v].y, s);
      s = fmaf(lf[t][v].z, b[v].z, s);
      s = fmaf(lf[t][v].w, b[v].w, s);
    }
    return s;
  };

  // Row blocks whose columns do not ascend inside rows: order-independent path,
  // rhs rows gathered from global memory, one row per 16-lane group.
  if (!block_rows_ok(row_ok, mblock * kBM, kBM)) {
    for (int t = 0; t < kRQ; ++t) {
      const int p0 = my_row[t] >= 0 ? row_offsets[my_row[t]] : 0;
      const int p1 = my_row[t] >= 0 ? row_offsets[my_row[t] + 1] : 0;
      for (int p = p0; p < p1; ++p) {
        float4 b[KV];
#pragma unroll
        for (int v = 0; v < KV; ++v)
          b[v] = *reinterpret_cast<const float4*>(
              rhs + static_cast<int64_t>(column_indices[p]) * kdim + 64 * v + 4 * i);
        const float total = group_sum<16>(dot(t, b));
        if (i == 0) out[p] = total;
      }
    }
    return;
  }

  const int* __restrict__ my_table = table + slot0 + g;
  int ps[kRQ], pe[kRQ], wcol[kRQ][kWin];
#pragma unroll
  for (int t = 0; t < kRQ; ++t) {
    ps[t] = my_table[4 * t];
    pe[t] = my_table[slots + 4 * t];
#pragma unroll
    for (int w = 0; w < kWin; ++w) wcol[t][w] = column_indices[min(ps[t] + 16 * w + i, last)];
  }

  stage_rhs<KV>(tile[0], rhs, kdim, n, 0, wave, lane);
  wait_vm<0>();
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunks;
    if (more) stage_rhs<KV>(tile[buf ^ 1], rhs, kdim, n, (c + 1) * kBK, wave, lane);

    int pe_next[kRQ], ncol[kRQ][kWin];
#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      pe_next[t] = more ? my_table[static_cast<int64_t>(c + 2) * slots + 4 * t] : pe[t];
#pragma unroll
      for (int w = 0; w < kWin; ++w)
        ncol[t][w] = more ? column_indices[min(pe[t] + 16 * w + i, last)] : 0;
    }

    const char* __restrict__ lane_base = reinterpret_cast<const char*>(&tile[buf][0] + i * 4);
    const int jc = c * kBK;

#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      const int cnt = pe[t] - ps[t];  // this group's row; the same in its 16 lanes

      // One 16-entry window: the group's lanes stop at their own row's count.
      auto window = [&](int ecol, int w0) {
        const int left = cnt - w0;
        const bool valid = i < left;
        const int roff = valid ? ((ecol - jc) * (kdim * 4)) : 0;
        float result = 0.f;
        auto four = [&](auto G) {
          constexpr int kG = decltype(G)::value;
          const int o0 = row_bcast_i<kG + 0>(roff), o1 = row_bcast_i<kG + 1>(roff);
          const int o2 = row_bcast_i<kG + 2>(roff), o3 = row_bcast_i<kG + 3>(roff);
          float4 b0[KV], b1[KV], b2[KV], b3[KV];
#pragma unroll
          for (int v = 0; v < KV; ++v) {
            b0[v] = *reinterpret_cast<const float4*>(lane_base + o0 + 256 * v);
            b1[v] = *reinterpret_cast<const float4*>(lane_base + o1 + 256 * v);
            b2[v] = *reinterpret_cast<const float4*>(lane_base + o2 + 256 * v);
            b3[v] = *reinterpret_cast<const float4*>(lane_base + o3 + 256 * v);
          }
          const float t0 = group_sum<16>(dot(t, b0)), t1 = group_sum<16>(dot(t, b1));
          const float t2 = group_sum<16>(dot(t, b2)), t3 = group_sum<16>(dot(t, b3));
          result = (i == kG + 0) ? t0 : result;
          result = (i == kG + 1) ? t1 : result;
          result = (i == kG + 2) ? t2 : result;
          result = (i == kG + 3) ? t3 : result;
        };
        if (left > 0) four(std::integral_constant<int, 0>{});
        if (left > 4) four(std::integral_constant<int, 4>{});
        if (left > 8) four(std::integral_constant<int, 8>{});
        if (left > 12) four(std::integral_constant<int, 12>{});
        if (valid) out[ps[t] + w0 + i] = result;
      };
#pragma unroll
      for (int w = 0; w < kWin; ++w) window(wcol[t][w], 16 * w);
      const int longest = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                              max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
      for (int w0 = 16 * kWin; w0 < longest; w0 += 16)
        window(column_indices[min(ps[t] + w0 + i, last)], w0);
    }

#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      ps[t] = pe[t];
      pe[t] = pe_next[t];
#pragma unroll
      for (int w = 0; w < kWin; ++w) wcol[t][w] = ncol[t][w];
    }
    wait_vm<0>();     // the next rhs tile has landed
    __syncthreads();  // ... for every wave, and the current buffer is free
  }
}

// ---------------------------------------------------------------------------
// rhs-stationary variant.  The kernel above re-stages rhs chunk after chunk
// and pays a barrier plus an exposed global->LDS latency per chunk; with a few
// thousand nonzeros per workgroup and chunk that skeleton was a third of its
// time.  Here a workgroup keeps ONE slab of rhs rows (64 KiB: 256 rows at
// k = 64, 128 at k = 128) in LDS for its whole life and walks mask rows
// instead: for each row, the entries whose column falls in the slab (found
// with the chunk table) are computed exactly as above.  One barrier per
// workgroup; everything after it is ordinary loads the compiler schedules.
// Outputs of different slabs are disjoint, so no reduction is needed.
// grid = (slabs, row blocks, replicas).
// ---------------------------------------------------------------------------
constexpr int kSlabBytes = 64 * 1024;
constexpr int kSWaves = 8;
constexpr int kSThreads = kSWaves * kWave;
constexpr int kSGroups = kSWaves * 4;  // 16-lane row groups per workgroup
constexpr int kSRows = 8;              // mask rows per group: a workgroup owns 256 rows

template <int KV>
__global__ __launch_bounds__(kSThreads) void sddmm_stationary_kernel(
    int m, int n, int nonzeros, int slots,
    const int* __restrict__ row_indices, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, const int* __restrict__ table,
    const int* __restrict__ row_ok, const float* __restrict__ lhs, int64_t lhs_stride,
    const float* __restrict__ rhs, int64_t rhs_stride, float* __restrict__ out,
    int64_t out_stride, int debug) {
  constexpr int kdim = 64 * KV;
  constexpr int kRows = kSlabBytes / (kdim * 4);
  __shared__ float tile[kRows * kdim];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const int slab = blockIdx.x;
  const int replica = blockIdx.z;
  lhs += replica * lhs_stride;
  rhs += replica * rhs_stride;
  out += replica * out_stride;
  const int jc = slab * kRows;
  const int last = nonzeros - 1;

  if (!(debug & 2)) {  // stage the slab: one wave instruction copies 1 KiB = 4/KV rows
    constexpr int kLanesPerRow = 16 * KV;
    constexpr int kRowsPerCopy = kWave / kLanesPerRow;
    constexpr int kCopies = kRows / kRowsPerCopy / kSWaves;
    const int lr = lane / kLanesPerRow, lc = lane % kLanesPerRow;
#pragma unroll
    for (int j = 0; j < kCopies; ++j) {
      const int r0 = (wave + j * kSWaves) * kRowsPerCopy;
      const int src_row = min(jc + r0 + lr, n - 1);
      const unsigned off =
          (static_cast<unsigned>(src_row) * static_cast<unsigned>(kdim) + lc * 4u) * 4u;
      lds_dma_row(rhs, off, tile + r0 * kdim);
    }
  }

  // This group's kSRows mask rows.  All their bookkeeping (row id, first entry
  // inside the slab, count) is fetched in one go while the slab is still in
  // flight; a row's column windows and lhs fragment are fetched two rows ahead.
  // Everything is statically indexed (the row loop is fully unrolled), so no
  // register is ever copied while its load is outstanding.
  const int slot_begin = blockIdx.y * (kSGroups * kSRows);
  const int gid = wave * 4 + g;
  const int* __restrict__ tab0 = table + static_cast<int64_t>(slab) * slots;
  const int* __restrict__ tab1 = tab0 + slots;

  int row[kSRows], ps[kSRows], cnt[kSRows];
#pragma unroll
  for (int r = 0; r < kSRows; ++r) {
    const int slot = slot_begin + r * kSGroups + gid;
    const bool live = slot < m;
    const int sl = live ? slot : 0;
    row[r] = row_indices[sl];
    const int a0 = tab0[sl], a1 = tab1[sl];
    // rows whose columns do not ascend have no valid table entries: slab 0 does
    // the whole row in storage order (negative count), the other slabs skip it
    const bool ok = row_ok[sl] != 0;
    ps[r] = ok ? a0 : row_offsets[row[r]];
    const int len = ok ? a1 - a0 : row_offsets[row[r] + 1] - ps[r];
    cnt[r] = !live ? 0 : ok ? len : (slab == 0 ? -len : 0);
  }

  constexpr int kRing = 3;
  int wcol[kRing][kWin];
  float4 lf[kRing][KV];
  auto fetch = [&](int r, int slot_in_ring) {
#pragma unroll
    for (int w = 0; w < kWin; ++w)
      wcol[slot_in_ring][w] = column_indices[min(ps[r] + 16 * w + i, last)];
#pragma unroll
    for (int v = 0; v < KV; ++v)
      lf[slot_in_ring][v] = *reinterpret_cast<const float4*>(
          lhs + static_cast<int64_t>(row[r]) * kdim + 64 * v + 4 * i);
  };
  fetch(0, 0);
  fetch(1, 1);
  wait_vm<0>();
  __syncthreads();

  const char* __restrict__ lane_base = reinterpret_cast<const char*>(&tile[0] + i * 4);
#pragma unroll
  for (int r = 0; r < kSRows; ++r) {
    if (r + 2 < kSRows) fetch(r + 2, (r + 2) % kRing);
    const float4 (&cur_lf)[KV] = lf[r % kRing];
    const int cur_ps = ps[r];

    auto dot = [&](const float4 (&b)[KV]) {
      float acc = 0.f;
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        acc = fmaf(cur_lf[v].x, b[v].x, acc);
        acc = fmaf(cur_lf[v].y, b[v].y, acc);
        acc = fmaf(cur_lf[v].z, b[v].z, acc);
        acc = fmaf(cur_lf[v].w, b[v].w, acc);
      }
      return acc;
    };

    if (cnt[r] < 0) {
      // unsorted row (rare): rhs rows gathered from global memory, any column
      const int p1 = cur_ps - cnt[r];
      for (int p = cur_ps; p < p1; ++p) {
        float4 b[KV];
#pragma unroll
        for (int v = 0; v < KV; ++v)
          b[v] = *reinterpret_cast<const float4*>(
              rhs + static_cast<int64_t>(column_indices[p]) * kdim + 64 * v + 4 * i);
        const float total = group_sum<16>(dot(b));
        if (i == 0) out[p] = total;
      }
    }
    const int n_here = (debug & 1) ? 0 : max(cnt[r], 0);
    auto window = [&](int ecol, int w0) {
      const int left = n_here - w0;
      const bool valid = i < left;
      const int roff = valid ? ((ecol - jc) * (kdim * 4)) : 0;
      float result = 0.f;
      // partial dot products of entries G..G+3 (this lane's 4*KV inner elements)
      auto four = [&](auto G, float& d0, float& d1, float& d2, float& d3) {
        constexpr int kG = decltype(G)::value;
        const int o0 = row_bcast_i<kG + 0>(roff), o1 = row_bcast_i<kG + 1>(roff);
        const int o2 = row_bcast_i<kG + 2>(roff), o3 = row_bcast_i<kG + 3>(roff);
        float4 b0[KV], b1[KV], b2[KV], b3[KV];
#pragma unroll
        for (int v = 0; v < KV; ++v) {
          b0[v] = *reinterpret_cast<const float4*>(lane_base + o0 + 256 * v);
          b1[v] = *reinterpret_cast<const float4*>(lane_base + o1 + 256 * v);
          b2[v] = *reinterpret_cast<const float4*>(lane_base + o2 + 256 * v);
          b3[v] = *reinterpret_cast<const float4*>(lane_base + o3 + 256 * v);
        }
        d0 = dot(b0);
        d1 = dot(b1);
        d2 = dot(b2);
        d3 = dot(b3);
      };
      if (left > 4) {
        // 5..16 entries: all partials first, then ONE transposing reduction
        // that leaves entry u's sum in lane u
        float p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) p[u] = 0.f;
        four(std::integral_constant<int, 0>{}, p[0], p[1], p[2], p[3]);
        four(std::integral_constant<int, 4>{}, p[4], p[5], p[6], p[7]);
        if (left > 8) four(std::integral_constant<int, 8>{}, p[8], p[9], p[10], p[11]);
        if (left > 12) four(std::integral_constant<int, 12>{}, p[12], p[13], p[14], p[15]);
        result = row_transpose_sum16(p, i);
      } else if (left > 0) {
        float d0, d1, d2, d3;
        four(std::integral_constant<int, 0>{}, d0, d1, d2, d3);
        const float t0 = group_sum<16>(d0), t1 = group_sum<16>(d1);
        const float t2 = group_sum<16>(d2), t3 = group_sum<16>(d3);
        result = (i == 0) ? t0 : (i == 1) ? t1 : (i == 2) ? t2 : t3;
      }
      if (valid) out[cur_ps + w0 + i] = result;
    };
#pragma unroll
    for (int w = 0; w < kWin; ++w) window(wcol[r % kRing][w], 16 * w);
    const int longest =
        max(max(__builtin_amdgcn_readlane(n_here, 0), __builtin_amdgcn_readlane(n_here, 16)),
            max(__builtin_amdgcn_readlane(n_here, 32), __builtin_amdgcn_readlane(n_here, 48)));
    for (int w0 = 16 * kWin; w0 < longest; w0 += 16)
      window(column_indices[min(cur_ps + w0 + i, last)], w0);
  }
}

inline int slots_of(int m) { return ceil_div(m, kBM) * kBM; }
inline int chunks_of(int n) { return ceil_div(n, kBK); }

}  // namespace

bool sddmm_tiled_applicable(int m, int k, int n, int nonzeros, const float* lhs,
                            int64_t lhs_stride, const float* rhs, int64_t rhs_stride) {
  return (k == 64 || k == 128) && n >= 64 && m >= 16 && nonzeros >= 4 * static_cast<int64_t>(m) &&
         static_cast<int64_t>(n) * k * 4 < (int64_t{1} << 32) && aligned_to(lhs, 16) &&
         aligned_to(rhs, 16) && lhs_stride % 4 == 0 && rhs_stride % 4 == 0;
}

size_t sddmm_tiled_workspace_bytes(int m, int k, int n, int nonzeros) {
  if (!(k == 64 || k == 128) || n < 64 || m < 16 || nonzeros < 4 * static_cast<int64_t>(m)) return 0;
  return row_ok_bytes(slots_of(m)) +
         sizeof(int) * static_cast<size_t>(chunks_of(n) + 1) * slots_of(m);
}

int sddmm_tiled_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                       const int* row_offsets, const int* column_indices, const float* lhs,
                       int64_t lhs_stride, const float* rhs, int64_t rhs_stride, float* out,
                       int64_t out_stride, void* workspace, hipStream_t stream) {
  static const int streamed = [] {
    const char* e = getenv("SPUTNIK_HIP_SDDMM_STREAMED");  // developer knob: 1 = chunk-streaming kernel
    return e ? atoi(e) : 0;
  }();
  static const int debug = [] {
    const char* e = getenv("SPUTNIK_HIP_SDDMM_DEBUG");  // timing experiments only
    return e ? atoi(e) : 0;
  }();
  const int slots = slots_of(m);
  int* row_ok = static_cast<int*>(workspace);
  int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + row_ok_bytes(slots));
  int st;
  if (!streamed) {
    // rhs-stationary: the chunk table is cut at slab boundaries
    const int slab_rows = kSlabBytes / (k * 4);
    const int slabs = ceil_div(n, slab_rows);
    if (k == 64) {
      hipLaunchKernelGGL((spmm_chunk_table_kernel<ilog2(kSlabBytes / 256)>),
                         dim3(ceil_div(slots, 4)), dim3(256), 0, stream, m, n, slots, slabs,
                         row_indices, row_offsets, column_indices, table, row_ok);
    } else {
      hipLaunchKernelGGL((spmm_chunk_table_kernel<ilog2(kSlabBytes / 512)>),
                         dim3(ceil_div(slots, 4)), dim3(256), 0, stream, m, n, slots, slabs,
                         row_indices, row_offsets, column_indices, table, row_ok);
    }
    st = launch_status();
    if (st != 0) return st;
    const int row_blocks = ceil_div(m, kSGroups * kSRows);
    if (row_blocks > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
    for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
      const int rz = min(replicas - r0, kMaxGridYZ);
      const dim3 grid(slabs, row_blocks, rz);
      if (k == 64) {
        hipLaunchKernelGGL(sddmm_stationary_kernel<1>, grid, dim3(kSThreads), 0, stream, m, n,
                           nonzeros, slots, row_indices, row_offsets,
                           column_indices, table, row_ok, lhs + r0 * lhs_stride, lhs_stride,
                           rhs + r0 * rhs_stride, rhs_stride, out + r0 * out_stride, out_stride, debug);
      } else {
        hipLaunchKernelGGL(sddmm_stationary_kernel<2>, grid, dim3(kSThreads), 0, stream, m, n,
                           nonzeros, slots, row_indices, row_offsets,
                           column_indices, table, row_ok, lhs + r0 * lhs_stride, lhs_stride,
                           rhs + r0 * rhs_stride, rhs_stride, out + r0 * out_stride, out_stride,
                           debug);
      }
      st = launch_status();
      if (st != 0) return st;
    }
    return 0;
  }
  const int nchunks = chunks_of(n);
  // The chunk table is the SpMM one with the mask's columns (n) in the role of k.
  hipLaunchKernelGGL((spmm_chunk_table_kernel<ilog2(kBK)>), dim3(ceil_div(slots, 4)), dim3(256),
                     0, stream, m, n, slots, nchunks, row_indices, row_offsets, column_indices,
                     table, row_ok);
  st = launch_status();
  if (st != 0) return st;
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    const dim3 grid(slots / kBM, ry);
    if (k == 64) {
      hipLaunchKernelGGL(sddmm_tiled_kernel<1>, grid, dim3(kThreads), 0, stream, m, n, nonzeros,
                         slots, nchunks, row_indices, row_offsets, column_indices, table, row_ok,
                         lhs + r0 * lhs_stride, lhs_stride, rhs + r0 * rhs_stride, rhs_stride,
                         out + r0 * out_stride, out_stride);
    } else {
      hipLaunchKernelGGL(sddmm_tiled_kernel<2>, grid, dim3(kThreads), 0, stream, m, n, nonzeros,
                         slots, nchunks, row_indices, row_offsets, column_indices, table, row_ok,
                         lhs + r0 * lhs_stride, lhs_stride, rhs + r0 * rhs_stride, rhs_stride,
                         out + r0 * out_stride, out_stride);
    }
    st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

}  // namespace sputnik_hip
