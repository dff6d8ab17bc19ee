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
  const int slots = slots_of(m), nchunks = chunks_of(n);
  int* row_ok = static_cast<int*>(workspace);
  int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + row_ok_bytes(slots));
  // The chunk table is the SpMM one with the mask's columns (n) in the role of k.
  hipLaunchKernelGGL((spmm_chunk_table_kernel<ilog2(kBK)>), dim3(ceil_div(slots, 4)), dim3(256),
                     0, stream, m, n, slots, nchunks, row_indices, row_offsets, column_indices,
                     table, row_ok);
  int st = launch_status();
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
