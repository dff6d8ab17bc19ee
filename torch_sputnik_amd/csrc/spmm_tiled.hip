// LDS-tiled SpMM for gfx950: C[m,n] = A_csr[m,k] * B[k,n], fp32, vector FMA
// only (no MFMA: the operand pattern is an irregular gather).
//
// Why a second kernel.  In the row-gather kernel (spmm.hip) every FMA pulls
// its B operand through L2: nnz*n*4 bytes of cache traffic (27.5 GB at
// 4096^3, density 0.1), which caps it near 15 TFLOP/s.  Here a workgroup
// owns BM rows x BN columns of C and walks K in chunks of BK rows of B that
// are staged ONCE per workgroup into LDS (direct global->LDS loads, double
// buffered) and then gathered from LDS by all BM rows:
//
//   * B traffic from L2 drops from nnz*n*4 to (m/BM)*k*n*4 bytes;
//   * each wave owns RPW rows whose accumulators stay in registers for the
//     whole K walk (RPW*BN/64 VGPRs), so C is written exactly once and no
//     atomics or partial sums exist -> bitwise reproducible;
//   * a row's (column, value) stream is wave-uniform, so it is read with
//     SCALAR loads (s_load through the scalar cache) and the value enters
//     v_fma_f32 as an SGPR operand: the vector memory path and LDS carry
//     only B.  Per nonzero and lane: one ds_read_b128 + 4 FMA.
//
// The binding limit is LDS bandwidth: one B dword per FMA, 256 B/clk/CU
// -> 64 FMA/clk/CU = 78.6 TFLOP/s chip-wide (= the plain v_fma_f32 rate).
//
// Rows are dealt to waves in `row_indices` order (similar lengths together).
// Splitting a row's nonzeros by K chunk needs the column indices of a row to
// ascend; a small pre-pass builds, per row and chunk boundary, the position
// of the first nonzero at or past the boundary (the "chunk table", in the
// caller's workspace) and verifies the order.  If any row is not ascending
// the pre-pass clears a device flag: this kernel then exits at once and the
// row-gather kernel, launched behind it with the opposite test, does the work.
// No host synchronisation is involved.
#include <stdlib.h>

#include "common.h"
#include "wave_utils.h"

namespace sputnik_hip {

int spmm_rowgather_launch(int m, int n, int replicas, const int* row_indices,
                          const float* values, int64_t values_stride, const int* row_offsets,
                          const int* column_indices, const float* dense, int64_t dense_stride,
                          float* out, int64_t out_stride, const int* skip_flag,
                          hipStream_t stream);

namespace {

constexpr int kFlagBytes = 256;  // flag word + padding so the table stays aligned

#define AS_GLOBAL(p) ((__attribute__((address_space(1))) void*)(p))
#define AS_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------
// Pre-pass: chunk table + order check.  One wave per row slot.
// table[c * slots + slot], c in [0, nchunks]: index of the first nonzero of
// row row_indices[slot] whose column is >= c*BK (row end if none).  Padding
// slots (slot >= m) get 0 everywhere, i.e. empty rows.
// ---------------------------------------------------------------------------
template <int BK_LOG2>
__global__ __launch_bounds__(256) void spmm_chunk_table_kernel(
    int m, int k, int slots, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ table, int* __restrict__ sorted_flag) {
  const int lane = threadIdx.x % kWave;
  const int slot = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
  if (slot >= slots) return;
  if (slot >= m) {
    for (int c = lane; c <= nchunks; c += kWave) table[static_cast<int64_t>(c) * slots + slot] = 0;
    return;
  }
  const int row = row_indices[slot];
  const int p0 = row_offsets[row];
  const int p1 = row_offsets[row + 1];
  bool ok = true;
  for (int base = p0; base < p1; base += kWave) {
    const int p = base + lane;
    if (p < p1) {
      const int cur = column_indices[p];
      const int prev = (p > p0) ? column_indices[p - 1] : -1;
      if (cur <= prev || cur >= k) {
        ok = false;
      } else {
        const int cb = cur >> BK_LOG2;
        const int pb = (prev < 0) ? -1 : (prev >> BK_LOG2);
        for (int c = pb + 1; c <= cb; ++c) table[static_cast<int64_t>(c) * slots + slot] = p;
      }
    }
  }
  int last = -1;
  if (p1 > p0) {
    const int lc = column_indices[p1 - 1];
    last = (lc >= 0 && lc < k) ? (lc >> BK_LOG2) : nchunks;
  }
  for (int c = last + 1 + lane; c <= nchunks; c += kWave)
    table[static_cast<int64_t>(c) * slots + slot] = p1;
  if (!ok) *sorted_flag = 0;
}

// ---------------------------------------------------------------------------
// Main kernel.
// ---------------------------------------------------------------------------
template <int BN, int WAVES, int RPW, int BK>
struct TileConfig {
  static constexpr int kBN = BN;        // columns of C per workgroup
  static constexpr int kWaves = WAVES;  // waves per workgroup
  static constexpr int kRPW = RPW;      // rows of C per wave
  static constexpr int kBK = BK;        // rows of B per LDS stage
  static constexpr int kBM = WAVES * RPW;
  static constexpr int kVec = BN / kWave;  // floats per lane: 4 -> ds_read_b128
  static constexpr int kThreads = WAVES * kWave;
  static constexpr int kStageRowsPerWave = BK / WAVES;
  static_assert(BN == 256, "one 1 KiB LDS-DMA piece per B row");
  static_assert(BK % WAVES == 0, "stage rows split evenly over the waves");
};

// Direct global->LDS copy of one 1 KiB row segment (64 lanes x 16 B):
// LDS destination = M0 + lane*16, global source = per-lane address.
// Written as inline asm on purpose: for the builtin form hipcc treats the
// copy as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of every
// following ds_read, which would serialise the prefetch of the next stage
// with the compute on the current one.  The waits are placed by hand instead
// (wait_stage() before the barrier that publishes a stage).
__device__ __forceinline__ void lds_dma_row(const float* src, const float* lds_dst) {
  const unsigned lds_addr =
      static_cast<unsigned>(reinterpret_cast<uintptr_t>(AS_LDS(const_cast<float*>(lds_dst))));
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off"
      :
      : "s"(lds_addr), "v"(src)
      : "memory", "m0");
}

__device__ __forceinline__ void wait_stage() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename Cfg>
__device__ __forceinline__ void stage_chunk(float* __restrict__ tile, const float* __restrict__ dense,
                                            int n, int k, int n0, int kc, int wave, int lane) {
  // Wave w copies B rows kc + w, kc + w + WAVES, ...; one wave instruction
  // moves one row segment straight into the row-major tile row.
#pragma unroll
  for (int i = 0; i < Cfg::kStageRowsPerWave; ++i) {
    const int r = wave + i * Cfg::kWaves;
    if (kc + r < k)
      lds_dma_row(dense + static_cast<int64_t>(kc + r) * n + n0 + lane * 4, tile + r * Cfg::kBN);
  }
}

// One nonzero against the staged tile: acc[0..3] += a * tile[j - kc][lane*4 .. +3].
#define SPUTNIK_HIP_FMA4(ACC, A, B)          \
  do {                                       \
    (ACC)[0] = fmaf((A), (B).x, (ACC)[0]);   \
    (ACC)[1] = fmaf((A), (B).y, (ACC)[1]);   \
    (ACC)[2] = fmaf((A), (B).z, (ACC)[2]);   \
    (ACC)[3] = fmaf((A), (B).w, (ACC)[3]);   \
  } while (0)

// MODE 0: the row's (column, value) stream is read with scalar loads.
// MODE 1: each row keeps the next 64 entries of its stream in two VGPRs
//         (lane u = entry u), prefetched one K chunk ahead with vector loads
//         and handed out with v_readlane; per (row, chunk) this costs two
//         vector loads instead of a chain of dependent scalar loads.
template <typename Cfg, int MODE>
__global__ __launch_bounds__(Cfg::kThreads) void spmm_tiled_kernel(
    int m, int k, int n, int nonzeros, int slots, int nchunks, int n_tiles,
    const int* __restrict__ row_indices, const float* __restrict__ values,
    int64_t values_stride, const int* __restrict__ column_indices,
    const int* __restrict__ table, const float* __restrict__ dense, int64_t dense_stride,
    float* __restrict__ out, int64_t out_stride, const int* __restrict__ sorted_flag) {
  if (*sorted_flag == 0) return;  // unsorted columns: the row-gather kernel runs instead

  constexpr int BN = Cfg::kBN, BK = Cfg::kBK, RPW = Cfg::kRPW, VEC = Cfg::kVec;
  static_assert(BK <= kWave, "a row has at most BK <= 64 entries per chunk (MODE 1 window)");
  __shared__ float tile[2][BK * BN];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);

  // Workgroup -> (row block, column tile).  Consecutive workgroup ids are
  // dealt round-robin to the 8 XCDs; give each XCD a contiguous set of column
  // tiles so the B panels its workgroups stage are shared in that XCD's L2.
  const int bid = blockIdx.x;
  int ntile, mblock;
  if (n_tiles % 8 == 0) {
    const int per_xcd = n_tiles / 8;
    const int xcd = bid % 8, i = bid / 8;
    ntile = xcd * per_xcd + i % per_xcd;
    mblock = i / per_xcd;
  } else {
    ntile = bid % n_tiles;
    mblock = bid / n_tiles;
  }
  const int replica = blockIdx.y;
  values += replica * values_stride;
  dense += replica * dense_stride;
  out += replica * out_stride;

  const int n0 = ntile * BN;
  const int slot0 = mblock * Cfg::kBM + wave * RPW;

  float acc[RPW][VEC];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[r][v] = 0.f;

  const float* lane_tile = &tile[0][0] + lane * VEC;

  if constexpr (MODE == 0) {
    int ps[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) ps[r] = table[slot0 + r];

    stage_chunk<Cfg>(tile[0], dense, n, k, n0, 0, wave, lane);
    wait_stage();
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
      const int buf = c & 1;
      if (c + 1 < nchunks)
        stage_chunk<Cfg>(tile[buf ^ 1], dense, n, k, n0, (c + 1) * BK, wave, lane);
      const int* __restrict__ next_ptr = table + static_cast<int64_t>(c + 1) * slots + slot0;
      const float* __restrict__ btile = lane_tile + buf * (BK * BN);
      const int kc = c * BK;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int pe = next_ptr[r];
        int p = ps[r];
        for (; p + 4 <= pe; p += 4) {
          int j[4];
          float a[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            j[u] = column_indices[p + u];
            a[u] = values[p + u];
          }
          float4 b[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            b[u] = *reinterpret_cast<const float4*>(btile + (j[u] - kc) * BN);
#pragma unroll
          for (int u = 0; u < 4; ++u) SPUTNIK_HIP_FMA4(acc[r], a[u], b[u]);
        }
        for (; p < pe; ++p) {
          const int j = column_indices[p];
          const float a = values[p];
          const float4 b = *reinterpret_cast<const float4*>(btile + (j - kc) * BN);
          SPUTNIK_HIP_FMA4(acc[r], a, b);
        }
        ps[r] = pe;
      }
      wait_stage();     // this wave's share of the next stage has landed in LDS
      __syncthreads();  // everyone's has, and the current buffer is free to overwrite
    }
  } else {
    // Lane r (< RPW) of these holds the stream position of row r at the
    // start of the current chunk / of the next chunk / of the one after.
    const int ptr_lane = min(lane, RPW - 1);
    const int* __restrict__ my_table = table + slot0 + ptr_lane;
    int v_ps = my_table[0];
    int v_pe = my_table[slots];
    const int last = nonzeros - 1;

    int vcol[RPW];
    float vval[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int idx = min(__builtin_amdgcn_readlane(v_ps, r) + lane, last);
      vcol[r] = column_indices[idx];
      vval[r] = values[idx];
    }

    stage_chunk<Cfg>(tile[0], dense, n, k, n0, 0, wave, lane);
    wait_stage();
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
      const int buf = c & 1;
      if (c + 1 < nchunks)
        stage_chunk<Cfg>(tile[buf ^ 1], dense, n, k, n0, (c + 1) * BK, wave, lane);
      const int v_pe_next =
          (c + 2 <= nchunks) ? my_table[static_cast<int64_t>(c + 2) * slots] : 0;
      const float* __restrict__ btile = lane_tile + buf * (BK * BN);
      const int kc = c * BK;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int s_pe = __builtin_amdgcn_readlane(v_pe, r);
        const int cnt = s_pe - __builtin_amdgcn_readlane(v_ps, r);
        int u = 0;
        for (; u + 4 <= cnt; u += 4) {
          int j[4];
          float a[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            j[t] = __builtin_amdgcn_readlane(vcol[r], u + t);
            a[t] = readlane_f32(vval[r], u + t);
          }
          float4 b[4];
#pragma unroll
          for (int t = 0; t < 4; ++t)
            b[t] = *reinterpret_cast<const float4*>(btile + (j[t] - kc) * BN);
#pragma unroll
          for (int t = 0; t < 4; ++t) SPUTNIK_HIP_FMA4(acc[r], a[t], b[t]);
        }
        if (cnt & 2) {
          const int j0 = __builtin_amdgcn_readlane(vcol[r], u);
          const int j1 = __builtin_amdgcn_readlane(vcol[r], u + 1);
          const float a0 = readlane_f32(vval[r], u), a1 = readlane_f32(vval[r], u + 1);
          const float4 b0 = *reinterpret_cast<const float4*>(btile + (j0 - kc) * BN);
          const float4 b1 = *reinterpret_cast<const float4*>(btile + (j1 - kc) * BN);
          SPUTNIK_HIP_FMA4(acc[r], a0, b0);
          SPUTNIK_HIP_FMA4(acc[r], a1, b1);
          u += 2;
        }
        if (cnt & 1) {
          const int j0 = __builtin_amdgcn_readlane(vcol[r], u);
          const float a0 = readlane_f32(vval[r], u);
          const float4 b0 = *reinterpret_cast<const float4*>(btile + (j0 - kc) * BN);
          SPUTNIK_HIP_FMA4(acc[r], a0, b0);
        }
        // This row's window for the next chunk (in flight until the barrier).
        if (c + 1 < nchunks) {
          const int idx = min(s_pe + lane, last);
          vcol[r] = column_indices[idx];
          vval[r] = values[idx];
        }
      }
      v_ps = v_pe;
      v_pe = v_pe_next;
      wait_stage();     // next stage of B and the next entry windows have landed
      __syncthreads();  // ... for every wave, and the current buffer is free
    }
  }

#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int slot = slot0 + r;
    if (slot < m) {
      const int row = row_indices[slot];
      *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + n0 + lane * VEC) =
          make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
  }
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

struct Plan {
  bool use;
  int bm, bk, slots, nchunks, n_tiles;
  size_t table_bytes;
};

template <typename Cfg>
Plan make_plan(int m, int k, int n) {
  Plan p;
  p.bm = Cfg::kBM;
  p.bk = Cfg::kBK;
  p.slots = ceil_div(m, Cfg::kBM) * Cfg::kBM;
  p.nchunks = ceil_div(k, Cfg::kBK);
  p.n_tiles = n / Cfg::kBN;
  p.table_bytes = sizeof(int) * static_cast<size_t>(p.nchunks + 1) * p.slots;
  p.use = true;
  return p;
}

using CfgLarge = TileConfig<256, 16, 16, 64>;  // 256 x 256 tile of C per workgroup

inline bool tiled_applicable(int m, int k, int n, int nonzeros) {
  // Needs full column tiles, and enough work per row block to amortise staging
  // B (each workgroup stages k x 256 floats): mean row length >= 16.
  return n % CfgLarge::kBN == 0 && k >= CfgLarge::kBK && m >= 64 &&
         nonzeros >= 16 * static_cast<int64_t>(m);
}

}  // namespace

size_t spmm_tiled_workspace_bytes(int m, int k, int n, int nonzeros) {
  if (!tiled_applicable(m, k, n, nonzeros)) return 0;
  return kFlagBytes + make_plan<CfgLarge>(m, k, n).table_bytes;
}

// Pre-pass only: topology -> chunk table + order flag in `workspace`.  Depends
// on the topology alone, so a caller with a static pattern can run it once and
// reuse the workspace for any number of spmm_tiled_exec calls.
int spmm_tiled_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                    const int* row_offsets, const int* column_indices, void* workspace,
                    size_t workspace_bytes, hipStream_t stream, bool* planned) {
  *planned = false;
  if (!tiled_applicable(m, k, n, nonzeros)) return 0;
  using Cfg = CfgLarge;
  const Plan plan = make_plan<Cfg>(m, k, n);
  if (workspace == nullptr || workspace_bytes < kFlagBytes + plan.table_bytes ||
      !aligned_to(workspace, 16))
    return 0;
  int* flag = static_cast<int*>(workspace);
  int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + kFlagBytes);
  const hipError_t e = hipMemsetAsync(flag, 1, sizeof(int), stream);  // nonzero = "sorted so far"
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL((spmm_chunk_table_kernel<ilog2(Cfg::kBK)>), dim3(ceil_div(plan.slots, 4)),
                     dim3(256), 0, stream, m, k, plan.slots, plan.nchunks, row_indices,
                     row_offsets, column_indices, table, flag);
  *planned = true;
  return launch_status();
}

// Main kernel (+ the flag-gated row-gather fallback) on a planned workspace.
int spmm_tiled_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                    const float* values, int64_t values_stride, const int* row_offsets,
                    const int* column_indices, const float* dense, int64_t dense_stride,
                    float* out, int64_t out_stride, const void* workspace,
                    size_t workspace_bytes, hipStream_t stream, bool* handled) {
  *handled = false;
  if (!tiled_applicable(m, k, n, nonzeros)) return 0;
  using Cfg = CfgLarge;
  const Plan plan = make_plan<Cfg>(m, k, n);
  if (workspace == nullptr || workspace_bytes < kFlagBytes + plan.table_bytes ||
      !aligned_to(workspace, 16))
    return 0;
  if (!aligned_to(dense, 16) || !aligned_to(out, 16) || dense_stride % 4 != 0 ||
      out_stride % 4 != 0 || replicas > kMaxGridYZ)
    return 0;
  const int* flag = static_cast<const int*>(workspace);
  const int* table =
      reinterpret_cast<const int*>(static_cast<const char*>(workspace) + kFlagBytes);

  const int blocks = (plan.slots / Cfg::kBM) * plan.n_tiles;
  static const int mode = [] {
    const char* e = getenv("SPUTNIK_HIP_SPMM_MODE");  // developer knob, see DESIGN.md
    return e ? atoi(e) : 1;
  }();
  if (mode == 0) {
    hipLaunchKernelGGL((spmm_tiled_kernel<Cfg, 0>), dim3(blocks, replicas), dim3(Cfg::kThreads),
                       0, stream, m, k, n, nonzeros, plan.slots, plan.nchunks, plan.n_tiles,
                       row_indices, values, values_stride, column_indices, table, dense,
                       dense_stride, out, out_stride, flag);
  } else {
    hipLaunchKernelGGL((spmm_tiled_kernel<Cfg, 1>), dim3(blocks, replicas), dim3(Cfg::kThreads),
                       0, stream, m, k, n, nonzeros, plan.slots, plan.nchunks, plan.n_tiles,
                       row_indices, values, values_stride, column_indices, table, dense,
                       dense_stride, out, out_stride, flag);
  }
  int st = launch_status();
  if (st != 0) return st;

  // Fallback for unsorted column indices: its skip test is "flag != 0", i.e. it
  // exits at once when the tiled kernel did the work.
  st = spmm_rowgather_launch(m, n, replicas, row_indices, values, values_stride, row_offsets,
                             column_indices, dense, dense_stride, out, out_stride, flag, stream);
  *handled = true;
  return st;
}

}  // namespace sputnik_hip
