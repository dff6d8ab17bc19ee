// LDS-tiled SpMM for NARROW dense operands (any n >= 64 that is a multiple of 4
// and that the 256- / 512-column kernels of spmm_tiled.hip do not take; the last
// column tile may be partial): attention heads (n = head_dim), SparseLinear with
// short sequences, the reference's own test shape n = 72 (tests/test_spmm.py:13).
//
// Same decomposition as spmm_tiled.hip -- B staged per K chunk into LDS by
// direct global->LDS copies, C accumulators in registers for the whole K walk,
// rows dealt in row_indices order, chunk table pre-pass -- but a wavefront
// works on FOUR rows at a time: each 16-lane row of the wave owns one matrix
// row and its 64 output columns (4 per lane).  That is exactly the shape DPP
// `row_newbcast` serves: the 16 lanes of a row group hold 16 consecutive
// entries of THEIR row's (column, value) stream, and entry u is broadcast
// inside the group by the DPP modifier -- four different nonzeros, one per
// group, are processed by every wave instruction (one ds_read_b128 fetches four
// different 256-byte B rows; since the tile's row stride is one full LDS bank
// row, this is conflict-free).  Groups whose row has fewer entries in the
// chunk run the extra steps with a zero value.
//
// Register budget is small here (16 accumulators), so every row keeps TWO
// 16-entry windows per chunk, and the windows of the NEXT chunk are requested
// at the start of the current one into a second register set: they have a
// whole chunk to land and the end-of-chunk `vmcnt(0)` (needed for the B copy)
// costs nothing.  Rows with more than 32 entries in one chunk fetch the rest on
// demand.  The four row groups of a wave run their own number of steps
// (EXEC-masked), so no LDS traffic is spent on the shorter rows' padding.
//
// K SPLIT (round 4).  One product with a narrow dense operand gives few workgroups:
// 4096 x 4096 against n = 72 columns is 32 row blocks x 2 column tiles = 64 workgroups
// for 256 CUs that hold two each (3.3 TFLOP/s in rounds 2-3; the reference's own test
// shape is n = 72, tests/test_spmm.py:13).  The K chunks are therefore dealt to up to 8
// workgroups per tile, each writing its partial tile to the workspace, and a second
// kernel adds the partial tiles in chunk order (fixed order: deterministic) and applies
// the epilogue.  Only for a single replica (a batch fills the chip by itself).
#include <type_traits>

#include "options.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {

namespace {

using namespace tiled;

constexpr int kBN = 64;     // columns of C per workgroup
constexpr int kWaves = 8;   // waves per workgroup
constexpr int kRQ = 4;      // row quads per wave (4 rows each)
constexpr int kBK = 128;    // rows of B per LDS stage
constexpr int kBM = kWaves * kRQ * 4;
constexpr int kThreads = kWaves * kWave;
constexpr int kTileFloats = kBK * kBN;                 // 32 KiB
constexpr int kCopiesPerStage = kBK / 4;               // one wave instruction moves 4 tile rows
constexpr int kCopiesPerWave = kCopiesPerStage / kWaves;
static_assert(kCopiesPerStage % kWaves == 0, "stage copies split evenly over the waves");

// One wave instruction copies tile rows 4j .. 4j+3 (4 x 256 B, lane l -> row
// l/16, bytes (l%16)*16) from B rows kc+4j .. kc+4j+3 into LDS.
__device__ __forceinline__ void stage_chunk64(float* __restrict__ tile,
                                              const float* __restrict__ dense, int n, int k,
                                              int n0, int kc, int wave, int lane) {
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int j = 0; j < kCopiesPerWave; ++j) {
    const int r0 = (wave + j * kWaves) * 4;
    const int src_row = min(kc + r0 + g, k - 1);  // past the end of B: re-read the last row
    // (last, partial column tile: lanes past the row's end re-read its last 16
    // bytes -- never stored -- so that no copy leaves the row, let alone B)
    const unsigned off = static_cast<unsigned>(src_row) * static_cast<unsigned>(n) * 4u +
                         static_cast<unsigned>(min(n0 + i * 4, n - 4)) * 4u;
    lds_dma_row(dense, off, tile + r0 * kBN);
  }
}

__global__ __launch_bounds__(kThreads) void spmm_tiled64_kernel(
    int m, int k, int n, int nonzeros, int slots, int nchunks, int n_tiles,
    const int* __restrict__ row_indices, const float* __restrict__ values,
    int64_t values_stride, const int* __restrict__ column_indices,
    const int* __restrict__ table, const float* __restrict__ dense, int64_t dense_stride,
    float* __restrict__ out, int64_t out_stride, const int* __restrict__ row_ok,
    const int* __restrict__ row_offsets, int debug, Epilogue epi, int ksplits,
    float* __restrict__ partials /* [ksplits][m][n] when ksplits > 1 */) {
  __shared__ float tile[2][kTileFloats];
  const bool dbg_no_compute = debug & 1, dbg_no_stage = debug & 2;

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;

  // (row blocks that stage the same B tile -- same replica and column tile -- run
  // on one XCD: consecutive work indices, see xcd_local_index)
  const unsigned long long work = xcd_local_index();
  const unsigned mblocks = gridDim.x / (n_tiles * ksplits);
  const int mblock = static_cast<int>(work % mblocks);
  const int ntile = static_cast<int>((work / mblocks) % n_tiles);
  const unsigned long long rest = work / (static_cast<unsigned long long>(mblocks) * n_tiles);
  const int split = static_cast<int>(rest % ksplits);     // (one replica when ksplits > 1)
  const int replica = static_cast<int>(rest / ksplits);
  values += replica * values_stride;
  dense += replica * dense_stride;
  out += replica * out_stride;
  // this workgroup's K chunks; with a split the tile goes to the partial buffer as it is
  const int per_split = (nchunks + ksplits - 1) / ksplits;
  const int c_begin = min(split * per_split, nchunks), c_end = min(c_begin + per_split, nchunks);
  if (ksplits > 1) {
    out = partials + static_cast<int64_t>(split) * m * n;
    epi = Epilogue{};
  }
  const int n0 = ntile * kBN;
  const int slot0 = mblock * kBM + wave * (kRQ * 4);  // this wave's first row slot
  const int last = nonzeros - 1;

  // Row blocks whose column indices do not ascend inside rows take the
  // order-independent path (B gathered from L2), one row per 16-lane group.
  if (!block_rows_ok(row_ok, mblock * kBM, kBM)) {
    for (int t = 0; t < kRQ; ++t) {
      const int entry = dealt_index(slot0 + 4 * t + g, slots, kBM);
      if (entry < m && n0 + i * 4 < n) {
        const int row = row_indices[entry];
        const int col = min(n0 + i * 4, n - 4);   // (any n: see the store phase below)
        // (K split: the first split computes the whole row, the others contribute zeros)
        const float4 acc4 = split == 0 ? gather_row_strip(values, column_indices, row_offsets[row],
                                                          row_offsets[row + 1], dense + col, n)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + col) =
            apply_epilogue(acc4, epi, row);
      }
    }
    return;
  }

  float acc[kRQ][4];
#pragma unroll
  for (int t = 0; t < kRQ; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[t][v] = 0.f;

  // Per quad t, this lane's row is slot0 + 4t + g.  Its stream positions at the
  // chunk boundaries c .. c+2 sit in p0 .. p2 (p3 = boundary c+3 is in flight),
  // and its first 32 entries of a chunk (window w, lane i = entry 16w + i) in
  // one of THREE register sets: the windows of chunk c+2 are requested while
  // chunk c is computed, so their (HBM) latency has two chunks to pass.  Every
  // vector-memory operation of the loop is issued from inline asm in a fixed
  // order and number per chunk,
  //   D  kCopiesPerWave LDS-DMA copies of the B tile of chunk c+1
  //   T  kRQ loads of the positions at boundary c+3
  //   W  2*kRQ*kWin loads of the windows of chunk c+2
  // so the waits are counted by hand (vmcnt retires in order): the windows of
  // chunk c were issued two groups ago (two whole groups may stay in flight);
  // at the end of the chunk D and T of this group must have landed and W may
  // stay in flight.  The last chunks issue the same (clamped, unused)
  // operations so that the counts hold for every iteration.
  constexpr int kWin = 2;
  constexpr int kWindowOps = 2 * kRQ * kWin;
  constexpr int kGroupOps = kCopiesPerWave + kRQ + kWindowOps;
  static_assert(2 * kGroupOps <= 63, "vmcnt is a 6-bit counter");
  const unsigned tab_lane = static_cast<unsigned>(slot0 + g) * 4u;
  auto load_positions = [&](int (&p)[kRQ], int boundary) {
    const unsigned row_off =
        static_cast<unsigned>(min(boundary, nchunks)) * static_cast<unsigned>(slots) * 4u + tab_lane;
#pragma unroll
    for (int t = 0; t < kRQ; ++t) p[t] = untracked_load_i32(table, row_off + 16u * t);
  };
  int wc[3][kRQ][kWin];
  float wv[3][kRQ][kWin];
  auto request = [&](auto SET, const int (&start)[kRQ]) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int t = 0; t < kRQ; ++t)
#pragma unroll
      for (int w = 0; w < kWin; ++w) {
        const unsigned off = static_cast<unsigned>(min(start[t] + 16 * w + i, last)) * 4u;
        wc[set][t][w] = untracked_load_i32(column_indices, off);
        wv[set][t][w] = untracked_load_f32(values, off);
      }
  };
  auto arrived = [&](auto SET) {  // ties the set's registers to the wait before them
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int t = 0; t < kRQ; ++t)
#pragma unroll
      for (int w = 0; w < kWin; ++w) {
        tie_reg(wc[set][t][w]);
        tie_reg(wv[set][t][w]);
      }
  };

  int p0[kRQ], p1[kRQ], p2[kRQ], p3[kRQ];
  load_positions(p0, c_begin);
  load_positions(p1, c_begin + 1);
  load_positions(p2, c_begin + 2);
  wait_vm<0>();
#pragma unroll
  for (int t = 0; t < kRQ; ++t) {
    tie_reg(p0[t]);
    tie_reg(p1[t]);
    tie_reg(p2[t]);
  }
  stage_chunk64(tile[c_begin & 1], dense, n, k, n0, min(c_begin, nchunks - 1) * kBK, wave, lane);
  request(std::integral_constant<int, 0>{}, p0);
  request(std::integral_constant<int, 1>{}, p1);
  wait_vm<0>();
  arrived(std::integral_constant<int, 0>{});
  arrived(std::integral_constant<int, 1>{});
  __syncthreads();

  auto chunk = [&](auto R, int c) {
    constexpr int r = decltype(R)::value;  // = c % 3: the register set of this chunk's windows
    const int buf = c & 1;
    // D, T, W (the timing experiment without staging still issues the copies,
    // from chunk 0, so that the operation count the waits assume is unchanged)
    stage_chunk64(tile[buf ^ 1], dense, n, k, n0,
                  dbg_no_stage ? 0 : min(c + 1, nchunks - 1) * kBK, wave, lane);
    load_positions(p3, c + 3);
    request(std::integral_constant<int, (r + 2) % 3>{}, p2);
    wait_vm<2 * kGroupOps>();
    arrived(R);

    const char* __restrict__ lane_base = reinterpret_cast<const char*>(&tile[buf][0] + i * 4);
    const int kc = c * kBK;
#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      const int cnt = dbg_no_compute ? 0 : p1[t] - p0[t];  // this group's row; the same in its 16 lanes
      // One 16-entry window of this group's row: groups of four entries, each
      // group of lanes stopping at its own row's count (EXEC-masked).
      auto window = [&](int ecol, float eval, int w0) {
        const int left = cnt - w0;  // entries of this row at or after the window start
        const bool valid = i < left;
        const int roff = valid ? ((ecol - kc) * (kBN * 4)) : 0;
        const float rval = valid ? eval : 0.f;
        if (left > 0) dpp_group4<0>(acc[t], roff, rval, lane_base);
        if (left > 4) dpp_group4<4>(acc[t], roff, rval, lane_base);
        if (left > 8) dpp_group4<8>(acc[t], roff, rval, lane_base);
        if (left > 12) dpp_group4<12>(acc[t], roff, rval, lane_base);
      };
#pragma unroll
      for (int w = 0; w < kWin; ++w) window(wc[r][t][w], wv[r][t][w], 16 * w);
      // longer rows (more than 32 entries inside one chunk): fetch on demand
      // (ordinary loads: the compiler's own wait drains everything in flight)
      const int longest = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                              max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
      for (int w0 = 16 * kWin; w0 < longest; w0 += 16) {
        const int idx = min(p0[t] + w0 + i, last);
        window(column_indices[idx], values[idx], w0);
      }
    }

    wait_vm<kWindowOps>();  // D and T of this group have landed; W stays in flight
#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      tie_reg(p3[t]);
      p0[t] = p1[t];
      p1[t] = p2[t];
      p2[t] = p3[t];
    }
    __syncthreads();  // the next tile is there for every wave, and the current buffer is free
  };
  for (int c0 = c_begin; c0 < c_end; c0 += 3) {   // (register set of a chunk: (c - c_begin) % 3)
    chunk(std::integral_constant<int, 0>{}, c0);
    if (c0 + 1 < c_end) chunk(std::integral_constant<int, 1>{}, c0 + 1);
    if (c0 + 2 < c_end) chunk(std::integral_constant<int, 2>{}, c0 + 2);
  }
  wait_vm<0>();  // nothing may be in flight (LDS-DMA!) when the wave ends
  // the windows of the two chunks past the end were requested and never read:
  // keep their registers allocated up to the wait (see spmm_tiled.hip)
  arrived(std::integral_constant<int, 0>{});
  arrived(std::integral_constant<int, 1>{});
  arrived(std::integral_constant<int, 2>{});

#pragma unroll
  for (int t = 0; t < kRQ; ++t) {
    const int entry = dealt_index(slot0 + 4 * t + g, slots, kBM);
    if (entry < m && n0 + i * 4 < n) {   // (the last column tile may be partial)
      const int row = row_indices[entry];
      // ANY n (src/spmm_cuda.cu:32): a lane whose four columns would cross the end of
      // the row works on the row's last four columns instead, in the copies of B and
      // here alike (it stores up to three of its neighbour's values again); the 16-byte
      // accesses are then dword aligned, which is all global memory asks for
      const int col = min(n0 + i * 4, n - 4);
      *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + col) =
          apply_epilogue(make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]), epi, row);
    }
  }
}

inline int slots_of(int m) { return ceil_div(m, kBM) * kBM; }
inline int chunks_of(int k) { return ceil_div(k, kBK); }

// out[row][c] = epilogue(sum over the splits, in chunk order, of partials[s][row][c])
__global__ __launch_bounds__(256) void spmm_tiled64_sum_kernel(int64_t quads /* m * n / 4 */, int n,
                                                               int ksplits,
                                                               const float* __restrict__ partials,
                                                               float* __restrict__ out, Epilogue epi) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= quads) return;
  float4 sum = reinterpret_cast<const float4*>(partials)[j];
  for (int s = 1; s < ksplits; ++s) {
    const float4 v = reinterpret_cast<const float4*>(partials)[static_cast<int64_t>(s) * quads + j];
    sum.x += v.x;
    sum.y += v.y;
    sum.z += v.z;
    sum.w += v.w;
  }
  reinterpret_cast<float4*>(out)[j] = apply_epilogue(sum, epi, static_cast<int>(j * 4 / n));
}

// Splits of the K walk for ONE replica: up to two workgroups per CU (512), at least two
// chunks each, at most 8 (each split writes a partial tile); n a multiple of 4 (the
// partial tiles are added in 16-byte pieces, a piece inside one row).  Measured
// (tools/narrow_n_bench.py, density 0.1, us of the whole call, splits 1 / 2 / 4 / 8 / 16):
//   4096^2 x 72   113 / 69 / 46 / 40 / 48      4096^2 x 200  114 / 74 / 59 / 66 / 78
//   2048^2 x 72    59 / 38 / 28 / 24 / 28      2048^2 x 256   58 / 41 / 32 / 33 / 39
inline int ksplits_of(int m, int k, int n) {
  if (n % 4 != 0) return 1;
  const int64_t groups = static_cast<int64_t>(slots_of(m) / kBM) * ceil_div(n, kBN);
  const int forced = options().spmm_tile;   // (developer knob, 16 + s: s splits)
  if (forced >= 16) return max(1, min(forced - 16, chunks_of(k)));
  int splits = 1;
  while (splits < 8 && groups * splits * 2 <= 512 && chunks_of(k) >= 4 * splits) splits *= 2;
  return splits;
}

}  // namespace

// (for the dispatcher: a single product that the K split spreads over the chip)
int spmm_tiled64_ksplits(int m, int k, int n) { return ksplits_of(m, k, n); }

bool spmm_tiled64_applicable(int m, int k, int n, int nonzeros) {
  // B rows are addressed with 32-bit byte offsets; enough work per staged tile.
  return n >= kBN && k >= 32 && m >= 16 && nonzeros >= 4 * static_cast<int64_t>(m) &&
         static_cast<int64_t>(k) * n * 4 < (int64_t{1} << 32);
}

namespace {
inline size_t tables_bytes(int m, int k) {
  return (row_ok_bytes(slots_of(m)) + sizeof(int) * static_cast<size_t>(chunks_of(k) + 1) * slots_of(m) +
          255) / 256 * 256;
}
}  // namespace

// row order words + chunk table [+ the partial tiles of a K split]
size_t spmm_tiled64_workspace_bytes(int m, int k, int n) {
  const int splits = ksplits_of(m, k, n);
  return tables_bytes(m, k) +
         (splits > 1 ? sizeof(float) * static_cast<size_t>(splits) * m * n : size_t{0});
}

int spmm_tiled64_plan(int m, int k, const int* row_indices, const int* row_offsets,
                      const int* column_indices, void* workspace, hipStream_t stream) {
  const int slots = slots_of(m);
  int* row_ok = static_cast<int*>(workspace);
  int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + row_ok_bytes(slots));
  hipLaunchKernelGGL((spmm_chunk_table_kernel<kBK>), dim3(ceil_div(slots, 4)), dim3(256),
                     0, stream, m, k, slots, kBM, chunks_of(k), row_indices, row_offsets,
                     column_indices, table, row_ok);
  return launch_status();
}

int spmm_tiled64_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                      const float* values, int64_t values_stride, const int* row_offsets,
                      const int* column_indices, const float* dense, int64_t dense_stride,
                      float* out, int64_t out_stride, const void* workspace,
                      hipStream_t stream, Epilogue epi) {
  const int slots = slots_of(m);
  const int* row_ok = static_cast<const int*>(workspace);
  const int* table =
      reinterpret_cast<const int*>(static_cast<const char*>(workspace) + row_ok_bytes(slots));
  const int n_tiles = ceil_div(n, kBN);
  const int debug = options().spmm_debug;  // timing experiments only
  // (debug bit 6: no K split -- the one-workgroup-per-tile form of rounds 1-3)
  const int splits = replicas == 1 && !(debug & 64) && aligned_to(out, 16) ? ksplits_of(m, k, n) : 1;
  float* partials = splits > 1 ? reinterpret_cast<float*>(const_cast<char*>(
                                     static_cast<const char*>(workspace) + tables_bytes(m, k)))
                               : nullptr;
  hipLaunchKernelGGL(spmm_tiled64_kernel, dim3((slots / kBM) * n_tiles * splits, replicas),
                     dim3(kThreads), 0, stream, m, k, n, nonzeros, slots, chunks_of(k), n_tiles,
                     row_indices, values, values_stride, column_indices, table, dense, dense_stride,
                     out, out_stride, row_ok, row_offsets, debug, epi, splits, partials);
  int st = launch_status();
  if (st != 0 || splits == 1) return st;
  const int64_t quads = static_cast<int64_t>(m) * n / 4;
  hipLaunchKernelGGL(spmm_tiled64_sum_kernel, dim3(static_cast<unsigned>(ceil_div64(quads, 256))),
                     dim3(256), 0, stream, quads, n, splits, partials, out, epi);
  return launch_status();
}

}  // namespace sputnik_hip
