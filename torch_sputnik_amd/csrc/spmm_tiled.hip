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
//   * a row's (column, value) stream is wave-uniform.  Its next entries are
//     prefetched into two VGPRs (a "window": 64 entries, lane = entry, replicated
//     16 at a time with one ds_bpermute pair -- or, for short segments, 16
//     entries that the load itself replicates into every 16-lane row) and
//     handed out with DPP row_newbcast: per nonzero one v_mov_b64_dpp (tile
//     offset and value as a register pair), one v_add_u32 (LDS address), one or
//     two ds_read_b128 and two or four v_pk_fma_f32 -- no VGPR->SGPR traffic.
//     (Two earlier forms, scalar loads of the stream and v_readlane hand-out,
//     were measured slower and removed: DESIGN.md section 3.1.)
//
// Tiles: 128 x 512 (eight columns per lane: one broadcast and address feed two
// LDS reads and eight FMAs, the best instruction mix; for problems that give
// every CU a tile), 256 x 256, 128 x 256, 64 x 256 (TileConfig below).
//
// The roof is LDS bandwidth: one B dword per FMA, 256 B/clk/CU -> 64
// FMA/clk/CU = 78.6 TFLOP/s chip-wide (= the plain v_fma_f32 rate).  What
// binds in practice, at the densities of the headline benchmark, is the serial
// instruction stream of a (row, chunk) visit -- see the comments in the main
// loop: everything scalar that a vector instruction or no instruction can do
// has been moved or removed.
//
// Rows are dealt to workgroups interleaved (dealt_index), so that every
// workgroup gets the same mix of long and short rows whatever the order of the
// caller's `row_indices`.  Splitting a row's nonzeros by K chunk needs the
// column indices of a row to ascend; a small pre-pass builds, per row and chunk
// boundary, the position of the first nonzero at or past the boundary (the
// "chunk table", in the caller's workspace) and records per row whether its
// columns ascend.  A workgroup that finds a non-ascending row in its block
// takes an order-independent path (B gathered from L2) inside the same launch:
// no host synchronisation, no second kernel.
#include <algorithm>

#include "options.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {

bool spmm_tiled64_applicable(int m, int k, int n, int nonzeros);
size_t spmm_tiled64_workspace_bytes(int m, int k, int n);
int spmm_tiled64_ksplits(int m, int k, int n);
int spmm_tiled64_plan(int m, int k, const int* row_indices, const int* row_offsets,
                      const int* column_indices, void* workspace, hipStream_t stream);
int spmm_tiled64_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                      const float* values, int64_t values_stride, const int* row_offsets,
                      const int* column_indices, const float* dense, int64_t dense_stride,
                      float* out, int64_t out_stride, const void* workspace,
                      hipStream_t stream, Epilogue epi);

// Flat-stream form of the 128 x 512 tile (spmm_flat.hip).
bool spmm_flat_applicable(int m, int k, int n, int nonzeros);
int64_t spmm_flat_tiles(int m, int n);
const char* spmm_flat_kernel_name(int m, int k, int nonzeros);
size_t spmm_flat_workspace_bytes(int m, int k, int n, int nonzeros);
int spmm_flat_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                   const int* row_offsets, const int* column_indices, void* workspace,
                   hipStream_t stream);
int spmm_flat_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                   const float* values, int64_t values_stride, const int* row_offsets,
                   const int* column_indices, const float* dense, int64_t dense_stride, float* out,
                   int64_t out_stride, const void* workspace, hipStream_t stream, Epilogue epi);

namespace {

using namespace tiled;

// ---------------------------------------------------------------------------
// Main kernel.
// ---------------------------------------------------------------------------
// Row slots are dealt to workgroups in runs of this many (dealt_index): the
// largest tile's row count, so that the 128- and 64-row tiles, which share the
// table, get balanced halves / quarters of such a run.
constexpr int kDealPer = 256;

template <int BN, int WAVES, int RPW, int BK>
struct TileConfig {
  static constexpr int kBN = BN;        // columns of C per workgroup
  static constexpr int kWaves = WAVES;  // waves per workgroup
  static constexpr int kRPW = RPW;      // rows of C per wave
  static constexpr int kBK = BK;        // rows of B per LDS stage
  static constexpr int kBM = WAVES * RPW;
  static constexpr int kVec = BN / kWave;  // floats per lane: 4 per 256-column piece
  static constexpr int kPieces = BN / 256;  // 1 KiB LDS-DMA pieces (= ds_read_b128 per lane) per B row
  static constexpr int kThreads = WAVES * kWave;
  static constexpr int kStageOps = BK * kPieces / WAVES;  // LDS-DMA copies per wave and stage
  // Entries per LDS round trip in the short-segment variant: the strips of a
  // batch are 32 registers with 8 columns per lane and batches of four; the
  // 8-row tiles with 4 columns per lane have the registers for batches of eight.
  static constexpr int kBatch = (RPW == 8 && BN == 256) ? 8 : 4;
  static_assert(BN == 256 || BN == 512, "one or two 1 KiB LDS-DMA pieces per B row");
  static_assert((BK * kPieces) % WAVES == 0, "the pieces of a stage split evenly over the waves");
};

template <typename Cfg>
__device__ __forceinline__ void stage_chunk(float* __restrict__ tile, const float* __restrict__ dense,
                                            int n, int k, int kc, int wave,
                                            const unsigned (&lane_byte_offset)[Cfg::kPieces]) {
  // The stage is BK * kPieces pieces of 1 KiB (256 columns of one B row); wave w
  // copies pieces w, w + WAVES, ...: one wave instruction moves one piece
  // (lane_byte_offset[h] = byte offset of the lane's 16 bytes of piece h inside a
  // B row: (n0 + h*256 + lane*4) * 4, clamped to the row's last 16 bytes in the
  // last, partial column tile -- such columns are never stored, and no copy
  // leaves the row, let alone B) straight into the row-major tile row.  Rows past the end
  // of B (last, partial chunk) re-read row k-1: no nonzero refers to them, and
  // every wave then issues exactly kStageOps copies per stage, which the counted
  // vmcnt waits of the main loop rely on.
#pragma unroll
  for (int i = 0; i < Cfg::kStageOps; ++i) {
    const int piece = wave + i * Cfg::kWaves;
    const int r = piece / Cfg::kPieces, h = piece % Cfg::kPieces;
    const int src_row = min(kc + r, k - 1);
    lds_dma_row(dense + static_cast<int64_t>(src_row) * n, lane_byte_offset[h],
                tile + r * Cfg::kBN + h * 256);
  }
}

// Main loop (see the file header).
//
// Every vector-memory operation inside the loop is issued from inline asm, in
// a fixed order and number per chunk, so the waits can be counted by hand:
//   A  kStageOps (S) LDS-DMA copies of B rows for chunk c+1
//   B  the rows' stream positions one chunk ahead: a SCALAR load into SGPRs
//      (not counted by vmcnt)
//   C  after each row r: 2 loads = the entry window of the row that is D rows
//      further down the walk (row r+D of this chunk, or row r+D-RPW of the
//      next one); D windows (2*D VGPRs) are live at any time.
// A window is consumed D rows after it was requested.  In between exactly
// D-1 other windows were requested (2*(D-1) operations), plus A if the
// chunk boundary was crossed (rows r < D): `vmcnt` with that count retires it
// and nothing newer.  At the end of a chunk `vmcnt(2*RPW)` retires A (the B
// tile of the next chunk) and leaves B and C in flight across the barrier.
// The last chunk issues the same (clamped, unused) operations so that the
// counts hold for every iteration; the first one starts from a full drain.
template <typename Cfg, bool SPARSE>
__device__ __forceinline__ void spmm_tiled_body_dpp(
    float (&acc)[Cfg::kRPW][Cfg::kVec], float* __restrict__ tile0, int lane, int wave, int slot0,
    int slots, int nchunks, int nonzeros, int n, int k, int n0, const float* __restrict__ values,
    const int* __restrict__ column_indices, const int* __restrict__ table,
    const float* __restrict__ dense, bool dbg_no_compute, bool dbg_no_stage,
    bool dbg_no_barrier = false) {
  constexpr int BN = Cfg::kBN, BK = Cfg::kBK, RPW = Cfg::kRPW;
  constexpr int S = Cfg::kStageOps;
  // windows in flight (the 512-column tile has no registers for more than four:
  // 64 accumulators + 32 for the B strips of a four-entry batch)
  constexpr int D = Cfg::kVec == 8 ? 4 : 8;
  static_assert(RPW % D == 0 && D <= RPW, "window ring");
  constexpr int kWaitSameChunk = 2 * (D - 1);
  constexpr int kWaitCrossChunk = 2 * (D - 1) + S;
  constexpr int kWaitStage = 2 * RPW;
  static_assert(kWaitStage <= 63, "vmcnt is a 6-bit counter");

  // SPARSE: the window holds 16 entries, already replicated in every 16-lane
  // row by the load itself (lane l reads entry l % 16): no ds_bpermute step, a
  // quarter of the window traffic, and the entries are processed exactly
  // (rotating groups).  For short segments only: the rare segment with more
  // than 16 entries fetches the rest on demand, draining the wave's loads.
  const int* __restrict__ my_table = table + slot0;  // wave-uniform
  const int e16x4 = (lane & 15) * 4;
  const int window_lane = SPARSE ? (lane & 15) : lane;  // the window entry this lane loads
  const int last_entry = nonzeros - 1;
  const float* lane_tile = tile0 + lane * 4;
  unsigned b_lane_off[Cfg::kPieces];
#pragma unroll
  for (int h = 0; h < Cfg::kPieces; ++h)
    b_lane_off[h] = static_cast<unsigned>(min(n0 + h * 256 + lane * 4, n - 4)) * 4u;

  // Stream positions of this wave's rows at the start of the current chunk and
  // at its end: SGPRs (scalar loads of RPW table entries; B in the comment above
  // is therefore not a vector-memory operation).
  using Pos = Positions<RPW>;
  typename Pos::type s_ps = Pos::load(my_table);
  typename Pos::type s_pe = Pos::load(my_table + slots);
  wait_positions(s_ps);
  wait_positions(s_pe);

  // Entry window of a row: 64 (short-segment variant: 16) consecutive stream entries starting at
  // `start` (lane = entry, or entry = lane % 16 in the short-segment variant).
  // The per-lane byte offset is formed on the VALU -- the loop is bound by
  // SCALAR instruction issue (about 45 scalar operations per (row, chunk) visit
  // at 1.75 ns each per SIMD against 96 ns per visit), so
  // nothing that a vector instruction can do is left to the scalar unit.  Lanes
  // past the end of the arrays re-read the last entry; the counts are exact, so
  // such entries are never used.
  int vcol[D];
  float vval[D];
  auto window_offset = [&](int start) {
    return static_cast<unsigned>(min(start + window_lane, last_entry)) * 4u;
  };
  auto request = [&](int slot, int start) {
    const unsigned off = window_offset(start);
    vcol[slot] = untracked_load_i32(column_indices, off);
    vval[slot] = untracked_load_f32(values, off);
  };
#pragma unroll
  for (int r = 0; r < D; ++r) request(r, s_ps[r]);
  stage_chunk<Cfg>(tile0, dense, n, k, 0, wave, b_lane_off);
  wait_vm<0>();
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    // A (the timing experiment without staging still issues the copies, from
    // chunk 0, so that the operation count the waits assume is unchanged)
    stage_chunk<Cfg>(tile0 + (buf ^ 1) * (BK * BN), dense, n, k,
                     dbg_no_stage ? 0 : min(c + 1, nchunks - 1) * BK, wave, b_lane_off);
    // B: positions at the end of chunk c+1, needed when that chunk begins (a
    // scalar load has the whole chunk to land)
    typename Pos::type s_pe_next =
        Pos::load(my_table + static_cast<int64_t>(min(c + 2, nchunks)) * slots);
    // Tile row of column j starts at (j - c*BK) * BN floats: fold the chunk's
    // first column into the lane's base address once per chunk.
    const char* __restrict__ lane_base = reinterpret_cast<const char*>(
        lane_tile + buf * (BK * BN) - static_cast<int64_t>(c) * (BK * BN));
    const int kc_off = c * (BK * BN * 4);  // byte offset that lane_base turns into tile row 0

#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      int& wcol = vcol[r % D];
      float& wval = vval[r % D];
      if (r < D) {
        wait_vm<kWaitCrossChunk>(wcol, wval);
      } else {
        wait_vm<kWaitSameChunk>(wcol, wval);
      }
      const int cnt = s_pe[r] - s_ps[r];
      if constexpr (SPARSE) {
        // Laid out for the fewest scalar compares and branches on the common
        // short segment (the loop is bound by its serial instruction stream): an
        // empty segment and one of 1..3 entries pass three tests (>= B, bit 1,
        // bit 0); the test for a segment longer than the window is only reached
        // by segments of a full batch or more.
        entry_pair e = make_entry(wcol * (BN * 4), wval);
        int tail = cnt;  // bits 0 and 1: the entries left after the batches of four
        constexpr int B = Cfg::kBatch;
        if (cnt >= B) {
          int left = min(16, cnt);
          do {
            dpp_group_at0<B>(acc[r], e, lane_base);
            e = row_rotate_entry<B>(e);
            left -= B;
          } while (left >= B);
          tail = left;
          if (cnt > 16) {  // rare: the rest of a long segment, 16 entries at a time (tail = 0 here)
            const int start = s_ps[r];
            for (int q0 = 16; q0 < cnt; q0 += 16) {
              const unsigned off = window_offset(start + q0);
              int c2 = untracked_load_i32(column_indices, off);
              float v2 = untracked_load_f32(values, off);
              wait_vm<0>();
              asm volatile("" : "+v"(c2), "+v"(v2));
              dpp_entries_exact(acc[r], min(16, cnt - q0), c2 * (BN * 4), v2, lane_base);
            }
          }
        }
        if constexpr (B == 8) {
          if (tail & 4) {
            dpp_group_at0<4>(acc[r], e, lane_base);
            e = row_rotate_entry<4>(e);
          }
        }
        if (tail & 2) {
          dpp_group_at0<2>(acc[r], e, lane_base);
          e = row_rotate_entry<2>(e);
        }
        if (tail & 1) dpp_group_at0<1>(acc[r], e, lane_base);
      } else {
      for (int q0 = 0; q0 < cnt; q0 += 16) {
        // replicate entries q0 .. q0+15 of the row into every 16-lane row
        const int idx = e16x4 + (q0 << 2);
        const int rcol = __builtin_amdgcn_ds_bpermute(idx, wcol);
        const float rval_all = __builtin_bit_cast(
            float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, wval)));
        // lanes standing for entries past the row's count: zero value, tile row 0
        const bool valid = e16x4 < ((cnt - q0) << 2);
        const int roff = valid ? rcol * (BN * 4) : kc_off;
        const float rval = valid ? rval_all : 0.f;
        dpp_entries(acc[r], min(16, cnt - q0), roff, rval, lane_base);
      }
      }
      // C: request the window that will be consumed D rows from now.
      request(r % D, (r + D < RPW) ? s_ps[(r + D) % RPW] : s_pe[(r + D) % RPW]);
    }
    wait_positions(s_pe_next);
    s_ps = s_pe;
    s_pe = dbg_no_compute ? s_pe : s_pe_next;  // timing experiment: every later segment is empty
    wait_vm<kWaitStage>();  // next B tile landed; windows and positions stay in flight
    if (!dbg_no_barrier) __syncthreads();
  }
  wait_vm<0>();  // nothing may be in flight (LDS-DMA!) when the wave ends
  // ... and the windows requested by the last rows (clamped, never used) must
  // stay ALLOCATED up to this wait: the compiler takes an asm-issued load as
  // complete where it is issued, so a register whose value is never read is free
  // to it at once -- it put the output address of the epilogue into such a
  // register pair ahead of the wait (one faster layout of the loop did; which one
  // does is a matter of scheduling), the late load landed on it, and the store
  // went to a wild address: an aperture violation in 40-75 % of the runs on a
  // warmed-up GPU, none on a cold one.
#pragma unroll
  for (int i = 0; i < D; ++i) {
    tie_reg(vcol[i]);
    tie_reg(vval[i]);
  }
}

// SPARSE = false: 64-entry windows, groups of four entries with a padded last
// group (long segments).  SPARSE = true: 16-entry windows replicated by the
// load itself, rotating groups with exact tails (short segments).
template <typename Cfg, bool SPARSE>
__global__ __launch_bounds__(Cfg::kThreads) void spmm_tiled_kernel(
    int m, int k, int n, int nonzeros, int slots, int nchunks, int n_tiles,
    const int* __restrict__ row_indices, const float* __restrict__ values,
    int64_t values_stride, const int* __restrict__ column_indices,
    const int* __restrict__ table, const float* __restrict__ dense, int64_t dense_stride,
    float* __restrict__ out, int64_t out_stride, const int* __restrict__ row_ok,
    const int* __restrict__ row_offsets, int debug, Epilogue epi) {
  // Timing experiments only (SPUTNIK_HIP_SPMM_DEBUG): 1 = no compute, 2 = no staging.
  const bool dbg_no_compute = debug & 1, dbg_no_stage = debug & 2;
  const bool dbg_no_barrier = debug & 4;  // wrong results: how much the per-chunk rendezvous costs

  constexpr int BN = Cfg::kBN, BK = Cfg::kBK, RPW = Cfg::kRPW, VEC = Cfg::kVec;
  static_assert(BK <= kWave, "a row has at most BK <= 64 entries per chunk (one window)");
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
  }
  int replica = blockIdx.y;
  if (n_tiles % 8 != 0) {
    // fewer (or an odd number of) column tiles: the row blocks that stage the
    // same B tile (same replica, same column tile) get consecutive work indices,
    // i.e. one XCD (xcd_local_index) -- with one 512-column tile and 8 replicas
    // every XCD would otherwise fetch every replica's B for itself
    const int work = xcd_local_index32();
    const int mblocks = gridDim.x / n_tiles;
    mblock = work % mblocks;
    ntile = (work / mblocks) % n_tiles;
    replica = work / (mblocks * n_tiles);
  }
  values += replica * values_stride;
  dense += replica * dense_stride;
  out += replica * out_stride;

  const int n0 = ntile * BN;
  const int slot0 = mblock * Cfg::kBM + wave * RPW;

  // Row blocks whose column indices do not ascend inside rows cannot be cut by
  // K chunk: they take the order-independent path (B gathered from L2).
  // (the 512-column tile has no LDS to spare for the workgroup-wide reduction)
  if (!(BN == 512 ? block_rows_ok_wave(row_ok, mblock * Cfg::kBM, Cfg::kBM)
                  : block_rows_ok(row_ok, mblock * Cfg::kBM, Cfg::kBM))) {
    for (int r = 0; r < RPW; ++r) {
      const int entry = dealt_index(slot0 + r, slots, kDealPer);
      if (entry >= m) continue;
      const int row = row_indices[entry];
#pragma unroll
      for (int h = 0; h < Cfg::kPieces; ++h) {
        const int col = n0 + h * 256 + lane * 4;
        if (col >= n) continue;   // last, partial column tile
        const float4 acc4 = gather_row_strip(values, column_indices, row_offsets[row],
                                             row_offsets[row + 1], dense + col, n);
        *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + col) =
            apply_epilogue(acc4, epi, row);
      }
    }
    return;
  }

  float acc[RPW][VEC];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[r][v] = 0.f;

  spmm_tiled_body_dpp<Cfg, SPARSE>(acc, &tile[0][0], lane, wave, slot0, slots, nchunks, nonzeros, n,
                                   k, n0, values, column_indices, table, dense, dbg_no_compute,
                                   dbg_no_stage, dbg_no_barrier);

#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int entry = dealt_index(slot0 + r, slots, kDealPer);
    if (entry < m) {
      const int row = row_indices[entry];
#pragma unroll
      for (int h = 0; h < Cfg::kPieces; ++h)
        if (n0 + h * 256 + lane * 4 < n)   // (the last column tile may be partial)
          *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + n0 + h * 256 + lane * 4) =
              apply_epilogue(make_float4(acc[r][4 * h], acc[r][4 * h + 1], acc[r][4 * h + 2],
                                         acc[r][4 * h + 3]),
                             epi, row);
    }
  }
}

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
  // whole runs of kDealPer slots (dealt_index), whatever the tile's row count
  constexpr int kUnit = Cfg::kBM > kDealPer ? Cfg::kBM : kDealPer;
  p.slots = ceil_div(m, kUnit) * kUnit;
  p.nchunks = ceil_div(k, Cfg::kBK);
  p.n_tiles = ceil_div(n, Cfg::kBN);
  p.table_bytes = sizeof(int) * static_cast<size_t>(p.nchunks + 1) * p.slots;
  p.use = true;
  return p;
}

using CfgLarge = TileConfig<256, 16, 16, 64>;  // 256 x 256 tile of C per workgroup
// Same kernel with 8 rows per wave (128 x 256 tiles): twice the workgroups, for
// problems whose 256-row blocks would leave most of the 256 CUs idle (e.g. one
// 2048^3 product is only 64 large tiles).  Same workspace layout.
using CfgMedium = TileConfig<256, 16, 8, 64>;
// 8 waves x 8 rows (64 x 256 tiles): four times the workgroups of the large
// tile for problems that would otherwise cover a fraction of the chip.
using CfgSmall = TileConfig<256, 8, 8, 64>;
// 128 x 512 tile of C: eight columns per lane, i.e. ONE entry broadcast and
// address per two ds_read_b128 and eight FMAs -- the inner step costs 2.0 ns per
// FMA instruction against 2.6 with four columns (tools/ubench.hip, step_pair_v8
// vs step_pair_v4).  The accumulators of 8 rows fill the registers that 16 rows
// take above, and 32 rows of B are all that two LDS stages hold: a chunk table
// of its own (BK = 32).  For problems large enough to give every CU a tile.
// (40-row chunks -- all 160 KiB of LDS, a fifth fewer visits -- measured the same at
// density 0.1 and 5 % slower at 0.5: longer segments cost LDS round trips.)
using CfgWide512 = TileConfig<512, 16, 8, 32>;
// The same with four rows per wave (64 x 512 tiles, same chunk table): twice the
// workgroups for grids that would leave CUs idle (config 5's left_spmm: 2048^2 x
// 512 x 8 replicas = 128 tiles of 128 rows, 256 of 64).
using CfgWide512Half = TileConfig<512, 16, 4, 32>;

// Developer / test knob SPUTNIK_HIP_SPMM_KERNEL (options.h: read once): "wide" =
// 256-column kernel whenever it applies, "wide512" = 512-column kernel whenever
// it applies (else as "wide"), "narrow" = 64-column kernel whenever it applies,
// "gather" = row-gather kernel, "flat" = as "wide512" with the flat-stream form
// of that tile (spmm_flat.hip) whenever it applies; anything else = the
// automatic choice.
// (The parity tests use it to reach every kernel with small inputs.)
inline int forced_kernel() { return options().spmm_kernel == 3 ? 0 : options().spmm_kernel; }

// Column tiles of BN columns serve any n that is a multiple of 4 (16-byte rows of
// B and C): the last tile may be partial (its LDS copies are clamped into the
// row, its stores predicated).  Taken when at most a quarter of the staged
// columns is padding: n = 1000 -> two 512-column tiles, 4000 -> eight,
// 200 -> one 256-column tile; n = 72 goes to the 64-column kernel.
inline bool fits_tiles(int n, int bn) {
  return n % 4 == 0 && static_cast<int64_t>(ceil_div(n, bn)) * bn * 3 <= static_cast<int64_t>(n) * 4;
}

inline bool tiled_applicable(int m, int k, int n, int nonzeros) {
  if (forced_kernel() > 0) return false;
  // Enough work per row block to amortise staging B (each workgroup stages
  // k x 256 floats): mean row length >= 16.
  return fits_tiles(n, CfgLarge::kBN) && k >= CfgLarge::kBK && m >= 64 &&
         nonzeros >= 16 * static_cast<int64_t>(m) && nonzeros < (1 << 30);  // 32-bit byte offsets
}

// Which tiled kernel serves a call.  Both may be applicable (n a multiple of
// 256): the 256-column kernel wins when even its 64-row tiles give about one
// workgroup per CU, the 64-column kernel when they would cover a fraction of the
// chip (measured at 2048 x 2048, density 0.2: n = 256 with 4 replicas 10.6 vs
// 15.2 TFLOP/s, with 16 replicas 28.5 vs 23.2).  The choice needs the replica
// count, which a plan made ahead of time does not know: then both tables are
// built (side by side in the workspace) and the call decides.
enum class Kernel { kNone, kWide, kNarrow, kEither, kWide512, kFlat };

inline bool tiled512_applicable(int m, int k, int n, int nonzeros) {
  return forced_kernel() <= 0 && fits_tiles(n, CfgWide512::kBN) && k >= CfgWide512::kBK && m >= 64 &&
         nonzeros >= 16 * static_cast<int64_t>(m) && nonzeros < (1 << 30);  // 32-bit byte offsets
}

inline int64_t tiles512(int m, int n) {
  return static_cast<int64_t>(ceil_div(m, CfgWide512::kBM)) * ceil_div(n, CfgWide512::kBN);
}
constexpr int64_t kTiles512From = 192;
// The flat-stream kernel (spmm_flat.hip) is taken when ONE replica gives about a
// workgroup per CU (its plan then does not depend on the replica count); the knob
// "flat" takes it for any shape it can serve, "wide512" / "wide" keep the
// visit-per-row kernels of this file.
// (round 5, tools/spmm_dispatch_sweep.py: at density 0.02 the stream's windows run a third
// full, and over SEVERAL rounds of the chip the 256-column visit-per-row kernel is 8-12 %
// ahead -- 4096^2 x 1024 x 64 replicas 3.20 against 3.59 ms, 2048^2 x 512 x 64: 422 against
// 469 us; in one or two rounds the flat kernel's larger tile still wins by a quarter --
// 4096^2 x 512 x 8: 145 against 190 us; at 0.05 it leads everywhere: the line is drawn at
// density 0.03 and 512 tiles)
// (and a short k: 8192 x 256 at density 0.2 against 2048 columns: 70 us against 54 for the
// 512-column kernel -- eight chunks do not pay for the stream's prologue; from 512 on the two
// are within 6 %)
inline bool flat_pays(int m, int k, int nonzeros, int64_t tiles) {
  return k >= 512 && (tiles < 512 || static_cast<double>(nonzeros) >= 0.03 * static_cast<double>(m) * k);
}
inline bool use_flat(int m, int k, int n, int nonzeros) {
  const int forced = forced_kernel();
  return (forced == -3 || (forced == 0 && spmm_flat_tiles(m, n) >= kTiles512From &&
                           flat_pays(m, k, nonzeros, spmm_flat_tiles(m, n)))) &&
         spmm_flat_applicable(m, k, n, nonzeros);
}
// The flat-stream kernel also serves shapes that need SEVERAL replicas to fill the chip
// (round 4: config 5 at its stated size, 2048^2 x 2048 x batch 8 = 64 tiles per replica).
// Its plan depends on the topology alone, so a workspace that has to serve any replica
// count carries it BEHIND the tables of the other kernels (as wide512_possible below),
// and the call decides.  (From 8 tiles per replica: fewer would need more than 24 replicas.)
inline bool flat_possible(int m, int k, int n, int nonzeros) {
  return forced_kernel() == 0 && !use_flat(m, k, n, nonzeros) && spmm_flat_tiles(m, n) >= 8 &&
         k >= 512 && spmm_flat_applicable(m, k, n, nonzeros);
}
inline bool flat_with_replicas(int m, int k, int n, int nonzeros, int replicas) {
  return replicas > 1 && flat_possible(m, k, n, nonzeros) &&
         spmm_flat_tiles(m, n) * replicas >= kTiles512From &&
         flat_pays(m, k, nonzeros, spmm_flat_tiles(m, n) * replicas);
}
// A row has more than about two entries per 32-row chunk (below that the 64-row
// chunks of the 256-column kernel win by 2 %: 4096^3 at density 0.05).
inline bool long_enough_for_512(int m, int k, int nonzeros) {
  return static_cast<int64_t>(nonzeros) * 5 >= int64_t{11} * m * ceil_div(k, CfgWide512::kBK);
}
// The 512-column kernel could serve this shape once enough replicas come in one
// call, but the shape alone does not select it: a workspace that has to serve
// any replica count carries its table BEHIND those of the other kernels.
inline bool wide512_possible(int m, int k, int n, int nonzeros) {
  return forced_kernel() == 0 && !use_flat(m, k, n, nonzeros) && tiled512_applicable(m, k, n, nonzeros) &&
         long_enough_for_512(m, k, nonzeros) && 2 * tiles512(m, n) < kTiles512From;
}

inline size_t wide512_workspace_bytes(int m, int k, int n, int nonzeros) {
  const Plan plan = make_plan<CfgWide512>(m, k, n);
  (void)nonzeros;
  return (row_ok_bytes(plan.slots) + plan.table_bytes + 15) / 16 * 16;
}

inline size_t wide_workspace_bytes(int m, int k, int n) {
  const Plan plan = make_plan<CfgLarge>(m, k, n);
  return (row_ok_bytes(plan.slots) + plan.table_bytes + 15) / 16 * 16;
}

inline Kernel choose_kernel(int m, int k, int n, int nonzeros, int replicas /* < 0: unknown */) {
  // Small calls are launch-latency bound: the workspace-free row-gather kernel is one launch,
  // the tiled kernels are a pre-pass plus a kernel with a staging pipeline to fill, and a lone
  // product gives them few workgroups.  The thresholds below are those of round 5's sweeps
  // (tools/spmm_dispatch_sweep.py: every kernel the knob can force against this choice on 342
  // shapes, profiles/r5_spmm_dispatch_sweep*; W = multiply-adds of the call), taken with the
  // row gather as it is since that round (spmm.hip: the next window and eight gathers of a
  // row in flight); rounds 1-4 drew them at 2^27 / 2^29 for a slower one.
  const int forced = forced_kernel();
  if (forced == 2) return Kernel::kNone;
  if (replicas >= 0 && forced == 0) {
    const int64_t work = static_cast<int64_t>(nonzeros) * n * replicas;
    // The row gather leads up to W = 2^28 for any batch (2048^2 x 64 x 8 replicas at density
    // 0.1: 42 against 60 us for the 64-column kernel) and to 2^29 for fewer than 8 replicas.
    // ONE product over a long K leaves it for the 64-column kernel's K split (round 4: K chunks
    // dealt to several workgroups per tile) where the rows are long enough for the gather's
    // serial walk, 0.09 us an entry, to show: from about 288 entries (1024^2 at density 0.3 x
    // 64 columns: 30 against 26 us; 512 x 4096 at 0.2, 819 entries: 69 against 35; at 205
    // entries the two tie or the gather leads: 2048^2 x 256 at density 0.1: 22 against 33).
    const bool long_rows = nonzeros >= 288 * static_cast<int64_t>(m);
    const bool small = work < (int64_t{1} << 28) || (replicas < 8 && work < (int64_t{1} << 29));
    if (small && replicas == 1 && long_rows && k >= 1024 && work >= (int64_t{1} << 22) &&
        spmm_tiled64_applicable(m, k, n, nonzeros) && spmm_tiled64_ksplits(m, k, n) >= 4 &&
        !use_flat(m, k, n, nonzeros))
      return Kernel::kNarrow;
    if (small) return Kernel::kNone;
  }
  if (use_flat(m, k, n, nonzeros) || flat_with_replicas(m, k, n, nonzeros, replicas))
    return Kernel::kFlat;
  // The 512-column kernel has one tile size: taken when the tiles of all replicas
  // give about one workgroup per CU.  (A plan made ahead of the call does not
  // know the replica count: when the shape alone does not decide, it builds this
  // kernel's table next to the others', see wide512_possible.)
  if (tiled512_applicable(m, k, n, nonzeros)) {
    // (with its 64-row variant the count that matters is that of 64-row tiles)
    const int64_t tiles = 2 * tiles512(m, n) * (replicas > 0 ? replicas : 1);
    if (forced == -2 || (forced == 0 && tiles >= kTiles512From && long_enough_for_512(m, k, nonzeros)))
      return Kernel::kWide512;
  }
  const bool wide = tiled_applicable(m, k, n, nonzeros);
  // (round 5, same sweep: the 64-column kernel visits every (row, 128-column chunk) -- with
  // fewer than about 4.5 entries per visit the row gather is ahead whatever the batch:
  // 4096^2 at density 0.02 x 64 columns x 8 replicas 42 against 85 us, x 64 replicas 274
  // against 371; at density 0.1 it is 143 against 121 the other way)
  const bool narrow = spmm_tiled64_applicable(m, k, n, nonzeros);
  // (and it needs workgroups -- one per 128 rows and 64 columns: with 128 of them or fewer
  // the row gather, which spreads rows over the whole chip, leads up to 2^30 multiply-adds:
  // 1024^2 at density 0.5 x 64 columns x 8 replicas, 64 workgroups: 63 against 92 us; 512 x
  // 4096 x 64 x 16 replicas at 0.2: 103 against 157; with 256 workgroups, 4096^2 x 64 x 8 at
  // density 0.1, it is 147 against 115 the other way)
  const bool sparse_visits = static_cast<double>(nonzeros) < 0.035 * static_cast<double>(m) * k;
  const bool few_workgroups =
      replicas >= 0 &&
      static_cast<int64_t>(ceil_div(m, 128)) * ceil_div(n, 64) * replicas *
              (replicas == 1 ? spmm_tiled64_ksplits(m, k, n) : 1) <= 128 &&   // (one product: its K split)
      static_cast<int64_t>(nonzeros) * n * replicas < (int64_t{1} << 30);
  const Kernel narrow_or_gather =
      forced == 0 && replicas >= 0 && (sparse_visits || few_workgroups) ? Kernel::kNone : Kernel::kNarrow;
  if (!wide) return narrow ? narrow_or_gather : Kernel::kNone;
  if (!narrow || forced < 0) return Kernel::kWide;
  const int64_t small_tiles = static_cast<int64_t>(ceil_div(m, CfgSmall::kBM)) * ceil_div(n, CfgSmall::kBN);
  // cross-over measured between 256 (narrow 8-10 % ahead) and 512 (wide 20 % ahead) tiles
  constexpr int64_t kWideFrom = 384;
  if (small_tiles >= kWideFrom) return Kernel::kWide;  // whatever the replica count
  if (replicas < 0) return Kernel::kEither;
  return small_tiles * replicas >= kWideFrom ? Kernel::kWide : narrow_or_gather;
}

}  // namespace

namespace {
// Bytes of the tables of the kernels that the shape alone admits (replica count unknown).
size_t base_workspace_bytes(int m, int k, int n, int nonzeros) {
  switch (choose_kernel(m, k, n, nonzeros, -1)) {
    case Kernel::kWide: return wide_workspace_bytes(m, k, n);
    case Kernel::kWide512: return wide512_workspace_bytes(m, k, n, nonzeros);
    case Kernel::kFlat: return spmm_flat_workspace_bytes(m, k, n, nonzeros);
    case Kernel::kNarrow: return spmm_tiled64_workspace_bytes(m, k, n);
    case Kernel::kEither: return wide_workspace_bytes(m, k, n) + spmm_tiled64_workspace_bytes(m, k, n);
    default: return 0;
  }
}
// Where the 512-column kernel's tables start: at the front when the shape alone
// selects it, behind the others' when only the replica count can.
size_t wide512_offset(int m, int k, int n, int nonzeros) {
  return wide512_possible(m, k, n, nonzeros)
             ? (base_workspace_bytes(m, k, n, nonzeros) + 255) / 256 * 256
             : 0;
}
// Everything but a flat plan that only the replica count selects.
size_t bytes_before_flat(int m, int k, int n, int nonzeros) {
  return wide512_possible(m, k, n, nonzeros)
             ? wide512_offset(m, k, n, nonzeros) + wide512_workspace_bytes(m, k, n, nonzeros)
             : base_workspace_bytes(m, k, n, nonzeros);
}
// Where the flat-stream plan starts: at the front when the shape alone selects the
// kernel, behind everything else when only the replica count can.
size_t flat_offset(int m, int k, int n, int nonzeros) {
  return flat_possible(m, k, n, nonzeros)
             ? (bytes_before_flat(m, k, n, nonzeros) + 255) / 256 * 256
             : 0;
}
}  // namespace

// 0 = row gather, 1 = 256-column, 2 = 64-column, 3 = either (by replica count), 4 = 512-column, 5 = flat stream
int spmm_tiled_choice(int m, int k, int n, int nonzeros, int replicas) {
  return static_cast<int>(choose_kernel(m, k, n, nonzeros, replicas));
}

const char* spmm_tiled_kernel_name(int m, int k, int n, int nonzeros, int replicas) {
  switch (choose_kernel(m, k, n, nonzeros, replicas)) {
    case Kernel::kFlat: return spmm_flat_kernel_name(m, k, nonzeros);
    case Kernel::kWide512: return "spmm_tiled_kernel<TileConfig<512";
    case Kernel::kWide: return "spmm_tiled_kernel<TileConfig<256";
    case Kernel::kNarrow: return "spmm_tiled64_kernel";
    case Kernel::kEither: return "spmm_tiled_kernel";
    default: return "spmm_rowgather_kernel";
  }
}

size_t spmm_tiled_workspace_bytes(int m, int k, int n, int nonzeros) {
  return flat_possible(m, k, n, nonzeros)
             ? flat_offset(m, k, n, nonzeros) + spmm_flat_workspace_bytes(m, k, n, nonzeros)
             : bytes_before_flat(m, k, n, nonzeros);
}

// Pre-pass only: topology -> per-row order status + chunk table in `workspace`.  Depends
// on the topology alone, so a caller with a static pattern can run it once and
// reuse the workspace for any number of spmm_tiled_exec calls (`replicas` < 0:
// not known yet).
int spmm_tiled_plan(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                    const int* row_offsets, const int* column_indices, void* workspace,
                    size_t workspace_bytes, hipStream_t stream, bool* planned) {
  *planned = false;
  const Kernel which = choose_kernel(m, k, n, nonzeros, replicas);
  // replica count unknown: also the 512-column kernel's table when a later call may take it
  const bool also512 = replicas < 0 && wide512_possible(m, k, n, nonzeros);
  const bool also_flat = replicas < 0 && flat_possible(m, k, n, nonzeros);
  if ((which == Kernel::kNone && !also512 && !also_flat) || workspace == nullptr ||
      !aligned_to(workspace, 16) ||
      workspace_bytes < spmm_tiled_workspace_bytes(m, k, n, nonzeros))
    return 0;
  // the narrow kernel's tables follow the wide kernel's whenever both could be needed
  const bool both_possible = choose_kernel(m, k, n, nonzeros, -1) == Kernel::kEither;
  if (which == Kernel::kNarrow || which == Kernel::kEither) {
    void* ws64 = both_possible ? static_cast<char*>(workspace) + wide_workspace_bytes(m, k, n)
                               : workspace;
    const int st =
        spmm_tiled64_plan(m, k, row_indices, row_offsets, column_indices, ws64, stream);
    if (st != 0) return st;
  }
  if (which == Kernel::kWide || which == Kernel::kEither) {
    const Plan plan = make_plan<CfgLarge>(m, k, n);
    int* row_ok = static_cast<int*>(workspace);
    int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + row_ok_bytes(plan.slots));
    hipLaunchKernelGGL((spmm_chunk_table_kernel<CfgLarge::kBK>), dim3(ceil_div(plan.slots, 4)),
                       dim3(256), 0, stream, m, k, plan.slots, kDealPer, plan.nchunks, row_indices,
                       row_offsets, column_indices, table, row_ok);
    const int st = launch_status();
    if (st != 0) return st;
  }
  if (which == Kernel::kFlat || also_flat) {
    const int st = spmm_flat_plan(m, k, n, nonzeros, row_indices, row_offsets, column_indices,
                                  static_cast<char*>(workspace) + flat_offset(m, k, n, nonzeros),
                                  stream);
    if (st != 0) return st;
  }
  if (which == Kernel::kWide512 || also512) {
    const Plan plan = make_plan<CfgWide512>(m, k, n);
    char* base = static_cast<char*>(workspace) + wide512_offset(m, k, n, nonzeros);
    int* row_ok = reinterpret_cast<int*>(base);
    int* table = reinterpret_cast<int*>(base + row_ok_bytes(plan.slots));
    hipLaunchKernelGGL((spmm_chunk_table_kernel<CfgWide512::kBK>), dim3(ceil_div(plan.slots, 4)),
                       dim3(256), 0, stream, m, k, plan.slots, kDealPer, plan.nchunks, row_indices,
                       row_offsets, column_indices, table, row_ok);
    const int st = launch_status();
    if (st != 0) return st;
  }
  *planned = true;
  return 0;
}

// Main kernel on a planned workspace.
int spmm_tiled_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                    const float* values, int64_t values_stride, const int* row_offsets,
                    const int* column_indices, const float* dense, int64_t dense_stride,
                    float* out, int64_t out_stride, const void* workspace,
                    size_t workspace_bytes, hipStream_t stream, Epilogue epi, bool* handled) {
  *handled = false;
  const Kernel which = choose_kernel(m, k, n, nonzeros, replicas);
  if (which == Kernel::kNone || workspace == nullptr || !aligned_to(workspace, 16) ||
      workspace_bytes < spmm_tiled_workspace_bytes(m, k, n, nonzeros) || replicas > kMaxGridYZ)
    return 0;
  // (the flat and the 64-column kernel's 16-byte accesses need dword alignment only:
  // any n, any stride)
  if (which != Kernel::kFlat && which != Kernel::kNarrow &&
      (!aligned_to(dense, 16) || !aligned_to(out, 16) || dense_stride % 4 != 0 ||
       out_stride % 4 != 0))
    return 0;
  if (which == Kernel::kNarrow) {
    const bool both_possible = choose_kernel(m, k, n, nonzeros, -1) == Kernel::kEither;
    const void* ws64 = both_possible
                           ? static_cast<const char*>(workspace) + wide_workspace_bytes(m, k, n)
                           : workspace;
    *handled = true;
    return spmm_tiled64_exec(m, k, n, nonzeros, replicas, row_indices, values, values_stride,
                             row_offsets, column_indices, dense, dense_stride, out, out_stride,
                             ws64, stream, epi);
  }
  using Cfg = CfgLarge;
  const bool w512 = which == Kernel::kWide512;
  if (which == Kernel::kFlat) {
    *handled = true;
    return spmm_flat_exec(m, k, n, nonzeros, replicas, row_indices, values, values_stride,
                          row_offsets, column_indices, dense, dense_stride, out, out_stride,
                          static_cast<const char*>(workspace) + flat_offset(m, k, n, nonzeros),
                          stream, epi);
  }
  const Plan plan = w512 ? make_plan<CfgWide512>(m, k, n) : make_plan<Cfg>(m, k, n);
  const char* ws_base =
      static_cast<const char*>(workspace) + (w512 ? wide512_offset(m, k, n, nonzeros) : 0);
  const int* row_ok = reinterpret_cast<const int*>(ws_base);
  const int* table = reinterpret_cast<const int*>(ws_base + row_ok_bytes(plan.slots));

  const int blocks = (plan.slots / Cfg::kBM) * plan.n_tiles;
  const int forced = options().spmm_sparse;  // developer knob: 0 / 1 forces the variant
  const int debug = options().spmm_debug;    // timing experiments only
  // Mean number of entries of a row inside one K chunk picks the variant
  // (measured cross-over with 64-row chunks, 4096^3 and 2048^3 x 8: density 0.2
  // short-segment form 7-9 % ahead, 0.25 a tie, 0.3 and above the long-segment form).
  // The 512-column tile always takes the short-segment variant: its 32-row chunks
  // hold at most 32 entries of a row, and it measured faster up to density 0.9
  // (55.3 vs 52.5 TFLOP/s), so the long-segment form is not even built for it.
  const bool sparse = w512 ? true
                      : forced >= 0
                          ? forced != 0
                          : static_cast<int64_t>(nonzeros) < int64_t{15} * m * plan.nchunks;
  const int force_tile = options().spmm_tile;  // developer knob: 1 = medium, 2 = small tile
  // Largest tile that still gives about one workgroup per CU (256 CUs).
  const int64_t large_blocks = static_cast<int64_t>(blocks) * replicas;
  const int tile = force_tile ? force_tile : large_blocks >= 192 ? 0 : 2 * large_blocks >= 192 ? 1 : 2;
#define SPUTNIK_HIP_LAUNCH_TILED(CFG, SPARSE_)                                                    \
  hipLaunchKernelGGL((spmm_tiled_kernel<CFG, SPARSE_>),                                           \
                     dim3((plan.slots / CFG::kBM) * plan.n_tiles, replicas), dim3(CFG::kThreads), \
                     0, stream, m, k, n, nonzeros, plan.slots, plan.nchunks, plan.n_tiles,        \
                     row_indices, values, values_stride, column_indices, table, dense,            \
                     dense_stride, out, out_stride, row_ok, row_offsets, debug, epi)
  if (w512) {
    // 128-row tiles when they give about one workgroup per CU, else 64-row tiles
    if (tiles512(m, n) * replicas >= kTiles512From && force_tile == 0) SPUTNIK_HIP_LAUNCH_TILED(CfgWide512, true);
    else SPUTNIK_HIP_LAUNCH_TILED(CfgWide512Half, true);
  } else if (tile == 0) {
    if (sparse) SPUTNIK_HIP_LAUNCH_TILED(CfgLarge, true);
    else SPUTNIK_HIP_LAUNCH_TILED(CfgLarge, false);
  } else if (tile == 1) {
    if (sparse) SPUTNIK_HIP_LAUNCH_TILED(CfgMedium, true);
    else SPUTNIK_HIP_LAUNCH_TILED(CfgMedium, false);
  } else {
    if (sparse) SPUTNIK_HIP_LAUNCH_TILED(CfgSmall, true);
    else SPUTNIK_HIP_LAUNCH_TILED(CfgSmall, false);
  }
#undef SPUTNIK_HIP_LAUNCH_TILED
  *handled = true;
  return launch_status();
}

}  // namespace sputnik_hip
