// Panel-resident SpMM for SMALL inner dimensions (k <= 512): C[m,n] = A_csr[m,k] * B[k,n].
//
// The LDS-tiled kernels walk K in chunks with a workgroup rendezvous and a B-tile
// copy per chunk; with k <= 512 that is four to sixteen chunks, and at the shapes
// this serves -- the 512 x 512 projection weights of the attention block against
// [512, 1024] x 8 activations (modules/sparse_attention.py:108-126) -- more than
// half of the launch was the skeleton of those few chunks (37 us for 0.43 GFLOP).
// Here a workgroup's whole B panel, k rows x 64 columns (<= 128 KiB), is copied
// to LDS ONCE (direct global->LDS copies, one rendezvous), and every row then
// runs its whole (column, value) stream against it: no chunk table, no pre-pass,
// no workspace, and no requirement that a row's columns ascend.
//
// Compute mapping as in spmm_tiled64.hip: a wavefront works on four rows at a
// time, each 16-lane group owning one row and its 64 output columns (4 per lane);
// the group holds 16 consecutive entries of its row (lane i = entry i) and hands
// entry u out with DPP row_newbcast: per nonzero one v_mov_b64_dpp, one
// v_add_u32, one ds_read_b128 and four FMAs.  The four groups run their own
// number of steps (EXEC-masked).  Rows are dealt to workgroups interleaved
// (dealt_index), neighbours in the caller's length-sorted row_indices share a
// quad.  The last column tile may be partial (n a multiple of 4).
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "options.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {

namespace {

using namespace tiled;

constexpr int kPBN = 64;      // columns of C per workgroup
constexpr int kPWaves = 16;   // waves per workgroup
constexpr int kPQuads = 4;    // row quads (4 rows) per wave
constexpr int kPBM = kPWaves * kPQuads * 4;   // 256 rows per workgroup
constexpr int kPThreads = kPWaves * kWave;
constexpr int kPMaxK = 512;   // rows of B per panel: 512 x 256 B = 128 KiB of LDS
constexpr int kPMaxPasses = 8;

// Element `idx` (>= 0) of an array behind a wave-uniform pointer: uniform base + 32-bit
// byte offset, i.e. no 64-bit vector arithmetic per address (the window loads of the
// stream loops: two per 16 entries and lane).
template <typename E>
__device__ __forceinline__ E at32(const E* __restrict__ base, int idx) {
  return *reinterpret_cast<const E*>(reinterpret_cast<const char*>(base) +
                                     static_cast<unsigned>(idx) * static_cast<unsigned>(sizeof(E)));
}

// Two independent row quads side by side: eight B strips in flight before the
// first FMA, so that one quad's LDS latency is covered by the other's arithmetic.
template <int G>
__device__ __forceinline__ void dpp_group4_pair(float (&acc_a)[4], float (&acc_b)[4], int roff_a,
                                                float rval_a, int roff_b, float rval_b,
                                                const char* __restrict__ lane_base) {
  const entry_pair ea = make_entry(roff_a, rval_a), eb = make_entry(roff_b, rval_b);
  const entry_pair a0 = row_bcast_entry<G + 0>(ea), a1 = row_bcast_entry<G + 1>(ea);
  const entry_pair a2 = row_bcast_entry<G + 2>(ea), a3 = row_bcast_entry<G + 3>(ea);
  const entry_pair b0 = row_bcast_entry<G + 0>(eb), b1 = row_bcast_entry<G + 1>(eb);
  const entry_pair b2 = row_bcast_entry<G + 2>(eb), b3 = row_bcast_entry<G + 3>(eb);
  BStrip<4> sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
  sa0.read(lane_base + entry_off(a0));
  sa1.read(lane_base + entry_off(a1));
  sa2.read(lane_base + entry_off(a2));
  sa3.read(lane_base + entry_off(a3));
  sb0.read(lane_base + entry_off(b0));
  sb1.read(lane_base + entry_off(b1));
  sb2.read(lane_base + entry_off(b2));
  sb3.read(lane_base + entry_off(b3));
  sa0.fma(acc_a, entry_val(a0));
  sa1.fma(acc_a, entry_val(a1));
  sa2.fma(acc_a, entry_val(a2));
  sa3.fma(acc_a, entry_val(a3));
  sb0.fma(acc_b, entry_val(b0));
  sb1.fma(acc_b, entry_val(b1));
  sb2.fma(acc_b, entry_val(b2));
  sb3.fma(acc_b, entry_val(b3));
}

// ---------------------------------------------------------------------------
// Building blocks shared by the kernels below.
// ---------------------------------------------------------------------------
// Which (row block, column tile, replica) a workgroup works on.  Workgroups are
// dealt to the 8 XCDs round-robin in launch order, and each XCD has its own L2:
// the row blocks that copy the SAME panel of B (same replica and column tile)
// are therefore given launch positions that are congruent modulo 8, so that one
// L2 fetches the panel from memory and serves the others (attention shapes: four
// row blocks per panel, which in plain order sit on four different XCDs).
struct Place {
  int mblock, ntile, replica;
};
__device__ __forceinline__ Place place_of_workgroup(int n_tiles, int mask_heads = 0) {
  // (many masks: the replicas of a mask dealt over the XCDs, see xcd_spread_replicas_index)
  const unsigned long long v =
      mask_heads > 0 ? xcd_spread_replicas_index(gridDim.x, gridDim.y * gridDim.z) : xcd_local_index();
  const unsigned mblocks = gridDim.x / n_tiles;
  Place p;
  p.mblock = static_cast<int>(v % mblocks);
  p.ntile = static_cast<int>((v / mblocks) % n_tiles);
  p.replica = static_cast<int>(v / (static_cast<unsigned long long>(mblocks) * n_tiles));
  return p;
}

// The 16 rows of a wave (4 row quads x 4 groups): row id (-1: padding), first
// stream position and length, as seen by the lanes of each 16-lane group.
struct Rows {
  int row[kPQuads], p0[kPQuads], cnt[kPQuads];
};

// IDENTITY: slot s is row s (a workgroup owns CONTIGUOUS rows: the transposing
// store and the accumulating group kernel need that); otherwise rows are dealt
// from the caller's length-sorted row_indices as in spmm_tiled.hip.
template <bool IDENTITY>
__device__ __forceinline__ void load_rows(Rows& r, int m, int slots, int mblock, int wave, int g,
                                          const int* __restrict__ row_indices,
                                          const int* __restrict__ row_offsets) {
#pragma unroll
  for (int t = 0; t < kPQuads; ++t) {
    const int slot = mblock * kPBM + wave * (kPQuads * 4) + 4 * t + g;
    const int entry = IDENTITY ? slot : dealt_index(slot, slots, kPBM);
    const bool live = entry < m;
    r.row[t] = IDENTITY ? (live ? entry : 0) : row_indices[live ? entry : 0];
    r.p0[t] = row_offsets[r.row[t]];
    r.cnt[t] = live ? row_offsets[r.row[t] + 1] - r.p0[t] : 0;
    if (!live) r.row[t] = -1;
  }
}

// Rows kbase .. kbase + 511 of B, columns col .. col + 3 per lane group -> panel.
// One wave instruction copies rows 4j .. 4j+3 (4 x 256 B; lane l -> row l / 16,
// bytes (l % 16) * 16).  Lanes past the end of a row of B (partial last column
// tile) or past the last row re-read valid bytes that are never used.  Returns
// with the wave's own copies landed; the caller joins the waves.
__device__ __forceinline__ void copy_panel_issue(float* panel, const float* __restrict__ dense,
                                                 int k, int n, int kbase, int col, int wave, int g,
                                                 bool skip = false /* timing experiment */) {
  const int rows_here = skip ? 0 : min(k - kbase, kPMaxK);
  for (int j = wave; j * 4 < rows_here; j += kPWaves) {
    const int src_row = min(kbase + 4 * j + g, k - 1);
    const unsigned off = static_cast<unsigned>(src_row) * static_cast<unsigned>(n) * 4u +
                         static_cast<unsigned>(col) * 4u;
    lds_dma_row(dense, off, panel + 4 * j * kPBN);
  }
}
__device__ __forceinline__ void copy_panel(float* panel, const float* __restrict__ dense, int k,
                                           int n, int kbase, int col, int wave, int g) {
  copy_panel_issue(panel, dense, k, n, kbase, col, wave, g);
  wait_vm<0>();
}

// First 16-entry window of every row quad, requested as soon as the rows' bounds
// are known: with ~50 entries per row a quad has only four windows, and a window
// fetched only when its quad starts costs a whole memory latency per quad pair.
template <bool PERM, typename TV = float>
__device__ __forceinline__ void fetch_first_windows(const Rows& r, int (&ecol)[kPQuads],
                                                    float (&eval)[kPQuads],
                                                    const int* __restrict__ column_indices,
                                                    const TV* __restrict__ values,
                                                    const int* __restrict__ value_permutation,
                                                    int last, int i) {
#pragma unroll
  for (int t = 0; t < kPQuads; ++t) {
    const int idx = max(min(r.p0[t] + i, last), 0);
    ecol[t] = at32(column_indices, idx);
    eval[t] = static_cast<float>(at32(values, PERM ? at32(value_permutation, idx) : idx));
  }
}

// The whole streams of a wave's rows against the resident panel (which holds all
// of B's rows: k <= 512), two row quads side by side.  Window w0 = entries w0 ..
// w0+15 of a group's row, the next one requested before the current one is
// worked on.  PERM: entry p takes values[value_permutation[p]].
// Round 4: the windows are requested kDepth + 1 = THREE ahead (a fourth set of registers spills).  With one window ahead a
// request had the four bundles of one window -- 0.26 us -- to land in, a third of a
// memory latency, and the waves of a workgroup, started together and joined by the panel's
// rendezvous, wait and compute in phase: rocprofv3 on the attention P.V (tools/
// spmm_c3_bench.py, SPUTNIK_HIP_SPMM_DEBUG=1): 25 us of the launch's 42 are there with the
// arithmetic switched off, and the arithmetic's 17 us come ON TOP.
// CHECK: every entry's column must lie in the panel, [kbase, kbase + kPMaxK) -- the rows
// were cut by a search that trusts their order; an entry outside is skipped and reported
// in `outside` (per lane; the caller votes and redoes the rows the slow way).
template <bool PERM, typename TV = float, bool CHECK = false>
__device__ __forceinline__ void stream_pairs(float (&acc)[kPQuads][4], const Rows& r,
                                             const int (&first_col)[kPQuads],
                                             const float (&first_val)[kPQuads],
                                             const int* __restrict__ column_indices,
                                             const TV* __restrict__ values,
                                             const int* __restrict__ value_permutation, int last,
                                             int i, const char* __restrict__ lane_base,
                                             int kbase = 0 /* first row of B in the panel */,
                                             int debug = 0 /* bit 0 (timing experiment): no arithmetic */,
                                             bool* outside = nullptr) {
  constexpr int kDepth = 1;
#pragma unroll
  for (int t = 0; t < kPQuads; t += 2) {
    const int n_a = r.cnt[t], n_b = r.cnt[t + 1];
    const int n_max = max(n_a, n_b);
    const int longest =
        max(max(__builtin_amdgcn_readlane(n_max, 0), __builtin_amdgcn_readlane(n_max, 16)),
            max(__builtin_amdgcn_readlane(n_max, 32), __builtin_amdgcn_readlane(n_max, 48)));
    // ring of the windows in flight: slot 0 = the window being worked on
    int col_a[kDepth + 1], col_b[kDepth + 1];
    float val_a[kDepth + 1], val_b[kDepth + 1];
    auto request = [&](int slot, int w0) {
      const int idx_a = max(min(r.p0[t] + w0 + i, last), 0);
      const int idx_b = max(min(r.p0[t + 1] + w0 + i, last), 0);
      col_a[slot] = at32(column_indices, idx_a);
      col_b[slot] = at32(column_indices, idx_b);
      val_a[slot] = static_cast<float>(at32(values, PERM ? at32(value_permutation, idx_a) : idx_a));
      val_b[slot] = static_cast<float>(at32(values, PERM ? at32(value_permutation, idx_b) : idx_b));
    };
    col_a[0] = first_col[t];
    col_b[0] = first_col[t + 1];
    val_a[0] = first_val[t];
    val_b[0] = first_val[t + 1];
#pragma unroll
    for (int d = 1; d <= kDepth; ++d) {
      col_a[d] = col_b[d] = 0;
      val_a[d] = val_b[d] = 0.f;
      if (16 * d < longest) request(d, 16 * d);
    }
    for (int w0 = 0; w0 < longest; w0 += 16) {
      const int left_a = n_a - w0, left_b = n_b - w0;
      bool in_a = i < left_a, in_b = i < left_b;
      if constexpr (CHECK) {
        const bool ok_a = static_cast<unsigned>(col_a[0] - kbase) < static_cast<unsigned>(kPMaxK);
        const bool ok_b = static_cast<unsigned>(col_b[0] - kbase) < static_cast<unsigned>(kPMaxK);
        *outside = *outside || (in_a && !ok_a) || (in_b && !ok_b);
        in_a = in_a && ok_a;
        in_b = in_b && ok_b;
      }
      const int roff_a = in_a ? (col_a[0] - kbase) * (kPBN * 4) : 0;
      const int roff_b = in_b ? (col_b[0] - kbase) * (kPBN * 4) : 0;
      const float rval_a = in_a ? val_a[0] : 0.f;
      const float rval_b = in_b ? val_b[0] : 0.f;
      // the ring moves on; the window kDepth + 1 ahead is requested before this one's work
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        col_a[d] = col_a[d + 1];
        col_b[d] = col_b[d + 1];
        val_a[d] = val_a[d + 1];
        val_b[d] = val_b[d + 1];
      }
      if (w0 + 16 * (kDepth + 1) < longest) request(kDepth, w0 + 16 * (kDepth + 1));
      const int left = (debug & 1) ? 0 : max(left_a, left_b);   // (entries past a row's end carry a zero value)
      if (left > 0) dpp_group4_pair<0>(acc[t], acc[t + 1], roff_a, rval_a, roff_b, rval_b, lane_base);
      if (left > 4) dpp_group4_pair<4>(acc[t], acc[t + 1], roff_a, rval_a, roff_b, rval_b, lane_base);
      if (left > 8) dpp_group4_pair<8>(acc[t], acc[t + 1], roff_a, rval_a, roff_b, rval_b, lane_base);
      if (left > 12) dpp_group4_pair<12>(acc[t], acc[t + 1], roff_a, rval_a, roff_b, rval_b, lane_base);
    }
  }
}

// One pass of the multi-panel form: the panel holds B's rows kbase .. kbase+511;
// every row's stream is walked, and only the groups of four entries that have a
// column inside the panel are worked on (bit u of in_panel = entry u).  The walk
// of quad t starts at window start[t] and reports, in start[t], the first window
// in which any of the wave's four rows held an entry of a LATER panel: the
// windows before it are done for good, whatever the order of the columns (rows
// with ascending columns thus walk each window about once over all passes).
template <bool PERM, typename TV = float>
__device__ __forceinline__ void stream_masked(float (&acc)[kPQuads][4], const Rows& r, int kbase,
                                              int (&start)[kPQuads],
                                              const int* __restrict__ column_indices,
                                              const TV* __restrict__ values,
                                              const int* __restrict__ value_permutation, int last,
                                              int g, int i, const char* __restrict__ lane_base) {
#pragma unroll
  for (int t = 0; t < kPQuads; ++t) {
    const int n_here = r.cnt[t];
    const int longest =
        max(max(__builtin_amdgcn_readlane(n_here, 0), __builtin_amdgcn_readlane(n_here, 16)),
            max(__builtin_amdgcn_readlane(n_here, 32), __builtin_amdgcn_readlane(n_here, 48)));
    const int begin = start[t];        // wave-uniform, a multiple of 16
    int first_later = longest;         // (rounded up below: "nothing left")
    bool seen_later = false;
    int idx = max(min(r.p0[t] + begin + i, last), 0);
    int ecol = at32(column_indices, idx);
    float eval = static_cast<float>(at32(values, PERM ? at32(value_permutation, idx) : idx));
    for (int w0 = begin; w0 < longest; w0 += 16) {
      const int cur_col = ecol - kbase;
      const float cur_val = eval;
      if (w0 + 16 < longest) {
        idx = min(r.p0[t] + w0 + 16 + i, last);
        ecol = at32(column_indices, idx);
        eval = static_cast<float>(at32(values, PERM ? at32(value_permutation, idx) : idx));
      }
      const int left = n_here - w0;   // entries of this group's row at or after the window start
      const bool in_row = i < left;
      const bool valid = in_row && static_cast<unsigned>(cur_col) < static_cast<unsigned>(kPMaxK);
      if (!seen_later && __builtin_amdgcn_ballot_w64(in_row && cur_col >= kPMaxK) != 0) {
        seen_later = true;             // (wave-uniform branch)
        first_later = w0;
      }
      const int roff = valid ? cur_col * (kPBN * 4) : 0;
      const float rval = valid ? cur_val : 0.f;
      const unsigned in_panel =
          static_cast<unsigned>(__builtin_amdgcn_ballot_w64(valid) >> (g * 16)) & 0xffffu;
      if (in_panel & 0x000fu) dpp_group4<0>(acc[t], roff, rval, lane_base);
      if (in_panel & 0x00f0u) dpp_group4<4>(acc[t], roff, rval, lane_base);
      if (in_panel & 0x0f00u) dpp_group4<8>(acc[t], roff, rval, lane_base);
      if (in_panel & 0xf000u) dpp_group4<12>(acc[t], roff, rval, lane_base);
    }
    start[t] = seen_later ? first_later : ((longest + 15) & ~15);
  }
}

// Two panels, rows with ascending columns (round 4).  stream_masked above serves any
// column order by walking every row's windows in every pass and masking: 11 window visits
// per row for 6.4 windows of entries at the attention shape, the straddling window worked
// on twice, 2.1 vector instructions per entry where the arithmetic step has 1.0
// (profiles/r3e_pmc_sq_attention_ops.json).  A row whose columns ascend is simply CUT at
// the first entry of the second panel: pass 0 runs entries [0, cut), pass 1 [cut, end),
// both as plain counted windows (stream_pairs: two row quads side by side, no mask).
// The cut is found by a 16-ary SEARCH for the panel boundary in the row's columns, while
// the first panel is being copied: two dependent requests per row (16 probes, then the 16
// entries between two probes) for rows of up to 256 entries, the four quads of a wave side
// by side.  (First form, measured: a scan of all of the row's windows that also checked
// the order -- 6 us of the launch's 42 at the attention shape.)  The search TRUSTS the
// order; what the passes need is only that every entry in front of the cut lies in the
// first panel and every entry behind it in the second, and that they check themselves
// (stream_pairs, CHECK): a workgroup in which any lane met an entry outside its panel
// throws its sums away and runs the masked walk.
__device__ __forceinline__ void search_cuts(const Rows& r, int (&cut)[kPQuads], int boundary,
                                            const int* __restrict__ column_indices, int last, int g,
                                            int i) {
  // the group's 16-bit slice of a wave ballot
  auto group_count = [&](bool flag) {
    return __popc(static_cast<unsigned>(__builtin_amdgcn_ballot_w64(flag) >> (g * 16)) & 0xffffu);
  };
  int lo[kPQuads], hi[kPQuads];   // the cut lies in [lo, hi]
#pragma unroll
  for (int t = 0; t < kPQuads; ++t) {
    lo[t] = 0;
    hi[t] = r.cnt[t];
  }
  // rounds: 16 probes spread over [lo, hi); rows of the wave differ in length, so every
  // quad runs the rounds of the longest bracket (wave-uniform trip count)
  for (;;) {
    int span = 0;
#pragma unroll
    for (int t = 0; t < kPQuads; ++t) span = max(span, hi[t] - lo[t]);
    const int widest =
        max(max(__builtin_amdgcn_readlane(span, 0), __builtin_amdgcn_readlane(span, 16)),
            max(__builtin_amdgcn_readlane(span, 32), __builtin_amdgcn_readlane(span, 48)));
    if (widest <= 0) break;
    int colv[kPQuads];
#pragma unroll
    for (int t = 0; t < kPQuads; ++t) {   // the four quads' requests together
      // probe i at lo + floor(i * len / 16); a bracket of at most 16: its entries themselves
      const int len = hi[t] - lo[t];
      const int probe = len <= 16 ? lo[t] + i : lo[t] + ((i * len) >> 4);
      colv[t] = at32(column_indices, max(min(r.p0[t] + probe, last), 0));
    }
#pragma unroll
    for (int t = 0; t < kPQuads; ++t) {
      const int len = hi[t] - lo[t];
      const bool small = len <= 16;
      // small: the entries in front of the boundary are counted and the cut is known.
      // else f probes lie in front of the boundary: the cut is behind probe f - 1 and not
      // behind probe f (the probes of a sorted row ascend; probe 0 is entry lo)
      const int f = group_count(colv[t] < boundary && (!small || i < len));
      const int new_lo = small ? lo[t] + f : (f > 0 ? lo[t] + (((f - 1) * len) >> 4) + 1 : lo[t]);
      const int new_hi = small ? new_lo : (f < 16 ? lo[t] + ((f * len) >> 4) : hi[t]);
      lo[t] = new_lo;
      hi[t] = max(new_hi, new_lo);
    }
  }
#pragma unroll
  for (int t = 0; t < kPQuads; ++t) cut[t] = min(max(lo[t], 0), r.cnt[t]);
}

__device__ __forceinline__ void store_rows(const float (&acc)[kPQuads][4], const Rows& r,
                                           float* __restrict__ out, int n, int n0, int i,
                                           const Epilogue& epi) {
#pragma unroll
  for (int t = 0; t < kPQuads; ++t)
    if (r.row[t] >= 0 && n0 + i * 4 < n)
      *reinterpret_cast<float4*>(out + static_cast<int64_t>(r.row[t]) * n + n0 + i * 4) =
          apply_epilogue(make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]), epi, r.row[t]);
}

// Transposed store in row blocks of `block_rows` (hd):
//   out[(row / hd) * n * hd + col * hd + row % hd] = C[row][col]
// i.e. every block of hd rows of C is written as its transpose [n][hd] -- the
// head split of modules/sparse_attention.py:38-45,108-126 (hd = head_dim), or
// the whole C^T (hd = m).  The workgroup's 256 x 64 tile goes through LDS in
// PHASES of kPhaseRows rows (row pitch 65 words: both the row-wise writes and the
// column-wise reads are conflict-free) and leaves as runs of consecutive rows,
// 256 contiguous bytes per column and 64 rows.  `tile` holds kPhaseRows x 65
// floats; rows are the workgroup's contiguous rows (load_rows<true>).  Contains
// workgroup barriers; the first one also separates the tile from whatever the
// LDS behind `tile` was used for before.
template <int kPhaseRows>
__device__ __forceinline__ void store_transposed(float* tile, const float (&acc)[kPQuads][4],
                                                 const Rows& r, float* __restrict__ out, int m,
                                                 int n, int n0, int mblock, int block_rows,
                                                 int wave, int g, int i, const Epilogue& epi) {
  constexpr int kPitch = kPBN + 1;
  constexpr int kWavesPerPhase = kPhaseRows / (kPQuads * 4);
  static_assert(kPBM % kPhaseRows == 0 && kPhaseRows % 64 == 0, "phases of whole 64-row runs");
#pragma unroll 1
  for (int phase = 0; phase < kPBM / kPhaseRows; ++phase) {
    __syncthreads();   // the tile is free (first phase: the panel is no longer read)
    if (wave / kWavesPerPhase == phase) {
#pragma unroll
      for (int t = 0; t < kPQuads; ++t) {
        const float4 v = apply_epilogue(make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]),
                                        epi, max(r.row[t], 0));
        float* dst = tile + ((wave % kWavesPerPhase) * (kPQuads * 4) + 4 * t + g) * kPitch + i * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      }
    }
    __syncthreads();
    const int row0 = mblock * kPBM + phase * kPhaseRows;
#pragma unroll
    for (int j = 0; j < (kPhaseRows / 4) * kPBN / kPThreads; ++j) {
      const int f = j * kPThreads + threadIdx.x;
      const int rq = (f % 16) + 16 * (f / (16 * kPBN));   // row quad of the phase's rows
      const int c = (f / 16) % kPBN;
      const int row = row0 + 4 * rq;
      if (row < m && n0 + c < n) {   // (m is a multiple of 4: a quad is inside or outside)
        const float* src = tile + (4 * rq) * kPitch + c;
        const float4 v = make_float4(src[0], src[kPitch], src[2 * kPitch], src[3 * kPitch]);
        const int64_t at = (static_cast<int64_t>(row / block_rows) * n + (n0 + c)) * block_rows +
                           row % block_rows;
        *reinterpret_cast<float4*>(out + at) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// One product per launch.
// PERM: entry p of the stream takes its value from values[value_permutation[p]]
// (a transposed topology over the values of the original one: the gather that
// a separate pass would do, 4-byte reads across a row of values that L2 holds,
// overlaps the arithmetic here).
// MULTI (k > 512): the panel is replaced every 512 rows of B; the C
// accumulators of a wave's 16 rows stay in registers, and every pass walks each
// row's whole stream again, working only on the groups of four entries that have
// a column inside the resident panel (a row with ascending columns pays each
// group once, plus the few that straddle a boundary; nothing has to be sorted).
// TOUT: transposed store (store_transposed), the whole tile in one phase through
// the LDS of the panel, which is no longer needed then.
// ---------------------------------------------------------------------------
template <bool MULTI, bool PERM, bool TOUT>
__global__ __launch_bounds__(kPThreads) void spmm_panel64_kernel(
    int m, int k, int n, int nonzeros, int slots, int n_tiles,
    const int* __restrict__ row_indices, const float* __restrict__ values,
    int64_t values_stride, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, const int* __restrict__ value_permutation,
    const float* __restrict__ dense, int64_t dense_stride, float* __restrict__ out,
    int64_t out_stride, Epilogue epi, int block_rows, int mask_heads, int first_replica,
    int debug /* bit 4: rows are never cut (the masked walk for every order); bit 5 (timing
                 experiment, wrong results): no panel copies */) {
  extern __shared__ float panel[];   // [min(k, 512)][64]

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const Place place = place_of_workgroup(n_tiles, mask_heads);
  const int ntile = place.ntile, mblock = place.mblock, replica = place.replica;
  values += replica * values_stride;
  dense += replica * dense_stride;
  out += replica * out_stride;
  const int n0 = ntile * kPBN;
  {
    const MaskPlace place_m = select_mask(mask_heads, first_replica + replica, m, nonzeros, row_offsets);
    row_offsets += static_cast<int64_t>(place_m.mask) * (m + 1);
    column_indices += place_m.first;
    if (row_indices != nullptr) row_indices += static_cast<int64_t>(place_m.mask) * m;
    nonzeros = place_m.nonzeros;
  }
  const int last = max(nonzeros - 1, 0);

  // the rows' bounds first: their latency overlaps the panel copy
  Rows rows;
  load_rows<TOUT>(rows, m, slots, mblock, wave, g, row_indices, row_offsets);

  float acc[kPQuads][4];
#pragma unroll
  for (int t = 0; t < kPQuads; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
  const int col = min(n0 + i * 4, n - 4);
  const char* __restrict__ lane_base = reinterpret_cast<const char*>(panel + i * 4);

  int first_col[kPQuads];
  float first_val[kPQuads];
  int start[kPQuads] = {};   // MULTI: first window a pass still has to look at
  if constexpr (!MULTI)   // (in flight while the panel is copied)
    fetch_first_windows<PERM>(rows, first_col, first_val, column_indices, values,
                              value_permutation, last, i);

  // MULTI with exactly two panels: rows cut at the panel boundary (search_cuts), found
  // while the first panel is on its way, the passes checking that the cut holds
  bool cut_rows = false, first_issued = false, outside = false;
  int cut[kPQuads] = {};
  if constexpr (MULTI) {
    if (k <= 2 * kPMaxK && !(debug & 16)) {
      // (the copies first: the search's loads wait behind them in the memory pipeline either way)
      copy_panel_issue(panel, dense, k, n, 0, col, wave, g, debug & 32);
      first_issued = true;
      cut_rows = true;
      if (!(debug & 256)) search_cuts(rows, cut, kPMaxK, column_indices, last, g, i);
    }
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    for (int kbase = 0; kbase < k; kbase += kPMaxK) {
      if (MULTI && (kbase > 0 || attempt > 0)) __syncthreads();   // every wave is done with the previous panel
      if (MULTI && cut_rows) {
        // this pass's part of every row, as a row of its own
        Rows part;
#pragma unroll
        for (int t = 0; t < kPQuads; ++t) {
          part.row[t] = rows.row[t];
          part.p0[t] = rows.p0[t] + (kbase > 0 ? cut[t] : 0);
          part.cnt[t] = kbase > 0 ? rows.cnt[t] - cut[t] : cut[t];
        }
        fetch_first_windows<PERM>(part, first_col, first_val, column_indices, values,
                                  value_permutation, last, i);   // (in flight while the panel is copied)
        if (kbase > 0) copy_panel_issue(panel, dense, k, n, kbase, col, wave, g, debug & 32);
        first_issued = false;
        wait_vm<0>();
        __syncthreads();
        if (!(debug & 128))   // (timing experiment: no stream at all)
          stream_pairs<PERM, float, true>(acc, part, first_col, first_val, column_indices, values,
                                          value_permutation, last, i, lane_base, kbase, debug, &outside);
        continue;
      }
      if (!(kbase == 0 && first_issued)) copy_panel_issue(panel, dense, k, n, kbase, col, wave, g, debug & 32);
      first_issued = false;
      wait_vm<0>();
      __syncthreads();
      if constexpr (!MULTI)
        stream_pairs<PERM>(acc, rows, first_col, first_val, column_indices, values,
                           value_permutation, last, i, lane_base);
      else
        stream_masked<PERM>(acc, rows, kbase, start, column_indices, values, value_permutation,
                            last, g, i, lane_base);
    }
    if constexpr (!MULTI) break;
    if (!cut_rows) break;
    // did every cut hold?  (one vote of the workgroup; an order the search cannot trust
    // is rare: the reference's topologies come from to_sparse_csr, columns ascending)
    if (!__syncthreads_or(outside ? 1 : 0)) break;
    cut_rows = false;
#pragma unroll
    for (int t = 0; t < kPQuads; ++t) {
      acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
      start[t] = 0;
    }
  }
  if constexpr (!TOUT)
    store_rows(acc, rows, out, n, n0, i, epi);
  else
    store_transposed<kPBM>(panel, acc, rows, out, m, n, n0, mblock, block_rows, wave, g, i, epi);
}

// ---------------------------------------------------------------------------
// Native half operands (round 3; the reference is float only, src/spmm_cuda.cu:42,51).
// TV / TB: storage type of the values / of the dense operand (float, _Float16,
// __bf16); the product is float.  A direct global->LDS copy cannot convert, so a
// half panel goes through registers: one 8-byte load per lane and row (its four
// columns), widened, one 16-byte LDS write -- half the bytes from HBM / L2, the
// panel in LDS and the whole compute loop as in the float kernel.  Values are
// widened as their 16-entry windows arrive.
// ---------------------------------------------------------------------------
template <typename TB>
__device__ __forceinline__ void copy_panel_widened(float* panel, const TB* __restrict__ dense, int k,
                                                   int n, int kbase, int col, int wave, int g,
                                                   int lane) {
  using raw4 = TB __attribute__((ext_vector_type(4)));
  using f4v = float __attribute__((ext_vector_type(4)));
  constexpr int kTrips = kPMaxK / 4 / kPWaves;   // rows 4j .. 4j+3 per wave instruction
  const int rows_here = min(k - kbase, kPMaxK);
  raw4 raw[kTrips];
#pragma unroll
  for (int u = 0; u < kTrips; ++u) {
    const int j = wave + u * kPWaves;
    const int src_row = min(kbase + 4 * j + g, k - 1);   // (past the end: valid bytes, never used)
    raw[u] = *reinterpret_cast<const raw4*>(dense + static_cast<int64_t>(src_row) * n + col);
  }
#pragma unroll
  for (int u = 0; u < kTrips; ++u) {
    const int j = wave + u * kPWaves;
    if (j * 4 < rows_here)
      *reinterpret_cast<f4v*>(panel + 4 * j * kPBN + lane * 4) = __builtin_convertvector(raw[u], f4v);
  }
}

template <bool MULTI, typename TV, typename TB>
__global__ __launch_bounds__(kPThreads) void spmm_panel64_typed_kernel(
    int m, int k, int n, int nonzeros, int slots, int n_tiles,
    const int* __restrict__ row_indices, const TV* __restrict__ values, int64_t values_stride,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const TB* __restrict__ dense, int64_t dense_stride, float* __restrict__ out,
    int64_t out_stride, Epilogue epi) {
  extern __shared__ float panel[];   // [min(k, 512)][64]
  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const Place place = place_of_workgroup(n_tiles);
  const int ntile = place.ntile, mblock = place.mblock, replica = place.replica;
  values += replica * values_stride;
  dense += replica * dense_stride;
  out += replica * out_stride;
  const int n0 = ntile * kPBN;
  const int last = max(nonzeros - 1, 0);

  Rows rows;
  load_rows<false>(rows, m, slots, mblock, wave, g, row_indices, row_offsets);
  float acc[kPQuads][4];
#pragma unroll
  for (int t = 0; t < kPQuads; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
  const int col = min(n0 + i * 4, n - 4);
  const char* __restrict__ lane_base = reinterpret_cast<const char*>(panel + i * 4);

  int first_col[kPQuads];
  float first_val[kPQuads];
  int start[kPQuads] = {};
  if constexpr (!MULTI)
    fetch_first_windows<false, TV>(rows, first_col, first_val, column_indices, values, nullptr,
                                   last, i);
  for (int kbase = 0; kbase < k; kbase += kPMaxK) {
    if (MULTI && kbase > 0) __syncthreads();   // every wave is done with the previous panel
    if constexpr (std::is_same_v<TB, float>)
      copy_panel(panel, dense, k, n, kbase, col, wave, g);
    else
      copy_panel_widened<TB>(panel, dense, k, n, kbase, col, wave, g, lane);
    __syncthreads();
    if constexpr (!MULTI)
      stream_pairs<false, TV>(acc, rows, first_col, first_val, column_indices, values, nullptr,
                              last, i, lane_base);
    else
      stream_masked<false, TV>(acc, rows, kbase, start, column_indices, values, nullptr, last, g,
                               i, lane_base);
  }
  store_rows(acc, rows, out, n, n0, i, epi);
}

// ---------------------------------------------------------------------------
// A GROUP of up to four products of one shape in one launch (k <= 512, values
// shared by the replicas): the projections of an attention block.
//   * same dense operand, separate outputs (ACCUM = false): the panel is copied
//     ONCE and the rows of every matrix run against it -- the q, k and v
//     projections of one input, each stored head split (TOUT);
//   * separate dense operands, ONE output that receives the sum (ACCUM = true):
//     the accumulators stay in registers while panel after panel is copied --
//     the input gradient  sum_w W_w^T dY_w  of those projections (PERM: the
//     transposed topologies over the weights' own value order), with no partial
//     results written and no additions afterwards.
// A workgroup owns contiguous rows whenever the stores are transposed or summed.
// The transposing store goes through a 64-row tile BEHIND the panel (the panel
// may still be needed by the next product): 128 + 16.25 KiB of LDS.
// ---------------------------------------------------------------------------
constexpr int kPMaxGroup = 4;
struct PanelProblem {
  const int* row_indices;
  const int* row_offsets;
  const int* column_indices;
  const float* values;
  const int* value_permutation;
  const float* dense;
  float* out;
  int nonzeros;
};
struct PanelGroup {
  PanelProblem p[kPMaxGroup];
  int count;
};
constexpr int kGroupPhaseRows = 64;
constexpr size_t kGroupTileBytes = kGroupPhaseRows * (kPBN + 1) * sizeof(float);

template <bool PERM, bool TOUT, bool ACCUM>
__global__ __launch_bounds__(kPThreads) void spmm_panel64_group_kernel(
    int m, int k, int n, int slots, int n_tiles, PanelGroup group, int64_t dense_stride,
    int64_t out_stride, int block_rows, int panel_floats) {
  extern __shared__ float panel[];   // [k][64], then the store's tile
  float* tile = panel + panel_floats;

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const Place place = place_of_workgroup(n_tiles);
  const int ntile = place.ntile, mblock = place.mblock, replica = place.replica;
  const int n0 = ntile * kPBN;
  const int col = min(n0 + i * 4, n - 4);
  const char* __restrict__ lane_base = reinterpret_cast<const char*>(panel + i * 4);
  const Epilogue none;

  float acc[kPQuads][4];
  const float* resident = nullptr;   // the dense operand the panel holds
#pragma unroll 1
  for (int pi = 0; pi < group.count; ++pi) {
    const PanelProblem& prob = group.p[pi];
    Rows rows;
    load_rows<(TOUT || ACCUM)>(rows, m, slots, mblock, wave, g, prob.row_indices, prob.row_offsets);
    int first_col[kPQuads];
    float first_val[kPQuads];
    fetch_first_windows<PERM>(rows, first_col, first_val, prob.column_indices, prob.values,
                              prob.value_permutation, prob.nonzeros - 1, i);
    const float* dense = prob.dense + replica * dense_stride;
    if (dense != resident) {   // (workgroup-uniform)
      if (pi > 0) __syncthreads();   // every wave is done with the previous panel
      copy_panel(panel, dense, k, n, 0, col, wave, g);
      __syncthreads();
      resident = dense;
    }
    if (pi == 0 || !ACCUM) {
#pragma unroll
      for (int t = 0; t < kPQuads; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
    }
    stream_pairs<PERM>(acc, rows, first_col, first_val, prob.column_indices, prob.values,
                       prob.value_permutation, prob.nonzeros - 1, i, lane_base);
    if (!ACCUM || pi == group.count - 1) {
      float* out = prob.out + replica * out_stride;
      if constexpr (TOUT)
        store_transposed<kGroupPhaseRows>(tile, acc, rows, out, m, n, n0, mblock, block_rows,
                                          wave, g, i, none);
      else
        store_rows(acc, rows, out, n, n0, i, none);
    }
  }
}

}  // namespace

// Shapes the panel kernel serves.  (Column indices must lie in [0, k): as for
// every kernel of the library, an out-of-range index is the caller's error.)
// LDS a workgroup of the current device may allocate (queried once per device): the
// panel kernels need 128 KiB, the group kernel 144.25 KiB; a device or partition
// that offers less takes the chunked kernels instead.
static size_t device_lds_bytes() {
  static std::atomic<int> cached[64];
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return 0;
  std::atomic<int>& slot = cached[device & 63];
  int v = slot.load(std::memory_order_acquire);
  if (v == 0) {
    int bytes = 0;
    if (hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess)
      bytes = 64 * 1024;
    v = bytes > 0 ? bytes : 64 * 1024;
    slot.store(v, std::memory_order_release);
  }
  return static_cast<size_t>(v);
}

bool spmm_panel_applicable(int m, int k, int n, int nonzeros, const float* dense,
                           int64_t dense_stride, const float* out, int64_t out_stride) {
  if (device_lds_bytes() < kPMaxK * kPBN * sizeof(float)) return false;
  return k >= 1 && k <= kPMaxK * kPMaxPasses && n % 4 == 0 && n >= kPBN && m >= 16 &&
         static_cast<int64_t>(k) * n * 4 < (int64_t{1} << 32) && aligned_to(dense, 16) &&
         aligned_to(out, 16) && dense_stride % 4 == 0 && out_stride % 4 == 0 && nonzeros >= 0 &&
         nonzeros < (1 << 29);   // (32-bit byte offsets into the entry arrays, at32)
}

int spmm_panel_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                      const float* values, int64_t values_stride, const int* row_offsets,
                      const int* column_indices, const float* dense, int64_t dense_stride,
                      float* out, int64_t out_stride, hipStream_t stream, Epilogue epi,
                      const int* value_permutation, int block_rows, int mask_heads) {
  const int slots = ceil_div(m, kPBM) * kPBM;
  const int n_tiles = ceil_div(n, kPBN);
  const int64_t blocks = static_cast<int64_t>(slots / kPBM) * n_tiles;
  if (blocks > 0x7fffffff) return SPUTNIK_HIP_INVALID_ARGUMENT;
  size_t lds = static_cast<size_t>(ceil_div(min(k, kPMaxK), 4) * 4) * kPBN * sizeof(float);
  if (block_rows > 0)   // the transposing store's tile (256 rows, pitch 65 words)
    lds = std::max(lds, static_cast<size_t>(kPBM) * (kPBN + 1) * sizeof(float));
  const bool multi = k > kPMaxK;
  // more than 64 KiB of dynamic LDS has to be asked for, once per device
  static std::atomic<uint64_t> asked{0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return launch_status();
  const uint64_t bit = uint64_t{1} << (device & 63);
  if (!(asked.load(std::memory_order_acquire) & bit)) {
    for (const void* f : {reinterpret_cast<const void*>(spmm_panel64_kernel<false, false, false>),
                          reinterpret_cast<const void*>(spmm_panel64_kernel<true, false, false>),
                          reinterpret_cast<const void*>(spmm_panel64_kernel<false, true, false>),
                          reinterpret_cast<const void*>(spmm_panel64_kernel<true, true, false>),
                          reinterpret_cast<const void*>(spmm_panel64_kernel<false, false, true>),
                          reinterpret_cast<const void*>(spmm_panel64_kernel<true, false, true>),
                          reinterpret_cast<const void*>(spmm_panel64_kernel<false, true, true>),
                          reinterpret_cast<const void*>(spmm_panel64_kernel<true, true, true>)}) {
      const hipError_t st = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                kPMaxK * kPBN * sizeof(float));
      if (st != hipSuccess) return static_cast<int>(st);
    }
    asked.fetch_or(bit, std::memory_order_release);
  }
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    const bool perm = value_permutation != nullptr, tout = block_rows > 0;
    const auto kernel =
        tout ? (perm ? (multi ? spmm_panel64_kernel<true, true, true> : spmm_panel64_kernel<false, true, true>)
                     : (multi ? spmm_panel64_kernel<true, false, true> : spmm_panel64_kernel<false, false, true>))
             : (perm ? (multi ? spmm_panel64_kernel<true, true, false> : spmm_panel64_kernel<false, true, false>)
                     : (multi ? spmm_panel64_kernel<true, false, false> : spmm_panel64_kernel<false, false, false>));
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks), ry), dim3(kPThreads), lds,
                       stream, m, k, n, nonzeros, slots, n_tiles, row_indices,
                       values + r0 * values_stride, values_stride, row_offsets, column_indices,
                       value_permutation, dense + r0 * dense_stride, dense_stride,
                       out + r0 * out_stride, out_stride, epi, block_rows, mask_heads, r0,
                       options().spmm_debug);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

// Typed form (any operand float16 / bfloat16): plain product, no permutation, no
// transposed store.  element sizes: alignment of the dense operand's 4-column pieces.
bool spmm_panel_applicable_typed(int m, int k, int n, int nonzeros, const void* dense,
                                 int dense_type, int64_t dense_stride, const float* out,
                                 int64_t out_stride) {
  if (device_lds_bytes() < kPMaxK * kPBN * sizeof(float)) return false;
  const size_t piece = dense_type == SPUTNIK_HIP_F32 ? 16 : 8;
  return k >= 1 && k <= kPMaxK * kPMaxPasses && n % 4 == 0 && n >= kPBN && m >= 16 &&
         static_cast<int64_t>(k) * n * 4 < (int64_t{1} << 32) && aligned_to(dense, piece) &&
         aligned_to(out, 16) && dense_stride % 4 == 0 && out_stride % 4 == 0 && nonzeros >= 0 &&
         nonzeros < (1 << 29);   // (32-bit byte offsets into the entry arrays, at32)
}

template <typename TV, typename TB>
static int panel_launch_typed(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_indices, const void* values_v, int64_t values_stride,
                              const int* row_offsets, const int* column_indices,
                              const void* dense_v, int64_t dense_stride, float* out,
                              int64_t out_stride, hipStream_t stream, Epilogue epi) {
  const TV* values = static_cast<const TV*>(values_v);
  const TB* dense = static_cast<const TB*>(dense_v);
  const int slots = ceil_div(m, kPBM) * kPBM;
  const int n_tiles = ceil_div(n, kPBN);
  const int64_t blocks = static_cast<int64_t>(slots / kPBM) * n_tiles;
  if (blocks > 0x7fffffff) return SPUTNIK_HIP_INVALID_ARGUMENT;
  const size_t lds = static_cast<size_t>(ceil_div(min(k, kPMaxK), 4) * 4) * kPBN * sizeof(float);
  const bool multi = k > kPMaxK;
  static std::atomic<uint64_t> asked{0};   // (one per instantiation)
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return launch_status();
  const uint64_t bit = uint64_t{1} << (device & 63);
  if (!(asked.load(std::memory_order_acquire) & bit)) {
    for (const void* f : {reinterpret_cast<const void*>(spmm_panel64_typed_kernel<false, TV, TB>),
                          reinterpret_cast<const void*>(spmm_panel64_typed_kernel<true, TV, TB>)}) {
      const hipError_t st = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                kPMaxK * kPBN * sizeof(float));
      if (st != hipSuccess) return static_cast<int>(st);
    }
    asked.fetch_or(bit, std::memory_order_release);
  }
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    const auto kernel = multi ? spmm_panel64_typed_kernel<true, TV, TB>
                              : spmm_panel64_typed_kernel<false, TV, TB>;
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks), ry), dim3(kPThreads), lds,
                       stream, m, k, n, nonzeros, slots, n_tiles, row_indices,
                       values + r0 * values_stride, values_stride, row_offsets, column_indices,
                       dense + r0 * dense_stride, dense_stride, out + r0 * out_stride, out_stride,
                       epi);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

int spmm_panel_launch_typed(int m, int k, int n, int nonzeros, int replicas,
                            const int* row_indices, const void* values, int values_type,
                            int64_t values_stride, const int* row_offsets,
                            const int* column_indices, const void* dense, int dense_type,
                            int64_t dense_stride, float* out, int64_t out_stride,
                            hipStream_t stream, Epilogue epi) {
#define SPUTNIK_HIP_PT(TV, TB)                                                                    \
  return panel_launch_typed<TV, TB>(m, k, n, nonzeros, replicas, row_indices, values,             \
                                    values_stride, row_offsets, column_indices, dense,            \
                                    dense_stride, out, out_stride, stream, epi)
  const int F = SPUTNIK_HIP_F32, H = SPUTNIK_HIP_F16, B = SPUTNIK_HIP_BF16;
  if (values_type == H && dense_type == H) SPUTNIK_HIP_PT(_Float16, _Float16);
  if (values_type == B && dense_type == B) SPUTNIK_HIP_PT(__bf16, __bf16);
  if (values_type == H && dense_type == F) SPUTNIK_HIP_PT(_Float16, float);
  if (values_type == B && dense_type == F) SPUTNIK_HIP_PT(__bf16, float);
  if (values_type == F && dense_type == H) SPUTNIK_HIP_PT(float, _Float16);
  if (values_type == F && dense_type == B) SPUTNIK_HIP_PT(float, __bf16);
#undef SPUTNIK_HIP_PT
  return SPUTNIK_HIP_UNSUPPORTED;   // (float / float: the untyped launch; half types of two kinds)
}

// Host side of the group kernel.  `problems`: host array of `count` entries.
// Returns SPUTNIK_HIP_UNSUPPORTED for shapes / combinations the kernel does not
// serve (the caller then runs the products one by one).
struct GroupProblemHost {   // = sputnik_hip_spmm_problem (include/sputnik_hip.h)
  const int* row_indices;
  const int* row_offsets;
  const int* column_indices;
  const float* values;
  const int* value_permutation;
  const float* dense;
  float* out;
  int nonzeros;
};

bool spmm_panel_group_supported(int m, int k, int n, int count, int block_rows, bool accumulate) {
  if (device_lds_bytes() < kPMaxK * kPBN * sizeof(float) + kGroupTileBytes) return false;
  return count >= 1 && count <= kPMaxGroup && k >= 1 && k <= kPMaxK && n % 4 == 0 && n >= kPBN &&
         m >= 16 && static_cast<int64_t>(k) * n * 4 < (int64_t{1} << 32) &&
         !(accumulate && block_rows > 0);
}

int spmm_panel_group_launch(int m, int k, int n, int replicas, int count,
                            const GroupProblemHost* problems, int64_t dense_stride,
                            int64_t out_stride, int block_rows, bool accumulate,
                            hipStream_t stream) {
  if (!spmm_panel_group_supported(m, k, n, count, block_rows, accumulate) ||
      dense_stride % 4 != 0 || out_stride % 4 != 0)
    return SPUTNIK_HIP_UNSUPPORTED;
  PanelGroup group{};
  group.count = count;
  bool all_perm = true, any_perm = false;
  for (int p = 0; p < count; ++p) {
    const GroupProblemHost& h = problems[p];
    if (h.nonzeros <= 0 || h.nonzeros >= (1 << 29) || !aligned_to(h.dense, 16) || !aligned_to(h.out, 16))
      return SPUTNIK_HIP_UNSUPPORTED;
    if (accumulate && h.out != problems[0].out) return SPUTNIK_HIP_INVALID_ARGUMENT;
    all_perm = all_perm && h.value_permutation != nullptr;
    any_perm = any_perm || h.value_permutation != nullptr;
    group.p[p] = PanelProblem{h.row_indices, h.row_offsets, h.column_indices, h.values,
                              h.value_permutation, h.dense, h.out, h.nonzeros};
  }
  if (any_perm && !all_perm) return SPUTNIK_HIP_UNSUPPORTED;
  const bool tout = block_rows > 0;
  using Kernel = void (*)(int, int, int, int, int, PanelGroup, int64_t, int64_t, int, int);
  Kernel kernel = nullptr;
  if (!any_perm && tout && !accumulate) kernel = spmm_panel64_group_kernel<false, true, false>;
  else if (any_perm && !tout && accumulate) kernel = spmm_panel64_group_kernel<true, false, true>;
  else if (!any_perm && !tout && accumulate) kernel = spmm_panel64_group_kernel<false, false, true>;
  else if (!any_perm && !tout && !accumulate) kernel = spmm_panel64_group_kernel<false, false, false>;
  else return SPUTNIK_HIP_UNSUPPORTED;
  if (!tout && !accumulate)
    for (int p = 0; p < count; ++p)
      if (problems[p].row_indices == nullptr) return SPUTNIK_HIP_INVALID_ARGUMENT;

  const int slots = ceil_div(m, kPBM) * kPBM;
  const int n_tiles = ceil_div(n, kPBN);
  const int64_t blocks = static_cast<int64_t>(slots / kPBM) * n_tiles;
  if (blocks > 0x7fffffff) return SPUTNIK_HIP_INVALID_ARGUMENT;
  const int panel_floats = ceil_div(k, 4) * 4 * kPBN;
  const size_t lds = panel_floats * sizeof(float) + (tout ? kGroupTileBytes : 0);
  static std::atomic<uint64_t> asked{0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return launch_status();
  const uint64_t bit = uint64_t{1} << (device & 63);
  if (!(asked.load(std::memory_order_acquire) & bit)) {
    for (Kernel f : {static_cast<Kernel>(spmm_panel64_group_kernel<false, true, false>),
                     static_cast<Kernel>(spmm_panel64_group_kernel<true, false, true>),
                     static_cast<Kernel>(spmm_panel64_group_kernel<false, false, true>),
                     static_cast<Kernel>(spmm_panel64_group_kernel<false, false, false>)}) {
      const hipError_t st =
          hipFuncSetAttribute(reinterpret_cast<const void*>(f),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              kPMaxK * kPBN * sizeof(float) + kGroupTileBytes);
      if (st != hipSuccess) return static_cast<int>(st);
    }
    asked.fetch_or(bit, std::memory_order_release);
  }
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    PanelGroup shifted = group;
    for (int p = 0; p < count; ++p) {
      shifted.p[p].dense += r0 * dense_stride;
      shifted.p[p].out += r0 * out_stride;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks), ry), dim3(kPThreads), lds,
                       stream, m, k, n, slots, n_tiles, shifted, dense_stride, out_stride,
                       block_rows, panel_floats);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

}  // namespace sputnik_hip
