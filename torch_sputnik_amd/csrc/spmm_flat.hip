// Flat-stream SpMM for gfx950: the 128 x 512 tile of spmm_tiled.hip with a
// different decomposition of the sparse operand.
//
// spmm_tiled.hip walks the tile (row, K chunk) by (row, K chunk): a visit at
// density 0.1 holds 3.2 nonzeros and costs 17 scalar instructions, two window
// requests and one or two dependent LDS round trips of bookkeeping (DESIGN.md
// section 3.1: 0.25 of the vector roof for two rounds).  The row structure is
// there because the accumulators of a row are fixed registers, so the code that
// multiplies a nonzero depends on its row.
//
// Here the accumulator row of a nonzero is selected at run time with the VGPR
// index mode (s_set_gpr_idx_on: destination and src2 of the four v_pk_fma_f32
// become M0-relative), so ONE instruction stream serves every row and a wave
// consumes a FLAT stream: the entries of all of its 8 rows, K chunk after K
// chunk, sorted by (column, row).  Two loops read it (gen_spmm_flat.py):
//   * "entry": every entry reads its B strip from LDS three entries ahead of
//     its FMAs, in a software pipeline without a taken branch -- the loop of
//     spmm_tiled.hip minus everything per (row, chunk); bound by the LDS (one B
//     dword per FMA);
//   * "group": entries of the same column of B follow each other in the stream
//     (a "column group") and the column's strip is read once per group, three
//     groups ahead -- 0.57 reads per entry at density 0.1, 0.25 at 0.5 -- which
//     takes the loop off the LDS roof at the price of branches at group starts.
// The dispatcher picks the loop by density (flat_mode below).  Per K chunk a wave
// pays one boundary (rendezvous + the copies of the next B tile), whatever the
// number of rows.
//
// The stream is made by the pre-pass from the topology alone (the plan is valid
// for any values, any number of replicas and both loops): per group of 8 row
// slots (one wave's) blocks ("windows") of 16 entries -- per entry the byte
// offset of its value in `values` (gathered by the kernel one window ahead), the
// tile row of its column, the accumulator index of its row, a flag on the first
// entry of a column group and, there, the tile row of the group three ahead.
// Per output element the summation order is the ascending-column order, i.e. the
// CSR order, as in every other kernel of this library (bitwise the same results).
//
// The main loop is generated assembly (gen_spmm_flat.py -> spmm_flat_body.inc);
// this file holds the pre-pass, the prologue / epilogue around the loop and the
// host side.  Replaces sputnik::CudaSpmm at /root/reference/src/spmm_cuda.cu:49-56
// for the shapes spmm_flat_applicable() names.
#include <algorithm>

#include "options.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {

namespace {

using namespace tiled;

constexpr int kBN = 512;       // columns of C per workgroup
constexpr int kBK = 32;        // rows of B per chunk (two 64 KiB stages)
constexpr int kWaves = 16;     // waves per workgroup
constexpr int kRPW = 8;        // rows per wave (the generated loop's accumulator map)
constexpr int kBM = kWaves * kRPW;
constexpr int kAhead = 3;      // column groups read ahead (strip ring of 4)
constexpr int kDealPer = 256;  // as spmm_tiled.hip: row slots are dealt in runs of 256
constexpr int kWindow = 16;    // entries per window
constexpr int kWindowBytes = 144;  // 16 x (tile rows, value offset) + 16 row bytes
constexpr int kTailWindows = 4;    // empty windows behind the last group (prefetch runs ahead)
constexpr int kMaxColumns = 65536; // the pre-pass keeps a row mask (one byte) per column in LDS
constexpr unsigned kNewGroup = 0x80;  // row byte: first entry of a column group

// (A 256 x 256 shape of the same loops -- 16 rows per wave, 64-row chunks, half
// the B copies per FMA, a quarter of the boundaries -- was built and measured:
// 0.42 ms against 0.33 at density 0.1.  Per entry it has half the FMAs for the
// same scalar instructions and branches.)

// Per group of row slots and K chunk: the number of column groups and of entries
// of the chunk, and the tile rows (column modulo 2 BK, a byte each) of the first
// three groups, whose LDS reads the group loop issues right behind the rendezvous.
struct ChunkInfo {
  int groups;
  unsigned first_rows;
  unsigned unused;
  int entries;
};

// ---------------------------------------------------------------------------
// Pre-pass: one workgroup per group of 8 row slots, one wave per row.
//   row_ok[slot]  the row's columns ascend and are in range (else its workgroup
//                 takes the order-independent path, as in spmm_tiled.hip)
//   cinfo[g][c]   ChunkInfo
//   gwin[g]       first window of the group's stream
//   stream        the windows
// The (column, row) order comes from one byte per column in LDS: bit r = row r
// of the group holds the column.  The position of an entry is the number of set
// bits in front of it, the first set bit of a byte starts a column group.
// ---------------------------------------------------------------------------
// masks (a byte per column) + per chunk: entries before, held columns, word prefixes;
// scratch; rounded to 16 bytes (the staged stream behind it moves in 16-byte pieces)
__host__ __device__ inline size_t fill_tables_bytes(int nchunks) {
  return (sizeof(int) * (static_cast<size_t>(nchunks) * (kBK / 4 + 4) + 2 * kRPW + 2) + 15) / 16 * 16;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

__global__ __launch_bounds__(kRPW * 64) void spmm_flat_fill_kernel(
    int m, int k, int slots, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ row_ok, ChunkInfo* __restrict__ cinfo, int* __restrict__ gwin,
    unsigned char* __restrict__ stream, int stage_windows, int debug) {
  extern __shared__ unsigned lds[];
  constexpr int RPW = kRPW, BK = kBK, NT = RPW * 64;
  constexpr int WPC = BK / 4;             // mask words per chunk (a byte per column)
  constexpr int UN = 4;                   // row entries a lane handles per round (loads in flight)
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid % 64, wave = tid / 64;
  const int groups = gridDim.x;
  unsigned* maskw = lds;                                        // [nchunks * WPC]
  int* ebase = reinterpret_cast<int*>(lds + nchunks * WPC);     // [nchunks]: entries before the chunk
  unsigned* nzmask = reinterpret_cast<unsigned*>(ebase + nchunks);   // [nchunks]: bit j = column j is held
  unsigned* wprefix = nzmask + nchunks;                         // [nchunks * 2]: entries before word i, a byte each
  int* scratch = reinterpret_cast<int*>(wprefix + 2 * nchunks); // [2 * RPW + 2]
  // [stage_windows * kWindowBytes], 16-byte aligned: the group's stream on its way out
  unsigned char* sbuf = reinterpret_cast<unsigned char*>(lds) + fill_tables_bytes(nchunks);

  for (int i = tid; i < nchunks * WPC; i += NT) maskw[i] = 0;
  // (1) this wave's row; windows of all groups before this one
  const int slot = g * RPW + wave;
  const int entry = dealt_index(slot, slots, kDealPer);
  int p0 = 0, p1 = 0;
  if (entry < m) {
    const int row = row_indices[entry];
    p0 = row_offsets[row];
    p1 = row_offsets[row + 1];
  }
  // (Round 4 measured the alternative -- the groups' window counts from a small kernel of
  // their own, summed here: that kernel alone takes 4.7 us, this loop 2.5.)
  int before = 0;
  for (int gp = tid; gp < g; gp += NT) {
    // two rounds of independent loads: row ids, then each row's two bounds as ONE 8-byte
    // load (4-byte aligned: all a global load needs)
    int rows[RPW], len = 0;
    int2 bounds[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int e = dealt_index(gp * RPW + r, slots, kDealPer);
      rows[r] = e < m ? row_indices[e] : -1;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      // (an 8-byte load of a 4-byte aligned pair: the type says so -- ADVICE r4)
      typedef int int2_a4 __attribute__((ext_vector_type(2), aligned(4)));
      if (rows[r] >= 0) {
        const int2_a4 pair = *reinterpret_cast<const int2_a4*>(row_offsets + rows[r]);
        bounds[r] = make_int2(pair.x, pair.y);
      } else {
        bounds[r] = make_int2(0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) len += bounds[r].y - bounds[r].x;
    before += (len + kWindow - 1) / kWindow;
  }
  before = wave_sum(before);
  if (lane == 0) {
    scratch[wave] = before;
    scratch[RPW + wave] = p1 - p0;
  }
  if (tid == 0) scratch[2 * RPW] = 1;
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    wbase += scratch[r];
    total += scratch[RPW + r];
  }
  if (tid == 0) gwin[g] = wbase;
  unsigned char* my_stream = stream + static_cast<int64_t>(wbase) * kWindowBytes;
  const int windows = (total + kWindow - 1) / kWindow;

  // (2) the rows: order check, column masks (UN independent loads per lane and round)
  bool ok = true;
  for (int base = p0; base < p1; base += 64 * UN) {
    int cur[UN], prev[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int p = base + u * 64 + lane;
      cur[u] = p < p1 ? column_indices[p] : -2;
    }
    // the entry before: the neighbouring lane's (lane 0: the last lane of the round before)
    const int before_base = base > p0 ? column_indices[base - 1] : -1;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int up = __shfl_up(cur[u], 1);
      const int carry = u == 0 ? before_base : __shfl(cur[u == 0 ? 0 : u - 1], 63);
      prev[u] = lane == 0 ? carry : up;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (cur[u] == -2) continue;
      if (cur[u] <= prev[u] || cur[u] >= k) ok = false;
      else atomicOr(&maskw[cur[u] >> 2], (1u << wave) << (8 * (cur[u] & 3)));
    }
  }
  const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
  if (lane == 0) {
    row_ok[slot] = wave_ok ? 1 : 0;
    if (!wave_ok) scratch[2 * RPW] = 0;
  }
  __syncthreads();
  ChunkInfo* my_info = cinfo + static_cast<int64_t>(g) * (nchunks + 1);
  if (scratch[2 * RPW] == 0) {
    // A row whose columns do not ascend: its workgroup takes the order-independent
    // path and never reads this stream -- but the prefetch of the group before
    // runs into it, so it must hold valid value offsets (zero) and group flags.
    unsigned* w = reinterpret_cast<unsigned*>(my_stream);
    const int nwin = windows + (g == groups - 1 ? kTailWindows : 0);
    for (int i = tid; i < nwin * (kWindowBytes / 4); i += NT)
      w[i] = i % (kWindowBytes / 4) >= 32 ? 0x80808080u : 0u;
    for (int c = tid; c <= nchunks; c += NT) my_info[c] = ChunkInfo{0, 0u, 0u, 0};
    return;
  }

  if (debug & 0x100) return;   // timing experiment: phases 1-2 only
  // (3) per chunk: entries, column groups, the tile rows of the first groups, and what
  // the entries' positions are computed from (held-column mask, entries before each word).
  // One lane per mask WORD (8 consecutive lanes = a chunk): a lane per chunk walked 32
  // bytes serially on a quarter of the workgroup's lanes.
  for (int t0 = 0; t0 < nchunks * WPC; t0 += NT) {
    const int t = t0 + tid;
    const bool live = t < nchunks * WPC;
    const int c = t / WPC, i = t % WPC;
    const unsigned w = live ? maskw[t] : 0u;
    const int cnt = __popc(w);
    unsigned nz = 0u;
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if ((w >> (8 * b)) & 0xffu) nz |= 1u << (4 * i + b);
    int incl = cnt;   // inclusive scan over the chunk's 8 words
#pragma unroll
    for (int o = 1; o < WPC; o <<= 1) {
      const int up = __shfl_up(incl, o, WPC);
      if (i >= o) incl += up;
    }
#pragma unroll
    for (int o = 1; o < WPC; o <<= 1) nz |= static_cast<unsigned>(__shfl_xor(static_cast<int>(nz), o, WPC));
    const int entries = __shfl(incl, WPC - 1, WPC);
    if (live) {
      reinterpret_cast<unsigned char*>(wprefix)[t] = static_cast<unsigned char>(incl - cnt);
      if (i == 0) {
        unsigned first = 0u, rest = nz;
        const int cgroups = __popc(nz);
#pragma unroll
        for (int q = 0; q < kAhead; ++q)
          if (rest) {
            first |= static_cast<unsigned>((c & 1) * BK + __ffs(rest) - 1) << (8 * q);
            rest &= rest - 1u;
          }
        ebase[c] = entries;
        nzmask[c] = nz;
        my_info[c] = ChunkInfo{cgroups, first, 0u, entries};
      }
    }
  }
  if (tid == 0) my_info[nchunks] = ChunkInfo{0, 0u, 0u, 0};   // (read one chunk ahead)
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the entry counts over the chunks
    int carry = 0;
    for (int c0 = 0; c0 < nchunks; c0 += 64) {
      const int c = c0 + lane;
      const int e = c < nchunks ? ebase[c] : 0;
      const int incl = wave_inclusive_scan(e, lane);
      if (c < nchunks) ebase[c] = carry + incl - e;
      carry += __shfl(incl, 63);
    }
  }
  __syncthreads();

  if (debug & 0x200) return;   // timing experiment: phases 1-3 only
  // (4) the entries.  A row's entries land (column, row)-sorted among those of the other 7
  // rows: 64 lanes of a store instruction hit 64 different windows.  Written straight to
  // memory that was two scattered stores (8 bytes + 1 byte) per entry; the group's stream
  // is therefore assembled in LDS when it fits and leaves in 16-byte pieces.
  static_assert(WPC == 8, "two prefix words per chunk");
  const int nwin_out = windows + (g == groups - 1 ? kTailWindows : 0);
  const bool staged = nwin_out <= stage_windows && !(debug & 0x1000);
  unsigned char* const dst_stream = staged ? sbuf : my_stream;
  for (int base = p0; base < p1; base += 64 * UN) {
    int cols[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int p = base + u * 64 + lane;
      cols[u] = p < p1 ? column_indices[p] : -1;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (cols[u] < 0) continue;
      const int p = base + u * 64 + lane;
      const int col = cols[u], c = col / BK, j = col % BK;
      const unsigned w = maskw[c * WPC + j / 4];
      const int prefix = static_cast<int>((wprefix[2 * c + j / 16] >> (8 * ((j / 4) % 4))) & 0xffu) +
                         __popc(w & ((1u << (8 * (j & 3))) - 1u));
      const unsigned byte = (w >> (8 * (j & 3))) & 0xffu;
      const int own = __popc(byte & ((1u << wave) - 1u));
      const int pos = ebase[c] + prefix + own;
      // tile rows (stage parity * BK + column in chunk): byte 0 the entry's own
      // column, byte 1 the column group kAhead groups further on, if the chunk has one
      unsigned rowbyte = wave * 8, tile_rows = static_cast<unsigned>((c & 1) * BK + j);
      if (own == 0) {
        rowbyte |= kNewGroup;
        unsigned ahead = j == BK - 1 ? 0u : nzmask[c] >> (j + 1);   // held columns behind j
        ahead &= ahead - 1u;   // (kAhead = 3: drop the next two)
        ahead &= ahead - 1u;
        if (ahead) tile_rows |= static_cast<unsigned>((c & 1) * BK + j + __ffs(ahead)) << 8;
      }
      if (debug & 0x400) continue;   // timing experiment: no stream writes
      unsigned char* block = dst_stream + static_cast<int64_t>(pos / kWindow) * kWindowBytes;
      const int e = pos % kWindow;
      *reinterpret_cast<uint2*>(block + 8 * e) = make_uint2(tile_rows, static_cast<unsigned>(p) * 4u);
      block[128 + e] = static_cast<unsigned char>(rowbyte);
    }
  }
  // (5) unused entries of the last window, and the empty windows behind the last group:
  // flagged as group starts, so that the loop meets its end-of-chunk test there
  for (int pos = total + tid; pos < windows * kWindow; pos += NT) {
    unsigned char* block = dst_stream + static_cast<int64_t>(pos / kWindow) * kWindowBytes;
    *reinterpret_cast<uint2*>(block + 8 * (pos % kWindow)) = make_uint2(0u, 0u);
    block[128 + pos % kWindow] = static_cast<unsigned char>(kNewGroup);
  }
  if (g == groups - 1) {
    unsigned* w = reinterpret_cast<unsigned*>(dst_stream + static_cast<int64_t>(windows) * kWindowBytes);
    for (int i = tid; i < kTailWindows * (kWindowBytes / 4); i += NT)
      w[i] = i % (kWindowBytes / 4) >= 32 ? 0x80808080u : 0u;
  }
  if (staged) {
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(sbuf);
    uint4* dst = reinterpret_cast<uint4*>(my_stream);   // (window blocks are 144 = 9 x 16 bytes)
    for (int i = tid; i < nwin_out * (kWindowBytes / 16); i += NT) dst[i] = src[i];
  }
}

// ---------------------------------------------------------------------------
// Main kernel: prologue, generated loop, epilogue.
// ---------------------------------------------------------------------------
using v8f = float __attribute__((ext_vector_type(8)));

#define SPUTNIK_HIP_FLAT_ASM_OPERANDS                                                              \
      : "={v[64:71]}"(a0), "={v[72:79]}"(a1), "={v[80:87]}"(a2), "={v[88:95]}"(a3),                \
        "={v[96:103]}"(a4), "={v[104:111]}"(a5), "={v[112:119]}"(a6), "={v[120:127]}"(a7)          \
      : "{v0}"(lane_base), "{v1}"(stage_off), "{v2}"(plan_off), "{v3}"(rows_off),                  \
        "{s[36:37]}"(my_stream), "{s[38:39]}"(values), "{s[40:41]}"(my_info), "{s[42:43]}"(dense), \
        "{s44}"(pitch), "{s45}"(kmax), "{s46}"(nchunks), "{s47}"(row0), "{s48}"(lds0),             \
        "{s49}"(debug)                                                                             \
      : "memory", "m0", "scc", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",     \
        "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v32", "v33", "v34", \
        "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
        "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", \
        "v61", "v62", "v63", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", \
        "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", \
        "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80"

// LOOP: 0 = entry loop, 1 = group loop, straight layout, 2 = group loop, diagonal layout
template <int LOOP>
__global__ __launch_bounds__(kWaves * 64) void spmm_flat_kernel(
    int m, int k, int n, int slots, int nchunks, int n_tiles, const int* __restrict__ row_indices,
    const float* __restrict__ values, int64_t values_stride, const ChunkInfo* __restrict__ cinfo,
    const int* __restrict__ gwin, const unsigned char* __restrict__ stream,
    const float* __restrict__ dense, int64_t dense_stride, float* __restrict__ out,
    int64_t out_stride, const int* __restrict__ row_ok, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, int debug, Epilogue epi) {
  constexpr int RPW = kRPW, BM = kBM, BN = kBN, BK = kBK, VEC = 8;
  constexpr int PPR = VEC / 4;   // 1 KiB pieces per B row
  __shared__ float tile[2][BK * BN];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);

  // Workgroup -> (row block, column tile, replica) as spmm_tiled_kernel: each XCD
  // gets a contiguous set of column tiles, so that the row blocks that stage the
  // same B panel share an L2.
  const int bid = blockIdx.x;
  int ntile, mblock;
  int replica = blockIdx.y;
  const int xcd_cols = 1 << ((debug >> 13) & 3);   // column tiles per XCD (xcd_columns below)
  if (xcd_cols > 1 && n_tiles == 8 && (gridDim.x / 8) % xcd_cols == 0) {
    // `xcd_cols` column tiles and 1 / xcd_cols of the row blocks per XCD: a row block's
    // entry stream is read behind 8 / xcd_cols L2s, a column tile of B behind xcd_cols
    const int xcd = bid % 8, i = bid / 8, col_groups = 8 / xcd_cols;
    ntile = (xcd % col_groups) * xcd_cols + i % xcd_cols;
    mblock = (xcd / col_groups) * ((gridDim.x / 8) / xcd_cols) + i / xcd_cols;
  } else if (n_tiles % 8 == 0) {
    const int per_xcd = n_tiles / 8;
    const int xcd = bid % 8, i = bid / 8;
    ntile = xcd * per_xcd + i % per_xcd;
    mblock = i / per_xcd;
  } else {
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
  const int slot0 = mblock * BM + wave * RPW;

  if (!block_rows_ok_wave(row_ok, mblock * BM, BM)) {
    for (int r = 0; r < RPW; ++r) {
      const int entry = dealt_index(slot0 + r, slots, kDealPer);
      if (entry >= m) continue;
      const int row = row_indices[entry];
#pragma unroll
      for (int h = 0; h < PPR; ++h) {
        if (n0 + h * 256 + lane * 4 >= n) continue;
        const int col = min(n0 + h * 256 + lane * 4, n - 4);   // (see the store phase below)
        const float4 acc4 = gather_row_strip(values, column_indices, row_offsets[row],
                                             row_offsets[row + 1], dense + col, n);
        *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + col) =
            apply_epilogue(acc4, epi, row);
      }
    }
    return;
  }

  const int group = slot0 / RPW;
  const unsigned char* my_stream =
      stream + static_cast<int64_t>(__builtin_amdgcn_readfirstlane(gwin[group])) * kWindowBytes;
  const ChunkInfo* my_info = cinfo + static_cast<int64_t>(group) * (nchunks + 1);
  const unsigned tile_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>(AS_LDS(&tile[0][0])));
  const unsigned lane_base = tile_lds + lane * 16;
  // this wave copies piece (wave % PPR) of rows wave / PPR + (16 / PPR) i, i < 4, of every chunk
  const unsigned stage_off =
      static_cast<unsigned>(min(n0 + (wave % PPR) * 256 + lane * 4, n - 4)) * 4u;
  const unsigned plan_off = (lane & 15) * 8, rows_off = 128 + (lane & 3) * 4;
  const int pitch = n * 4, kmax = k - 1, row0 = wave / PPR;
  const int lds0 = static_cast<int>(tile_lds) + (wave / PPR) * (BN * 4) + (wave % PPR) * 1024;

  v8f a0, a1, a2, a3, a4, a5, a6, a7;
  if constexpr (LOOP == 0) {
    asm volatile(
#include "spmm_flat_body_entry.inc"
        SPUTNIK_HIP_FLAT_ASM_OPERANDS);
  } else if constexpr (LOOP == 1) {
    asm volatile(
#include "spmm_flat_body_group_s.inc"
        SPUTNIK_HIP_FLAT_ASM_OPERANDS);
  } else {
    asm volatile(
#include "spmm_flat_body_group_d.inc"
        SPUTNIK_HIP_FLAT_ASM_OPERANDS);
  }

  // accumulator registers: row r of the wave at 64 + VEC * r
  const v8f acc[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int entry = dealt_index(slot0 + r, slots, kDealPer);
    if (entry < m) {
      const int row = row_indices[entry];
#pragma unroll
      for (int h = 0; h < PPR; ++h)
        if (n0 + h * 256 + lane * 4 < n) {
          // ANY n (the reference takes any, src/spmm_cuda.cu:32): a lane whose four
          // columns would cross the end of the row works on the row's LAST four columns
          // instead -- in the copies of B (stage_off) and here alike -- so it recomputes up
          // to three columns of its neighbour and stores the same values again.  Rows of B
          // and C then start at any multiple of 4 bytes: the 16-byte accesses are dword
          // aligned, which global memory accesses need no more than.
          const int col = min(n0 + h * 256 + lane * 4, n - 4);
          const int f = r * VEC + 4 * h;   // first of the four registers
          *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + col) =
              apply_epilogue(make_float4(acc[f / 8][f % 8], acc[f / 8][f % 8 + 1],
                                         acc[f / 8][f % 8 + 2], acc[f / 8][f % 8 + 3]),
                             epi, row);
        }
    }
  }
}

struct FlatPlan {
  int slots, nchunks, n_tiles, groups;
  size_t row_ok_off, cinfo_off, gwin_off, stream_off, bytes;
};

FlatPlan make_flat_plan(int m, int k, int n, int nonzeros) {
  FlatPlan p;
  p.slots = ceil_div(m, kDealPer) * kDealPer;
  p.nchunks = ceil_div(k, kBK);
  p.n_tiles = ceil_div(n, kBN);
  p.groups = p.slots / kRPW;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  p.row_ok_off = 0;
  p.cinfo_off = row_ok_bytes(p.slots);
  p.gwin_off = up(p.cinfo_off + sizeof(ChunkInfo) * static_cast<size_t>(p.nchunks + 1) * p.groups);
  p.stream_off = up(p.gwin_off + sizeof(int) * static_cast<size_t>(p.groups));
  const size_t windows = static_cast<size_t>(nonzeros) / kWindow + p.groups + kTailWindows;
  p.bytes = up(p.stream_off + windows * kWindowBytes);
  return p;
}

// Windows of a group's stream that the fill kernel can assemble in LDS: a quarter above
// the average group (rows are dealt, groups differ by a few per cent), within 64 KiB per
// workgroup together with the tables; a group beyond that writes straight to memory.
int fill_stage_windows(int nchunks, int nonzeros, int groups) {
  const size_t tables = fill_tables_bytes(nchunks);
  if (tables + kWindowBytes > 64 * 1024) return 0;
  const int64_t room = static_cast<int64_t>((64 * 1024 - tables) / kWindowBytes);
  const int64_t want = (static_cast<int64_t>(nonzeros) / kWindow / groups) * 5 / 4 + kTailWindows + 2;
  return static_cast<int>(std::min(room, want));
}
size_t fill_lds_bytes(int nchunks, int stage_windows) {
  return fill_tables_bytes(nchunks) + static_cast<size_t>(stage_windows) * kWindowBytes;
}

// Which loop reads the stream.  p = share of the entries that start a column
// group = (1 - (1 - d)^8) / (8 d) for density d and 8 rows per wave; the group
// loop reads p strips per entry instead of one but takes branches at group
// starts (gen_spmm_flat.py).  Measured at 4096^3 on one GPU (tools/flat_bench.py
// --loops 2,3,4; ms of the kernel alone, entry / group straight / group diagonal;
// last column the visit-per-row kernel of spmm_tiled.hip):
//   density 0.05   0.205  0.234  0.217   0.256
//           0.10   0.329  0.349  0.326   0.351
//           0.15   0.439  0.450  0.431   0.456
//           0.20   0.554  0.546  0.525   0.565
//           0.25   0.674  0.639  0.623   0.675
//           0.50   1.259  1.055  1.069   1.277
// Knob SPUTNIK_HIP_SPMM_SPARSE: 2 / 3 / 4 force entry / group straight / group
// diagonal.
int flat_mode(int m, int k, int nonzeros) {
  const int forced = options().spmm_sparse;
  if (forced >= 2 && forced <= 4) return forced - 2;
  const double d = static_cast<double>(nonzeros) / (static_cast<double>(m) * k);
  return d < 0.12 ? 0 : d < 0.36 ? 2 : 1;
}

}  // namespace

// The shapes the flat kernel can serve: any n whose column tiles are at most a
// quarter padding (n need not be a multiple of 4: see the kernel's store phase), within the pre-pass's LDS (a row mask per
// column) and its per-group prefix sum over the groups before.  (Whether it is
// TAKEN is the dispatcher's matter: spmm_tiled.hip, use_flat.)
// The fill kernel's tables for a long k need more than the default 64 KiB of dynamic LDS,
// asked for once; a device that refuses serves the shapes whose tables fit the default.
// (Part of the shape rule, so that a plan the dispatcher can ask for can always be made:
// spmm_tiled_plan builds this plan ahead of calls that may or may not take it.)
static bool fill_lds_available(int k) {
  static const bool extended = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_flat_fill_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(fill_lds_bytes(kMaxColumns / kBK, 0))) == hipSuccess;
  }();
  return extended || fill_lds_bytes(ceil_div(k, kBK), 0) <= 64 * 1024;
}

bool spmm_flat_applicable(int m, int k, int n, int nonzeros) {
  if (n < 4 || k < kBK || m < 64 || nonzeros < 16 * static_cast<int64_t>(m) ||
      nonzeros >= (1 << 29))
    return false;
  if (static_cast<int64_t>(ceil_div(n, kBN)) * kBN * 3 > static_cast<int64_t>(n) * 4) return false;
  return k <= kMaxColumns && m <= 16384 && fill_lds_available(k);
}

const char* spmm_flat_kernel_name(int m, int k, int nonzeros) {
  static const char* const names[] = {"spmm_flat_kernel<0>", "spmm_flat_kernel<1>", "spmm_flat_kernel<2>"};
  return names[flat_mode(m, k, nonzeros)];
}

// Workgroups of one replica (the dispatcher takes the kernel when they fill the chip).
int64_t spmm_flat_tiles(int m, int n) {
  return static_cast<int64_t>(ceil_div(m, kBM)) * ceil_div(n, kBN);
}

size_t spmm_flat_workspace_bytes(int m, int k, int n, int nonzeros) {
  return make_flat_plan(m, k, n, nonzeros).bytes;
}

int spmm_flat_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                   const int* row_offsets, const int* column_indices, void* workspace,
                   hipStream_t stream) {
  const FlatPlan p = make_flat_plan(m, k, n, nonzeros);
  char* base = static_cast<char*>(workspace);
  if (!fill_lds_available(k)) return SPUTNIK_HIP_UNSUPPORTED;   // (spmm_flat_applicable said so)
  const int stage_windows = fill_stage_windows(p.nchunks, nonzeros, p.groups);
  hipLaunchKernelGGL(spmm_flat_fill_kernel, dim3(p.groups), dim3(kRPW * 64),
                     fill_lds_bytes(p.nchunks, stage_windows), stream, m, k, p.slots, p.nchunks, row_indices,
                     row_offsets, column_indices, reinterpret_cast<int*>(base + p.row_ok_off),
                     reinterpret_cast<ChunkInfo*>(base + p.cinfo_off),
                     reinterpret_cast<int*>(base + p.gwin_off),
                     reinterpret_cast<unsigned char*>(base + p.stream_off), stage_windows,
                     options().spmm_debug);
  return launch_status();
}

// Which workgroups share an XCD (an L2) in the 8-column-tile shape (n = 4096).  Every XCD
// that holds a row block reads its entry stream (13 bytes per entry with the gathered
// value), every XCD that holds a column tile reads that tile's part of B (k x 512 floats):
// with c column tiles per XCD the launch fetches  stream * 8 / c  +  B * c  bytes.  One
// column tile per XCD (rounds 1-4) is the minimum only below density ~0.07; at 0.1 two are
// (measured: 270 -> 240 MB fetched per launch, time unchanged within 0.5 %), at 0.5 four.
// Code in bits 13-14 of the kernel's debug word (bit 15 of SPUTNIK_HIP_SPMM_DEBUG: take
// bits 13-14 of the knob instead -- measurements).
int xcd_columns_code(int k, int n, int nonzeros, int row_blocks) {
  const int forced = options().spmm_debug;
  if (forced & 0x8000) return (forced >> 13) & 3;
  if (ceil_div(n, kBN) != 8) return 0;
  const double stream = 13.0 * nonzeros, b = 4.0 * k * n;
  int best = 0;
  double best_bytes = stream * 8 + b;
  for (int code = 1; code <= 3; ++code) {
    const int c = 1 << code;
    if (row_blocks % c != 0) break;
    const double bytes = stream * 8 / c + b * c;
    if (bytes < 0.97 * best_bytes) {
      best = code;
      best_bytes = bytes;
    }
  }
  return best;
}

int spmm_flat_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                   const float* values, int64_t values_stride, const int* row_offsets,
                   const int* column_indices, const float* dense, int64_t dense_stride, float* out,
                   int64_t out_stride, const void* workspace, hipStream_t stream, Epilogue epi) {
  const FlatPlan p = make_flat_plan(m, k, n, nonzeros);
  const char* base = static_cast<const char*>(workspace);
#define SPUTNIK_HIP_LAUNCH_FLAT(LOOP)                                                              \
  hipLaunchKernelGGL((spmm_flat_kernel<LOOP>), dim3((p.slots / kBM) * p.n_tiles, replicas),        \
                     dim3(kWaves * 64), 0, stream, m, k, n, p.slots, p.nchunks, p.n_tiles,         \
                     row_indices, values, values_stride,                                           \
                     reinterpret_cast<const ChunkInfo*>(base + p.cinfo_off),                       \
                     reinterpret_cast<const int*>(base + p.gwin_off),                              \
                     reinterpret_cast<const unsigned char*>(base + p.stream_off), dense,           \
                     dense_stride, out, out_stride,                                                \
                     reinterpret_cast<const int*>(base + p.row_ok_off), row_offsets,               \
                     column_indices, debug_word, epi)
  const int debug_word = (options().spmm_debug & ~0xe000) |
                         (xcd_columns_code(k, n, nonzeros, p.slots / kBM) << 13);
  switch (flat_mode(m, k, nonzeros)) {
    case 0: SPUTNIK_HIP_LAUNCH_FLAT(0); break;
    case 1: SPUTNIK_HIP_LAUNCH_FLAT(1); break;
    default: SPUTNIK_HIP_LAUNCH_FLAT(2); break;
  }
#undef SPUTNIK_HIP_LAUNCH_FLAT
  return launch_status();
}

}  // namespace sputnik_hip
