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
// become M0-relative), so ONE instruction stream serves every row, and a wave
// consumes a FLAT stream per K chunk: the entries of all of its rows back to
// back, in a software pipeline (the LDS reads of entry j+3 are in flight while
// entry j is multiplied; tools/gpridx_bench.hip: 16 ns per entry and SIMD, what
// the static-register step costs).  Per chunk a wave pays one boundary
// (rendezvous + the copies of the next B tile), whatever the number of rows.
//
// The stream is made by the pre-pass from the topology alone (the plan is valid
// for any values and any number of replicas): per group of RPW rows (one
// wave's), the entries sorted by (K chunk, row, column) in blocks ("windows") of
// 16 -- per entry the LDS byte offset of its B row, the byte offset of its value
// in `values` (gathered by the kernel one window ahead), and the accumulator
// index of its row.  Per output element the summation order is the CSR order,
// as in every other kernel of this library (bitwise the same results).
//
// The main loop is generated assembly (gen_spmm_flat.py -> spmm_flat_body.inc);
// this file holds the pre-pass, the prologue / epilogue around the loop and the
// host side.  Replaces sputnik::CudaSpmm at /root/reference/src/spmm_cuda.cu:49-56
// for the shapes spmm_flat_applicable() names.
#include "options.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {

namespace {

using namespace tiled;

constexpr int kBN = 512;       // columns of C per workgroup
constexpr int kBK = 32;        // rows of B per chunk (two 64 KiB stages)
constexpr int kWaves = 16;     // waves per workgroup
constexpr int kDealPer = 256;  // as spmm_tiled.hip: row slots are dealt in runs of 256
constexpr int kWindow = 16;    // entries per window
constexpr int kWindowBytes = 144;  // 16 x (offset, value offset) + 16 row bytes
constexpr int kTailWindows = 4;    // zero windows behind the last group (prefetch runs ahead)
constexpr int kMaxFlat = 8192;     // chunks x rows per wave that the fill kernel scans in LDS

// ---------------------------------------------------------------------------
// Pre-pass, second half (the first is spmm_chunk_table_kernel<32>): one
// workgroup per group of RPW row slots, one wave per row.
//   ends[g][c]  stream position (in entries) of the group behind K chunk c
//   gwin[g]     first window of the group's stream
//   stream      the windows
// ---------------------------------------------------------------------------
template <int RPW>
__global__ __launch_bounds__(RPW * 64) void spmm_flat_fill_kernel(
    int m, int slots, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const int* __restrict__ table, const int* __restrict__ row_ok, int* __restrict__ ends,
    int* __restrict__ gwin, unsigned char* __restrict__ stream) {
  extern __shared__ int lds[];  // [nchunks * RPW] position deltas, then scratch
  constexpr int NT = RPW * 64;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid % 64, wave = tid / 64;
  const int groups = gridDim.x;
  const int flat = nchunks * RPW;
  int* delta = lds;
  int* scratch = lds + flat;  // NT + 2 words

  auto slot_len = [&](int slot) {
    const int entry = dealt_index(slot, slots, kDealPer);
    if (entry >= m) return 0;
    const int row = row_indices[entry];
    return row_offsets[row + 1] - row_offsets[row];
  };

  // (1) windows of all groups before this one; this group's entry count and state
  int before = 0;
  for (int gp = tid; gp < g; gp += NT) {
    int len = 0;
    for (int r = 0; r < RPW; ++r) len += slot_len(gp * RPW + r);
    before += (len + kWindow - 1) / kWindow;
  }
  scratch[tid] = before;
  if (tid == 0) {
    int len = 0, ok = 1;
    for (int r = 0; r < RPW; ++r) {
      len += slot_len(g * RPW + r);
      ok &= row_ok[g * RPW + r];
    }
    scratch[NT] = len;
    scratch[NT + 1] = ok;
  }
  __syncthreads();
  for (int s = NT / 2; s > 0; s /= 2) {
    if (tid < s) scratch[tid] += scratch[tid + s];
    __syncthreads();
  }
  const int wbase = scratch[0];
  const int total = scratch[NT];
  const bool group_ok = scratch[NT + 1] != 0;
  __syncthreads();
  if (tid == 0) gwin[g] = wbase;
  unsigned char* my_stream = stream + static_cast<int64_t>(wbase) * kWindowBytes;
  const int windows = (total + kWindow - 1) / kWindow;

  if (!group_ok) {
    // A row whose columns do not ascend: its workgroup takes the order-independent
    // path and never reads this stream -- but the prefetch of the group before
    // runs into it, so it must hold valid value offsets: all zero.
    int* w = reinterpret_cast<int*>(my_stream);
    const int words = (windows + (g == groups - 1 ? kTailWindows : 0)) * (kWindowBytes / 4);
    for (int i = tid; i < words; i += NT) w[i] = 0;
    for (int c = tid; c <= nchunks; c += NT) ends[static_cast<int64_t>(g) * (nchunks + 1) + c] = 0;
    return;
  }

  // (2) exclusive scan of the (chunk, row) counts in chunk-major order
  const int per = (flat + NT - 1) / NT;
  const int f0 = tid * per, f1 = min(f0 + per, flat);
  int sum = 0;
  for (int f = f0; f < f1; ++f) {
    const int c = f / RPW, slot = g * RPW + f % RPW;
    sum += table[static_cast<int64_t>(c + 1) * slots + slot] - table[static_cast<int64_t>(c) * slots + slot];
  }
  scratch[tid] = sum;
  __syncthreads();
  for (int off = 1; off < NT; off *= 2) {   // inclusive scan of the per-thread sums
    const int v = tid >= off ? scratch[tid - off] : 0;
    __syncthreads();
    scratch[tid] += v;
    __syncthreads();
  }
  int run = scratch[tid] - sum;
  int* my_ends = ends + static_cast<int64_t>(g) * (nchunks + 1);
  for (int f = f0; f < f1; ++f) {
    const int c = f / RPW, r = f % RPW, slot = g * RPW + r;
    const int first = table[static_cast<int64_t>(c) * slots + slot];
    const int cnt = table[static_cast<int64_t>(c + 1) * slots + slot] - first;
    delta[f] = run - first;   // stream position of the row's entry p in chunk c = delta + p
    run += cnt;
    if (r == RPW - 1) {
      my_ends[c] = run;
      if (c == nchunks - 1) my_ends[nchunks] = run;  // (read one chunk ahead)
    }
  }
  __syncthreads();

  // (3) the entries: wave = row
  {
    const int slot = g * RPW + wave;
    const int entry = dealt_index(slot, slots, kDealPer);
    if (entry < m) {
      const int row = row_indices[entry];
      const int p1 = row_offsets[row + 1];
      for (int p = row_offsets[row] + lane; p < p1; p += 64) {
        const int col = column_indices[p];
        const int pos = delta[(col / kBK) * RPW + wave] + p;
        unsigned char* block = my_stream + static_cast<int64_t>(pos / kWindow) * kWindowBytes;
        const int e = pos % kWindow;
        // LDS byte offset of the B row: chunk parity picks the stage, 2 KiB per row
        *reinterpret_cast<uint2*>(block + 8 * e) =
            make_uint2(static_cast<unsigned>(col % (2 * kBK)) * (kBN * 4), static_cast<unsigned>(p) * 4u);
        block[128 + e] = static_cast<unsigned char>(wave * 8);
      }
    }
  }
  // (4) unused entries of the last window, and the zero windows behind the last group
  for (int pos = total + tid; pos < windows * kWindow; pos += NT) {
    unsigned char* block = my_stream + static_cast<int64_t>(pos / kWindow) * kWindowBytes;
    *reinterpret_cast<uint2*>(block + 8 * (pos % kWindow)) = make_uint2(0u, 0u);
    block[128 + pos % kWindow] = 0;
  }
  if (g == groups - 1) {
    int* w = reinterpret_cast<int*>(my_stream + static_cast<int64_t>(windows) * kWindowBytes);
    for (int i = tid; i < kTailWindows * (kWindowBytes / 4); i += NT) w[i] = 0;
  }
}

// ---------------------------------------------------------------------------
// Main kernel: prologue, generated loop, epilogue.
// ---------------------------------------------------------------------------
using v8f = float __attribute__((ext_vector_type(8)));

template <int RPW>
__global__ __launch_bounds__(kWaves * 64) void spmm_flat_kernel(
    int m, int k, int n, int slots, int nchunks, int n_tiles, const int* __restrict__ row_indices,
    const float* __restrict__ values, int64_t values_stride, const int* __restrict__ ends,
    const int* __restrict__ gwin, const unsigned char* __restrict__ stream,
    const float* __restrict__ dense, int64_t dense_stride, float* __restrict__ out,
    int64_t out_stride, const int* __restrict__ row_ok, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, int debug, Epilogue epi) {
  static_assert(RPW == 8, "accumulator map of the generated loop");
  constexpr int BM = kWaves * RPW;
  __shared__ float tile[2][kBK * kBN];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);

  // Workgroup -> (row block, column tile, replica) as spmm_tiled_kernel: each XCD
  // gets a contiguous set of column tiles, so that the row blocks that stage the
  // same B panel share an L2.
  const int bid = blockIdx.x;
  int ntile, mblock;
  int replica = blockIdx.y;
  if (n_tiles % 8 == 0) {
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

  const int n0 = ntile * kBN;
  const int slot0 = mblock * BM + wave * RPW;

  if (!block_rows_ok_wave(row_ok, mblock * BM, BM)) {
    for (int r = 0; r < RPW; ++r) {
      const int entry = dealt_index(slot0 + r, slots, kDealPer);
      if (entry >= m) continue;
      const int row = row_indices[entry];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = n0 + h * 256 + lane * 4;
        if (col >= n) continue;
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
  const int* my_ends = ends + static_cast<int64_t>(group) * (nchunks + 1);
  const unsigned tile_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>(AS_LDS(&tile[0][0])));
  const unsigned lane_base = tile_lds + lane * 16;
  // this wave copies piece (wave % 2) of rows wave / 2 + 8 i of every chunk
  const unsigned stage_off = static_cast<unsigned>(min(n0 + (wave & 1) * 256 + lane * 4, n - 4)) * 4u;
  const unsigned plan_off = (lane & 15) * 8, rows_off = 128 + (lane & 3) * 4;
  const int pitch = n * 4, kmax = k - 1, row0 = wave / 2;
  const int lds0 = static_cast<int>(tile_lds) + (wave / 2) * (kBN * 4) + (wave & 1) * 1024;

  v8f a0, a1, a2, a3, a4, a5, a6, a7;
  asm volatile(
#include "spmm_flat_body.inc"
      : "={v[64:71]}"(a0), "={v[72:79]}"(a1), "={v[80:87]}"(a2), "={v[88:95]}"(a3),
        "={v[96:103]}"(a4), "={v[104:111]}"(a5), "={v[112:119]}"(a6), "={v[120:127]}"(a7)
      : "{v0}"(lane_base), "{v1}"(stage_off), "{v2}"(plan_off), "{v3}"(rows_off),
        "{s[36:37]}"(my_stream), "{s[38:39]}"(values), "{s[40:41]}"(my_ends), "{s[42:43]}"(dense),
        "{s44}"(pitch), "{s45}"(kmax), "{s46}"(nchunks), "{s47}"(row0), "{s48}"(lds0), "{s49}"(debug)
      : "memory", "m0", "scc", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",
        "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v32", "v33", "v34",
        "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
        "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60",
        "v61", "v62", "v63", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59",
        "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72",
        "s73", "s74", "s75");

  const v8f acc[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int entry = dealt_index(slot0 + r, slots, kDealPer);
    if (entry < m) {
      const int row = row_indices[entry];
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (n0 + h * 256 + lane * 4 < n)
          *reinterpret_cast<float4*>(out + static_cast<int64_t>(row) * n + n0 + h * 256 + lane * 4) =
              apply_epilogue(make_float4(acc[r][4 * h], acc[r][4 * h + 1], acc[r][4 * h + 2],
                                         acc[r][4 * h + 3]),
                             epi, row);
    }
  }
}

struct FlatPlan {
  int slots, nchunks, n_tiles, groups;
  size_t row_ok_off, table_off, ends_off, gwin_off, stream_off, bytes;
};

FlatPlan make_flat_plan(int m, int k, int n, int nonzeros) {
  constexpr int RPW = 8;
  FlatPlan p;
  p.slots = ceil_div(m, kDealPer) * kDealPer;
  p.nchunks = ceil_div(k, kBK);
  p.n_tiles = ceil_div(n, kBN);
  p.groups = p.slots / RPW;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  p.row_ok_off = 0;
  p.table_off = row_ok_bytes(p.slots);
  p.ends_off = up(p.table_off + sizeof(int) * static_cast<size_t>(p.nchunks + 1) * p.slots);
  p.gwin_off = up(p.ends_off + sizeof(int) * static_cast<size_t>(p.nchunks + 1) * p.groups);
  p.stream_off = up(p.gwin_off + sizeof(int) * static_cast<size_t>(p.groups));
  const size_t windows = static_cast<size_t>(nonzeros) / kWindow + p.groups + kTailWindows;
  p.bytes = up(p.stream_off + windows * kWindowBytes);
  return p;
}

}  // namespace

// The shapes the flat kernel takes: those of the 128 x 512 tile of
// spmm_tiled.hip when one replica alone gives about one workgroup per CU (the
// plan then does not depend on the replica count), within the fill kernel's LDS
// scan and its per-group prefix sum.
bool spmm_flat_applicable(int m, int k, int n, int nonzeros) {
  if (n % 4 != 0 || k < kBK || m < 64 || nonzeros < 16 * static_cast<int64_t>(m) ||
      nonzeros >= (1 << 29))
    return false;
  const int64_t tiles = static_cast<int64_t>(ceil_div(m, kWaves * 8)) * ceil_div(n, kBN);
  if (static_cast<int64_t>(ceil_div(n, kBN)) * kBN * 3 > static_cast<int64_t>(n) * 4) return false;
  return tiles >= 192 && ceil_div(k, kBK) * 8 <= kMaxFlat && m <= 16384;
}

size_t spmm_flat_workspace_bytes(int m, int k, int n, int nonzeros) {
  return make_flat_plan(m, k, n, nonzeros).bytes;
}

int spmm_flat_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                   const int* row_offsets, const int* column_indices, void* workspace,
                   hipStream_t stream) {
  constexpr int RPW = 8;
  const FlatPlan p = make_flat_plan(m, k, n, nonzeros);
  char* base = static_cast<char*>(workspace);
  int* row_ok = reinterpret_cast<int*>(base + p.row_ok_off);
  int* table = reinterpret_cast<int*>(base + p.table_off);
  hipLaunchKernelGGL((spmm_chunk_table_kernel<kBK>), dim3(ceil_div(p.slots, 4)), dim3(256), 0,
                     stream, m, k, p.slots, kDealPer, p.nchunks, row_indices, row_offsets,
                     column_indices, table, row_ok);
  int st = launch_status();
  if (st != 0) return st;
  const size_t lds = sizeof(int) * (static_cast<size_t>(p.nchunks) * RPW + RPW * 64 + 2);
  hipLaunchKernelGGL((spmm_flat_fill_kernel<RPW>), dim3(p.groups), dim3(RPW * 64), lds, stream, m,
                     p.slots, p.nchunks, row_indices, row_offsets, column_indices, table, row_ok,
                     reinterpret_cast<int*>(base + p.ends_off),
                     reinterpret_cast<int*>(base + p.gwin_off),
                     reinterpret_cast<unsigned char*>(base + p.stream_off));
  return launch_status();
}

int spmm_flat_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                   const float* values, int64_t values_stride, const int* row_offsets,
                   const int* column_indices, const float* dense, int64_t dense_stride, float* out,
                   int64_t out_stride, const void* workspace, hipStream_t stream, Epilogue epi) {
  constexpr int RPW = 8;
  const FlatPlan p = make_flat_plan(m, k, n, nonzeros);
  const char* base = static_cast<const char*>(workspace);
  hipLaunchKernelGGL((spmm_flat_kernel<RPW>), dim3((p.slots / (kWaves * RPW)) * p.n_tiles, replicas),
                     dim3(kWaves * 64), 0, stream, m, k, n, p.slots, p.nchunks, p.n_tiles,
                     row_indices, values, values_stride,
                     reinterpret_cast<const int*>(base + p.ends_off),
                     reinterpret_cast<const int*>(base + p.gwin_off),
                     reinterpret_cast<const unsigned char*>(base + p.stream_off), dense,
                     dense_stride, out, out_stride, reinterpret_cast<const int*>(base + p.row_ok_off),
                     row_offsets, column_indices, options().spmm_debug, epi);
  return launch_status();
}

}  // namespace sputnik_hip
