// LDS-tiled SDDMM for inner dimensions k that are a multiple of 64 (attention
// heads; the weight gradient of SparseLinear / Spmm, where k is the sequence
// length or the dense operand's width), walked in at most 8 panels of 64, 128,
// 256 or 512 columns (one launch per panel, later panels add into the output):
//
//   out[p] = < lhs[i_p, 0:k], rhs[j_p, 0:k] >   for every stored (i_p, j_p)
//
// The row-wave kernel of sddmm.hip gathers one rhs row per nonzero from L2
// (nnz*k*4 bytes of cache traffic: 1.7 GB for config 3, 13.7 GB for config 5,
// and it runs at the L2 gather rate).  Here the rhs operand is STATIONARY: a
// workgroup keeps one slab of rhs rows, full k wide, in LDS for its whole
// life (64 KiB up to panel width 128, 128 KiB above: 256 / 128 / 128 / 64 rows),
// staged once by direct global->LDS copies behind a single barrier, and walks
// mask rows instead.  For every mask row only the entries whose column falls
// in the slab are computed (found with the chunk table of the shared
// pre-pass); outputs of different slabs are disjoint, so nothing is reduced
// across workgroups.  grid = (slabs, row blocks of 256, replicas).
//
//   * each 16-lane row group of a wave owns one mask row at a time; its lhs row
//     lives in registers (k/16 floats per lane), fetched one or two rows ahead;
//   * the row's column indices inside the slab arrive 16 at a time (lane =
//     entry) and are handed out with DPP row_newbcast;
//   * per entry: one ds_read_b128 + 4 FMA per 64 inner elements and lane;
//   * per 16-entry window ONE transposing DPP reduction (row_transpose_sum16)
//     leaves entry u's sum in lane u, so 16 results leave as one 64-byte store.
// Needs ascending columns inside rows (checked per row by the pre-pass); a row
// that fails is computed whole by slab 0 with rhs gathered from global memory.
#include <type_traits>

#include "options.h"
#include "sddmm_dot.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {

// sddmm_flat.hip: the pair-flat kernel for planned masks and rows of 128 / 256 bytes
bool sddmm_flat_applicable(int m, int k, int n, int nonzeros, int elem_bytes);
size_t sddmm_flat_plan_bytes(int m, int n, int nonzeros);
int sddmm_flat_plan(int m, int n, int nonzeros, const int* row_indices, const int* row_offsets,
                    const int* column_indices, void* plan, hipStream_t stream);
int sddmm_flat_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                      const void* lhs, int64_t lhs_stride, const void* rhs, int64_t rhs_stride,
                      void* out, int64_t out_stride, int in_type, int out_type, const void* plan,
                      hipStream_t stream);

namespace {

using namespace tiled;

constexpr int kSWaves = 8;
constexpr int kSGroups = kSWaves * 4;  // 16-lane row groups per workgroup
constexpr int kSRows = 8;              // mask rows per group: a workgroup owns 256 rows
constexpr int kWin = 2;                // 16-entry column windows fetched ahead per row
// Bit of the kernels' `debug` word, set by the launcher (never by the knob): the launch holds
// ALL replicas of a many-mask batch and its plan carries the masks' start order.
constexpr int kMasksLargestFirst = 1 << 30;
// Grid y of the stationary kernel: all row blocks while the launch stays within about three
// rounds of the chip; beyond that a workgroup walks several row blocks against the slab it
// staged once -- the prologue (bookkeeping hops + up to 128 KiB of slab before anything is
// computed, 5 us during which its CU idles) is then paid once per workgroup instead of once
// per row block.  Measured (tools/sddmm_rows_bench.py, all row blocks / walked, us; boxes
// and repeats differ by +-5 %): 2048^2 at density 0.2, k = 512 x 8 replicas, plain 262 / 221,
// summed (config 5's weight gradient) 248 / 226-243; 4096^2 x 256 x 4 plain 165 / 130; 1024^2
// at 0.3, k = 1024 x 8 summed 177 / 151; the quad kernel (k = 64) 2048^2 at 0.1 x 64 replicas
// 165 / 145, 4096^2 at 0.05 x 16: 125 / 111; the sparse end pays a little (2048^2 at 0.05
// summed 130 / 136).  SPUTNIK_HIP_SDDMM_DEBUG bits 20..: that many hundred workgroups instead.
inline int launch_rows_y(int slabs, int row_blocks, int64_t z, int debug, int slab_bytes) {
  // (a many-mask launch starts its masks largest first: its order is that of the grid)
  if (debug & kMasksLargestFirst) return row_blocks;
  const int64_t per_y = static_cast<int64_t>(slabs) * z, total = per_y * row_blocks;
  const int knob = (debug >> 20) & 0x3ff;   // (bits 20-29 of the knob: hundreds of workgroups)
  // From four rounds of the chip on (two workgroups per CU with slabs of at most 80 KiB): config
  // 3's quad kernel, 1024 workgroups in two rounds, read 47.2 us as it is and 48.6 walked.
  // Down to 512 workgroups; the 80-row slabs of the summed form measured best with 1536.
  const int64_t per_round = 256 * (slab_bytes <= 80 * 1024 ? 2 : 1);
  if (knob == 0 && total < 4 * per_round) return row_blocks;
  const int64_t target = knob > 0 ? int64_t{100} * knob : slab_bytes == 80 * 1024 ? 1536 : 512;
  if (total <= target) return row_blocks;
  return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(row_blocks, (target + per_y - 1) / per_y)));
}

constexpr int default_slab_rows(int kv) { return (kv <= 2 ? 64 * 1024 : 128 * 1024) / (64 * kv * 4); }

// ROWS: rows of the slab.  The default is the 64 / 128 KiB slab; the SUMMED product's
// 256-wide panels also come with 80 rows (80 KiB, 8 waves): two workgroups per CU, so
// that one's prologue -- two dependent hops of row bookkeeping and the slab itself, 5 us
// during which a lone workgroup's CU computes nothing -- runs under the other's
// arithmetic (sum_slab_rows below says when).
template <int KV, int ROWS = default_slab_rows(KV)>
struct Slab {
  static constexpr int kdim = 64 * KV;
  static constexpr int kRows = ROWS;
  static constexpr int kBytes = kRows * kdim * 4;
  // lhs fragments in flight (KV float4 each): fetched 2 rows / 1 row ahead
  static constexpr int kRing = KV <= 2 ? 3 : 2;
  // Waves per workgroup.  A 128 KiB slab admits one workgroup per CU; with 8
  // waves that is two per SIMD, each issuing VALU work in 23 % of its cycles
  // (rocprofv3 SQ counters: the SIMD idles half the time).  Panel width 256 needs
  // few enough registers (<= 128) for four waves per SIMD, so its workgroup has
  // 16 waves that share the slab, each group walking 4 mask rows instead of 8
  // (the workgroup still owns 256 rows: plans and tables are unchanged).
  static constexpr int kWaves = (KV == 4 && kBytes > 80 * 1024) ? 2 * kSWaves : kSWaves;
  static constexpr int kThreads = kWaves * kWave;
  static constexpr int kGroups = kWaves * 4;            // 16-lane row groups per workgroup
  static constexpr int kGroupRows = kSGroups * kSRows / kGroups;   // mask rows per group
  static_assert(kRows >= 16, "a slab holds at least one full column window");
};

// Occupancy is set by LDS: two 64 KiB workgroups (4 waves per SIMD) or one of
// 128 KiB (2 waves per SIMD); telling the compiler stops it from spilling to
// keep a wave count the LDS footprint rules out anyway.
// ACC: a later k panel adds into the output.  A compile-time flag on purpose: as a
// run-time one the conditional load of the previous value made the compiler put
// `s_waitcnt vmcnt(0)` in front of EVERY result store, i.e. every window waited for
// all rows fetched ahead and all earlier stores (10 us of 52 at config 3).
template <int KV, bool ACC, int ROWS = default_slab_rows(KV)>
__global__ __launch_bounds__((Slab<KV, ROWS>::kThreads))
__attribute__((amdgpu_waves_per_eu(2, (KV <= 2 || KV == 4 ? 4 : 2))))
void sddmm_stationary_kernel(
    int m, int n, int nonzeros, int slots, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const int* __restrict__ table, const int* __restrict__ row_ok,
    const float* __restrict__ lhs, int64_t lhs_stride, const float* __restrict__ rhs,
    int64_t rhs_stride, int ld /* floats between rows of lhs / rhs */,
    float* __restrict__ out, int64_t out_stride, int panels /* of one replica along grid z */,
    int debug, int mask_heads, int64_t mask_plan_ints, int first_replica) {
  using S = Slab<KV, ROWS>;
  constexpr int kdim = S::kdim;  // panel width; lhs / rhs point at the panel's first column
  constexpr int kRowBytes = kdim * 4;
  __shared__ float tile[S::kBytes / 4];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  // (slab fastest, then row block, then replica / panel: the workgroups that share
  // an lhs row block or an rhs slab sit behind one XCD's L2, see xcd_local_index)
  const unsigned long long work = mask_heads > 0 && panels == 1   // (many masks: their replicas dealt over the XCDs)
                                      ? xcd_spread_replicas_index(gridDim.x * gridDim.y, gridDim.z)
                                      : xcd_local_index();
  const int slab = static_cast<int>(work % gridDim.x);
  // (round 5: grid y may be SMALLER than the number of row blocks -- a workgroup then walks
  // the row blocks y, y + gridDim.y, ... against the slab it staged once, see launch_rows_y)
  const int first_row_block = static_cast<int>((work / gridDim.x) % gridDim.y);
  int grid_z = static_cast<int>(work / (static_cast<unsigned long long>(gridDim.x) * gridDim.y));
  if (debug & kMasksLargestFirst)   // (many masks, all of them in this launch)
    grid_z = row_ok[mask_start_word(grid_z / mask_heads, mask_plan_ints)] * mask_heads + grid_z % mask_heads;
  // grid z = replica * panels + panel (panels == 1: the launch is one panel of
  // every replica; > 1: all panels at once, each into its own output, see
  // sddmm_tiled_launch_partials)
  const int z = grid_z;
  const int replica = panels > 1 ? z / panels : z;
  const int panel = z - replica * panels;
  lhs += replica * lhs_stride + panel * kdim;
  rhs += replica * rhs_stride + panel * kdim;
  out += z * out_stride;
  const int jc = slab * S::kRows;
  {  // "many mask": this replica's topology and plan (common.h, select_mask)
    const MaskPlace place = select_mask(mask_heads, first_replica + replica, m, nonzeros, row_offsets);
    row_offsets += static_cast<int64_t>(place.mask) * (m + 1);
    column_indices += place.first;
    row_indices += static_cast<int64_t>(place.mask) * m;
    table += place.mask * mask_plan_ints;
    row_ok += place.mask * mask_plan_ints;
    nonzeros = place.nonzeros;
  }
  const int last = max(nonzeros - 1, 0);

  if (!(debug & 2)) {  // stage the slab in 1 KiB pieces (64 lanes x 16 B, lane-linear in LDS)
    constexpr int kPieces = S::kBytes / 1024;
    static_assert(kPieces % S::kWaves == 0, "whole pieces per wave");
#pragma unroll
    for (int j = 0; j < kPieces / S::kWaves; ++j) {
      const int piece = wave + j * S::kWaves;
      const unsigned b = static_cast<unsigned>(piece) * 1024u + lane * 16u;  // byte in the slab
      const int src_row = min(jc + static_cast<int>(b / kRowBytes), n - 1);  // past the end: last row
      const unsigned off =
          static_cast<unsigned>(src_row) * (static_cast<unsigned>(ld) * 4u) + b % kRowBytes;
      lds_dma_row(rhs, off, tile + piece * 256);
    }
  }

  // This group's kSRows mask rows.  All their bookkeeping (row id, first entry
  // inside the slab, count) is fetched in one go while the slab is still in
  // flight; a row's column windows and lhs fragment are fetched kRing-1 rows
  // ahead.  Everything is statically indexed (the row loop is fully
  // unrolled), so no register is ever copied while its load is outstanding.
  const int gid = wave * 4 + g;
  const int* __restrict__ tab0 = table + static_cast<int64_t>(slab) * slots;
  const int* __restrict__ tab1 = tab0 + slots;
  const char* __restrict__ lane_base = reinterpret_cast<const char*>(&tile[0] + i * 4);
  const int row_blocks = slots / (kSGroups * kSRows);
  for (int row_block = first_row_block; row_block < row_blocks; row_block += gridDim.y) {
  // (the NEXT block's bookkeeping requested while this one is worked on was tried: 16 more
  // registers, which the 80-row slab's two workgroups per CU do not have -- 223 -> 248 us at
  // config 5's weight gradient)
  const int slot_begin = row_block * (kSGroups * kSRows);

  constexpr int kRowsHere = S::kGroupRows;
  int row[kRowsHere], ps[kRowsHere], cnt[kRowsHere];
#pragma unroll
  for (int r = 0; r < kRowsHere; ++r) {
    const int sl = slot_begin + r * S::kGroups + gid;  // < slots: table and status are padded
    const int entry = dealt_index(sl, slots, kSGroups * kSRows);
    const bool live = entry < m;
    row[r] = row_indices[live ? entry : 0];
    const int a0 = tab0[sl], a1 = tab1[sl];
    // rows whose columns do not ascend have no valid table entries: slab 0 does
    // the whole row in storage order (negative count), the other slabs skip it
    const bool ok = row_ok[sl] != 0;
    ps[r] = ok ? a0 : row_offsets[row[r]];
    const int len = ok ? a1 - a0 : row_offsets[row[r] + 1] - ps[r];
    cnt[r] = !live ? 0 : ok ? len : (slab == 0 ? -len : 0);
  }

  constexpr int kRing = S::kRing;
  int wcol[kRing][kWin];
  float4 lf[kRing][KV];
  auto fetch = [&](int r, int slot_in_ring) {
#pragma unroll
    for (int w = 0; w < kWin; ++w)
      wcol[slot_in_ring][w] = column_indices[min(ps[r] + 16 * w + i, last)];
#pragma unroll
    for (int v = 0; v < KV; ++v)
      lf[slot_in_ring][v] = *reinterpret_cast<const float4*>(
          lhs + static_cast<int64_t>(row[r]) * ld + 64 * v + 4 * i);
  };
#pragma unroll
  for (int r = 0; r < kRing - 1; ++r) fetch(r, r);
  wait_vm<0>();
  if (row_block == first_row_block) __syncthreads();   // (the slab is there: once per workgroup)

  static_for<kRowsHere>([&](auto R) {
    constexpr int r = decltype(R)::value;
    if constexpr (r + kRing - 1 < kRowsHere) fetch(r + kRing - 1, (r + kRing - 1) % kRing);
    const float4 (&cur_lf)[KV] = lf[r % kRing];
    const int cur_ps = ps[r];

    // this lane's share (4 floats per 64) of <lhs row, rhs row at `p`>
    auto load_row = [&](float4 (&b)[KV], const char* __restrict__ p) {
#pragma unroll
      for (int v = 0; v < KV; ++v) b[v] = *reinterpret_cast<const float4*>(p + 256 * v);
    };
    // (two interleaved partial sums: the even and the odd elements, so that every
    // pair of multiply-adds is ONE v_pk_fma_f32 -- a single chain leaves the
    // compiler nothing to pack: 4 v_fma_f32 per ds_read_b128 before, 2 packed now)
    auto dot = [&](const float4 (&b)[KV]) {
      v2f acc = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        acc = __builtin_elementwise_fma(v2f{cur_lf[v].x, cur_lf[v].y}, v2f{b[v].x, b[v].y}, acc);
        acc = __builtin_elementwise_fma(v2f{cur_lf[v].z, cur_lf[v].w}, v2f{b[v].z, b[v].w}, acc);
      }
      return acc.x + acc.y;
    };
    // Four entries.  k = 64: all reads in flight at once.  Longer
    // ones: two register sets, the reads of entry e+1 issued before the FMAs of
    // entry e.
    auto partial4 = [&](const char* a0, const char* a1, const char* a2, const char* a3, float& d0,
                        float& d1, float& d2, float& d3) {
      if constexpr (KV <= 1) {
        float4 b0[KV], b1[KV], b2[KV], b3[KV];
        load_row(b0, a0);
        load_row(b1, a1);
        load_row(b2, a2);
        load_row(b3, a3);
        d0 = dot(b0);
        d1 = dot(b1);
        d2 = dot(b2);
        d3 = dot(b3);
      } else {
        // Pinned order: left alone, the compiler sinks the FMAs down to the
        // reduction and hoists the reads of many entries at once (70
        // ds_read_b128 in a row at k = 512), which spills.  The empty asm
        // statements tie each partial sum (and, through the memory clobber,
        // the following reads) to its place.
        float4 ba[KV], bb[KV];
        load_row(ba, a0);
        load_row(bb, a1);
        d0 = dot(ba);
        asm volatile("" : "+v"(d0) : : "memory");
        load_row(ba, a2);
        d1 = dot(bb);
        asm volatile("" : "+v"(d1) : : "memory");
        load_row(bb, a3);
        d2 = dot(ba);
        asm volatile("" : "+v"(d2) : : "memory");
        d3 = dot(bb);
        asm volatile("" : "+v"(d3) : : "memory");
      }
    };

    if (cnt[r] < 0) {
      // unsorted row (rare): rhs rows gathered from global memory, any column
      const int p1 = cur_ps - cnt[r];
      for (int p = cur_ps; p < p1; ++p) {
        float4 b[KV];
        load_row(b, reinterpret_cast<const char*>(
                        rhs + static_cast<int64_t>(column_indices[p]) * ld + 4 * i));
        const float total = group_sum<16>(dot(b));
        if (i == 0) out[p] = ACC ? out[p] + total : total;
      }
    }
    const int n_here = (debug & 1) ? 0 : max(cnt[r], 0);
    auto window = [&](int ecol, int w0) {
      const int left = n_here - w0;
      const bool valid = i < left;
      const int roff = valid ? ((ecol - jc) * kRowBytes) : 0;
      float result = 0.f;
      if (left > 4) {
        // 5..16 entries: all partials first, then ONE transposing reduction
        // that leaves entry u's sum in lane u
        float p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) p[u] = 0.f;
        auto four = [&](auto G) {
          constexpr int kG = decltype(G)::value;
          partial4(lane_base + row_bcast_i<kG + 0>(roff), lane_base + row_bcast_i<kG + 1>(roff),
                   lane_base + row_bcast_i<kG + 2>(roff), lane_base + row_bcast_i<kG + 3>(roff),
                   p[kG + 0], p[kG + 1], p[kG + 2], p[kG + 3]);
        };
        four(std::integral_constant<int, 0>{});
        four(std::integral_constant<int, 4>{});
        if (left > 8) four(std::integral_constant<int, 8>{});
        if (left > 12) four(std::integral_constant<int, 12>{});
        result = row_transpose_sum16(p, i);
      } else if (left > 0) {
        float d0, d1, d2, d3;
        partial4(lane_base + row_bcast_i<0>(roff), lane_base + row_bcast_i<1>(roff),
                 lane_base + row_bcast_i<2>(roff), lane_base + row_bcast_i<3>(roff), d0, d1, d2,
                 d3);
        const float t0 = group_sum<16>(d0), t1 = group_sum<16>(d1);
        const float t2 = group_sum<16>(d2), t3 = group_sum<16>(d3);
        result = (i == 0) ? t0 : (i == 1) ? t1 : (i == 2) ? t2 : t3;
      }
      if (valid) {
        float* dst = out + cur_ps + w0 + i;
        if constexpr (ACC) *dst += result;
        else *dst = result;
      }
    };
#pragma unroll
    for (int w = 0; w < kWin; ++w) window(wcol[r % kRing][w], 16 * w);
    // more than 32 entries of one row inside the slab: fetch on demand
    const int longest =
        max(max(__builtin_amdgcn_readlane(n_here, 0), __builtin_amdgcn_readlane(n_here, 16)),
            max(__builtin_amdgcn_readlane(n_here, 32), __builtin_amdgcn_readlane(n_here, 48)));
    for (int w0 = 16 * kWin; w0 < longest; w0 += 16)
      window(column_indices[min(cur_ps + w0 + i, last)], w0);
  });
  }   // (row blocks of this workgroup)
}

// ----------------------------------------------------------------------------
// Quad form (round 3): FOUR lanes share an entry instead of sixteen.
//
// The kernel above spends, per entry and 16-lane group, one broadcast, one
// ds_read_b128, two v_pk_fma_f32 -- and then 1.9 selects, 0.9 DPP adds and the
// s_nop hazards of the transposing reduction that folds 16 partial sums per lane:
// 9.4 issued instructions for 2 useful ones at k = 64 (ISA count), a SIMD 92 % busy
// issuing.  Here a 16-lane group still owns one mask row at a time, but its four
// QUADS work on four different entries of the row: lane (quad q, t) keeps a QUARTER
// of the lhs row (k/4 elements) in registers, reads the matching quarter of the rhs
// row of "its" entry from LDS and the dot product is closed by two quad_perm DPP
// adds (one VALU instruction each for all 16 entries a wave has in flight): per
// step of 16 entries and wave C address adds (the quad broadcast of the row offset
// rides on them as a DPP operand), C ds_read_b128, 2C v_pk_fma_f32 (float; C
// v_dot2 pairs for the half types), 4 reduction / select instructions -- about one
// instruction per entry at k = 64.
//
//   * lane (q, t) of a group holds the column of entry 4t + q of each 16-entry
//     window, so the row offset the quads need in step s (entries 4s + q) is a
//     quad_perm:[s,s,s,s] broadcast, and after four steps lane (q, t) holds the
//     result of entry 4t + q: the 16 results leave as one 64-byte store;
//   * bank conflicts: the 16 lanes of a group read four different rhs rows in one
//     instruction; rows are multiples of 256 bytes (all 64 banks), so lane (q, t)
//     takes its quarter's 16-byte chunks in the order (c + q [+ 4t..]) mod C and
//     the 16 lanes together touch every bank exactly once (the lhs quarter is
//     loaded in the same rotated order, once per mask row).
//
// Storage type T: float, or _Float16 / __bf16 (native half operands, round 3): the
// slab is staged as raw bytes (half the LDS DMA and half the LDS read traffic), the
// products are v_dot2_f32_f16 / v_dot2_f32_bf16 -- exact products, float32 sums.
// The slab keeps the ROW COUNT of the float form, so plans do not depend on T.
template <int KV, int ROWS, typename T>
struct Quad {
  static constexpr int kdim = 64 * KV;
  static constexpr int kRowBytes = kdim * static_cast<int>(sizeof(T));
  static constexpr int kRows = ROWS;   // of the float slab the PLAN was made for: plans are shared
  static constexpr int kBytes = kRows * kRowBytes;
  static constexpr int kQuarter = kRowBytes / 4;          // bytes of a row per lane
  static constexpr int C = kQuarter / 16;                 // 16-byte chunks per lane
  static constexpr int kWaves = kSWaves;
  static constexpr int kThreads = kWaves * kWave;
  static_assert(C >= 2 && C <= 8, "quad form: 32 .. 128 bytes of a row per lane");
};

template <int KV, int ROWS, typename T, typename TO, bool ACC>
__global__ __launch_bounds__((Quad<KV, ROWS, T>::kThreads))
__attribute__((amdgpu_waves_per_eu((Quad<KV, ROWS, T>::C <= 4 ? 4 : 2), 4)))   // <= 128 registers: two workgroups per CU
void sddmm_quad_kernel(
    int m, int n, int nonzeros, int slots, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const int* __restrict__ table, const int* __restrict__ row_ok,
    const T* __restrict__ lhs, int64_t lhs_stride, const T* __restrict__ rhs,
    int64_t rhs_stride, int ld /* elements between rows of lhs / rhs */,
    TO* __restrict__ out, int64_t out_stride, int panels /* of one replica along grid z */,
    int debug, int mask_heads, int64_t mask_plan_ints, int first_replica) {
  using Q = Quad<KV, ROWS, T>;
  using chunk = typename Dot<T>::chunk;
  constexpr int C = Q::C;
  constexpr int kRowBytes = Q::kRowBytes;
  __shared__ __attribute__((aligned(1024))) char tile[Q::kBytes];
  __shared__ __attribute__((aligned(16))) char line[Q::kWaves * 4 * kRowBytes];   // lhs rows in transit

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, t = i & 3;       // quad of the group, lane of the quad
  const int e = 4 * t + q;               // this lane's entry of a 16-entry window
  const unsigned long long work = mask_heads > 0 && panels == 1   // (many masks: their replicas dealt over the XCDs)
                                      ? xcd_spread_replicas_index(gridDim.x * gridDim.y, gridDim.z)
                                      : xcd_local_index();
  const int slab = static_cast<int>(work % gridDim.x);
  // (grid y may be smaller than the number of row blocks: launch_rows_y)
  const int first_row_block = static_cast<int>((work / gridDim.x) % gridDim.y);
  int z = static_cast<int>(work / (static_cast<unsigned long long>(gridDim.x) * gridDim.y));
  if (debug & kMasksLargestFirst)   // (many masks, all of them in this launch)
    z = row_ok[mask_start_word(z / mask_heads, mask_plan_ints)] * mask_heads + z % mask_heads;
  const int replica = panels > 1 ? z / panels : z;
  const int panel = z - replica * panels;
  lhs += replica * lhs_stride + panel * Q::kdim;
  rhs += replica * rhs_stride + panel * Q::kdim;
  out += z * out_stride;
  const int jc = slab * Q::kRows;
  {
    const MaskPlace place = select_mask(mask_heads, first_replica + replica, m, nonzeros, row_offsets);
    row_offsets += static_cast<int64_t>(place.mask) * (m + 1);
    column_indices += place.first;
    row_indices += static_cast<int64_t>(place.mask) * m;
    table += place.mask * mask_plan_ints;
    row_ok += place.mask * mask_plan_ints;
    nonzeros = place.nonzeros;
  }
  const int last = max(nonzeros - 1, 0);

  if (!(debug & 2)) {  // stage the slab in 1 KiB pieces (64 lanes x 16 B, lane-linear in LDS)
    constexpr int kPieces = Q::kBytes / 1024;
    static_assert(kPieces % Q::kWaves == 0, "whole pieces per wave");
#pragma unroll
    for (int j = 0; j < kPieces / Q::kWaves; ++j) {
      const int piece = wave + j * Q::kWaves;
      const unsigned b = static_cast<unsigned>(piece) * 1024u + lane * 16u;  // byte in the slab
      const int src_row = min(jc + static_cast<int>(b / kRowBytes), n - 1);  // past the end: last row
      const unsigned off = static_cast<unsigned>(src_row) *
                               (static_cast<unsigned>(ld) * static_cast<unsigned>(sizeof(T))) +
                           b % kRowBytes;
      lds_dma_row(reinterpret_cast<const float*>(rhs), off,
                  reinterpret_cast<const float*>(tile + piece * 1024));
    }
  }

  const int gid = wave * 4 + g;
  const int* __restrict__ tab0 = table + static_cast<int64_t>(slab) * slots;
  const int* __restrict__ tab1 = tab0 + slots;
  constexpr int kRowsHere = kSRows;
  const int row_blocks = slots / (kSGroups * kSRows);
  for (int row_block = first_row_block; row_block < row_blocks; row_block += gridDim.y) {
  const int slot_begin = row_block * (kSGroups * kSRows);
  int row[kRowsHere], ps[kRowsHere], cnt[kRowsHere];
#pragma unroll
  for (int r = 0; r < kRowsHere; ++r) {
    const int sl = slot_begin + r * kSGroups + gid;
    const int entry = dealt_index(sl, slots, kSGroups * kSRows);
    const bool live = entry < m;
    row[r] = row_indices[live ? entry : 0];
    const int a0 = tab0[sl], a1 = tab1[sl];
    const bool ok = row_ok[sl] != 0;
    ps[r] = ok ? a0 : row_offsets[row[r]];
    const int len = ok ? a1 - a0 : row_offsets[row[r] + 1] - ps[r];
    cnt[r] = !live ? 0 : ok ? len : (slab == 0 ? -len : 0);
  }

  // this lane's chunks of a row, in its rotated order (see the header)
  const int rot = q + 4 * ((t * C) >> 4);
  int coff[C];    // byte offsets inside a row
#pragma unroll
  for (int c = 0; c < C; ++c) coff[c] = t * Q::kQuarter + 16 * ((c + rot) % C);

  // A row's lhs fragment is fetched ONCE by its group (lane i: piece i of the row,
  // kRowBytes / 16 bytes), kRing - 1 rows ahead; when the row's turn comes the pieces
  // go through a wave-private LDS line and come back as this lane's quarter in its
  // rotated chunk order (four quads x one quarter each would otherwise load every
  // row four times: 27 against 19 us of launch skeleton at config 3).
  constexpr int kRing = 3;
  constexpr int kPiece = kRowBytes / 16;           // bytes of a row per lane when fetched
  constexpr int kPieceWords = kPiece / 4;
  using piece_t = unsigned __attribute__((ext_vector_type(kPieceWords)));
  int wcol[kRing][kWin];
  piece_t lp[kRing];
  auto fetch = [&](int r, int slot_in_ring) {
    // the first 32 entries of the row in this slab, as PAIRS: lane (q, t) takes entries
    // 2e and 2e + 1 (e = 4t + q) -- one 8-byte load here, one 8-byte result store later
    // (half the window-load and store instructions of one 16-entry window after the other)
    // (the pair that would start at the very last entry is read one entry earlier)
    // (nonzeros >= 4 m >= 64 on this path: last - 1 is a valid index)
    const int want = ps[r] + 2 * e;
    // (uniform base + 32-bit byte offset: no 64-bit vector arithmetic per address)
    const int2 c2 = *reinterpret_cast<const int2*>(
        reinterpret_cast<const char*>(column_indices) + static_cast<unsigned>(min(want, last - 1)) * 4u);
    wcol[slot_in_ring][0] = want >= last ? c2.y : c2.x;
    wcol[slot_in_ring][1] = c2.y;
    lp[slot_in_ring] = *reinterpret_cast<const piece_t*>(
        reinterpret_cast<const char*>(lhs) +
        (static_cast<unsigned>(row[r]) * static_cast<unsigned>(ld) * static_cast<unsigned>(sizeof(T)) +
         static_cast<unsigned>(i * kPiece)));
  };
#pragma unroll
  for (int r = 0; r < kRing - 1; ++r) fetch(r, r);
  wait_vm<0>();
  if (row_block == first_row_block) __syncthreads();   // (the slab is there: once per workgroup)

  const unsigned tile_base =
      static_cast<unsigned>(reinterpret_cast<uintptr_t>(AS_LDS(reinterpret_cast<float*>(tile))));
  char* const my_line = line + (wave * 4 + g) * kRowBytes;

  static_for<kRowsHere>([&](auto R) {
    constexpr int r = decltype(R)::value;
    if constexpr (r + kRing - 1 < kRowsHere) fetch(r + kRing - 1, (r + kRing - 1) % kRing);
    const int cur_ps = ps[r];
    chunk cur_lf[C];
    *reinterpret_cast<piece_t*>(my_line + i * kPiece) = lp[r % kRing];
#pragma unroll
    for (int c = 0; c < C; ++c) cur_lf[c] = *reinterpret_cast<const chunk*>(my_line + coff[c]);

    if (cnt[r] < 0) {
      // unsorted row (rare): rhs rows gathered from global memory, any column;
      // every quad computes the same entry
      const int p1 = cur_ps - cnt[r];
      for (int p = cur_ps; p < p1; ++p) {
        const char* rrow =
            reinterpret_cast<const char*>(rhs + static_cast<int64_t>(column_indices[p]) * ld);
        v2f acc = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c)
          Dot<T>::mac(acc, cur_lf[c], *reinterpret_cast<const chunk*>(rrow + coff[c]));
        const float total = group_sum<4>(acc.x + acc.y);
        if (i == 0) {
          if constexpr (ACC) out[p] = static_cast<TO>(static_cast<float>(out[p]) + total);
          else out[p] = static_cast<TO>(total);
        }
      }
    }
    const int n_here = (debug & 1) ? 0 : max(cnt[r], 0);
    // the dot products of the entries whose (column, validity) the lanes hold: lane
    // (q, t) gets the result of ITS entry
    auto products = [&](int ecol, bool valid) -> float {
      asm volatile("" : : : "memory");   // (a window's reads stay behind the previous window's work)
      // (debug bit 4, wrong results: every entry reads slab row 0 -- what bank conflicts cost)
      const int roff = static_cast<int>(tile_base) + ((valid && !(debug & 4)) ? ((ecol - jc) * kRowBytes) : 0);
      float result = 0.f;
      // kGang steps (4 entries per group each) have their reads in flight together
      constexpr int kGang = C <= 2 ? 4 : C <= 4 ? 2 : 1;
      // lane (q, t) is valid <=> its entry exists: step s is needed iff some lane with
      // t == s is valid; a gang none of whose steps is needed is skipped (wave-uniform)
      const unsigned long long need = __builtin_amdgcn_ballot_w64(valid);
      static_for<4 / kGang>([&](auto Gc) {
        constexpr int kG = decltype(Gc)::value * kGang;
        constexpr unsigned long long kMask =
            0x1111111111111111ull * ((1ull << (kG + kGang)) - (1ull << kG));
        if ((need & kMask) == 0) return;   // (k = 128 in half, 13 entries per row and slab: 68 against 84 us)
        chunk b[kGang][C];
        static_for<kGang>([&](auto Sc) {
          constexpr int kS = kG + decltype(Sc)::value;
#pragma unroll
          for (int c = 0; c < C; ++c)
            b[kS - kG][c] = *reinterpret_cast<const __attribute__((address_space(3))) chunk*>(
                static_cast<unsigned>(quad_bcast_add<kS>(roff, coff[c])));
        });
        // the gang's dot products side by side: every multiply-add and every reduction
        // step has an independent neighbour (alone, each chain paid a hazard nop per link)
        v2f acc[kGang];
#pragma unroll
        for (int u = 0; u < kGang; ++u) acc[u] = v2f{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int u = 0; u < kGang; ++u) Dot<T>::mac(acc[u], cur_lf[c], b[u][c]);
        }
        float d[kGang];
#pragma unroll
        for (int u = 0; u < kGang; ++u) d[u] = acc[u].x + acc[u].y;
#pragma unroll
        for (int u = 0; u < kGang; ++u) d[u] += dpp_f32<kDppQuadXor1>(d[u]);
#pragma unroll
        for (int u = 0; u < kGang; ++u) d[u] += dpp_f32<kDppQuadXor2>(d[u]);
#pragma unroll
        for (int u = 0; u < kGang; ++u) result = (t == kG + u) ? d[u] : result;
      });
      return result;
    };
    auto put = [&](TO* dst, float result) {
      if constexpr (ACC) *dst = static_cast<TO>(static_cast<float>(*dst) + result);
      else *dst = static_cast<TO>(result);
    };
    // More than 32 entries of a row inside the slab (9 % of the visits at config 3,
    // i.e. a third of the wave steps): their columns are requested BEFORE the pairs are
    // worked on, each further window one ahead -- fetched on demand, every such step
    // waited a whole memory latency.
    const int longest =
        max(max(__builtin_amdgcn_readlane(n_here, 0), __builtin_amdgcn_readlane(n_here, 16)),
            max(__builtin_amdgcn_readlane(n_here, 32), __builtin_amdgcn_readlane(n_here, 48)));
    int col_next = 0;
    if (longest > 32)
      col_next = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(column_indices) +
                                               static_cast<unsigned>(min(cur_ps + 32 + e, last)) * 4u);
    // entries 0 .. 31 of the row in this slab: pairs (2e, 2e + 1)
    {
      const bool valid0 = 2 * e < n_here, valid1 = 2 * e + 1 < n_here;
      // (a half no group of the wave has entries in is skipped: wave-uniform)
      const float r0 = __builtin_amdgcn_ballot_w64(valid0) != 0 ? products(wcol[r % kRing][0], valid0) : 0.f;
      const float r1 = __builtin_amdgcn_ballot_w64(valid1) != 0 ? products(wcol[r % kRing][1], valid1) : 0.f;
      if (!(debug & 16)) {   // (bit 16, wrong results: no stores)
        TO* dst = reinterpret_cast<TO*>(reinterpret_cast<char*>(out) +
                                        static_cast<unsigned>(cur_ps + 2 * e) * static_cast<unsigned>(sizeof(TO)));
        if constexpr (!ACC && std::is_same_v<TO, float>) {
          if (valid1) *reinterpret_cast<v2f*>(dst) = v2f{r0, r1};   // 8 bytes, 4-byte aligned
          else if (valid0) *dst = r0;
        } else {
          if (valid0) put(dst, r0);
          if (valid1) put(dst + 1, r1);
        }
      }
    }
    // beyond 32 entries of one row inside the slab: 16 at a time, fetched on demand
    auto window = [&](int ecol, int w0) {
      const int left = n_here - w0;
      const bool valid = e < left;
      if (__builtin_amdgcn_ballot_w64(valid) == 0) return;
      const float result = products(ecol, valid);
      if (valid && !(debug & 16)) put(out + cur_ps + w0 + e, result);
    };
    for (int w0 = 32; w0 < longest; w0 += 16) {
      const int col = col_next;
      if (w0 + 16 < longest) col_next = column_indices[min(cur_ps + w0 + 16 + e, last)];
      window(col, w0);
    }
  });
  }   // (row blocks of this workgroup)
}

inline int slots_of(int m) { return ceil_div(m, kSGroups * kSRows) * (kSGroups * kSRows); }
constexpr int kMaxPanels = 8;
// Widest panel that divides k (0 if k is not a multiple of 64).
inline int panel_width(int k) {
  const int forced = options().sddmm_panel;   // developer knob: a width that divides k
  if (forced > 0 && k > 0 && k % forced == 0 && (forced == 64 || forced == 128 || forced == 256 || forced == 512))
    return forced;
  return k <= 0 ? 0 : k % 512 == 0 ? 512 : k % 256 == 0 ? 256 : k % 128 == 0 ? 128 : k % 64 == 0 ? 64 : 0;
}
inline int slab_rows_of_width(int w) { return (w <= 128 ? 64 * 1024 : 128 * 1024) / (w * 4); }
// Panel width of the SUMMED product (sddmm_tiled_launch_partials), whose panels
// run side by side, so a narrower panel costs no extra launch.  256 wherever it
// divides k: that width runs 16 waves per workgroup (Slab<4>: four per SIMD instead
// of two) and has the taller slab -- more entries of a mask row per slab visit;
// below about ten the 16-entry windows run mostly empty.  Measured
// (tools/sddmm_panel_bench.py, 8 replicas, widths 512 / 256 / 128):
//   512^2   x k 1024, density 0.1 (6.4 per visit at 512):   37 / 29 / 36 us
//   4096^2  x k 512,  density 0.05:                         647 / 449 / 528 us
//   2048^2  x k 512,  density 0.2 (12.8):                   246 / 229 / 250 us
//   1024^2  x k 1024, density 0.3 (19):                     168 / 160 / 180 us
//   512^2   x k 1024, density 0.5:                          66 / 66 / 79 us
// (with 8 waves per workgroup 256 was 5-10 % behind 512 at 12 and more entries
// per visit).  Up to 16 panels: the (replica, panel) pairs are one grid dimension.
inline int sum_panel_width(int m, int k, int n, int nonzeros) {
  const int w = panel_width(k);
  if (options().sddmm_panel > 0 || w <= 128 || m <= 0 || n <= 0) return w;
  if (k % 256 == 0 && k / 256 <= 16) return 256;
  const double per_visit = static_cast<double>(nonzeros) / m * slab_rows_of_width(w) / n;
  if (per_visit >= 10.0) return w;
  return k / 128 <= 16 ? 128 : w;
}
// Slab rows of the summed product's plan: 80-row slabs for 256-wide panels of LARGE masks.
// What the second workgroup of a CU hides are the prologues of the many workgroups a CU
// works through one after the other; a grid of a round or two has none to hide and pays
// for the shorter visits (a mask row brings 80 / 128 of the entries to a window).  Measured,
// 8 replicas, 128 / 80 rows (tools/sddmm_panel_bench.py under SPUTNIK_HIP_SDDMM_SLAB):
//   2048^2 x 512,  density 0.2  (config 5's weight gradient, 2048 workgroups):  226 / 215 us
//   4096^2 x 512,  density 0.05 (8192):                                         504 / 474
//   4096^2 x 512,  density 0.1, 4 replicas (4096):                              312 / 285
//   1024^2 x 1024, density 0.3  (1024):                                         156 / 154
//   512^2  x 1024, density 0.1  (256: the attention projections):                27 /  34
//   512^2  x 1024, density 0.5  (256):                                           62 /  74
// The plan does not know the batch: the rule counts the workgroups of ONE replica (256 of
// them = 2048 at a batch of 8).
inline int sum_slab_rows(int m, int k, int n, int nonzeros) {
  const int w = sum_panel_width(m, k, n, nonzeros);
  // (one panel and one replica: the call is the plain product, on the plain geometry)
  if (w != 256 || k / 256 < 2 || m <= 0 || n <= 0) return slab_rows_of_width(w);
  const int forced = options().sddmm_slab;
  if (forced == 80 || forced == 128) return forced;
  const int64_t per_replica = static_cast<int64_t>(ceil_div(n, 128)) * ceil_div(m, kSGroups * kSRows) * (k / 256);
  return per_replica >= 256 ? 80 : 128;
}
inline int rows_for(int m, int k, int n, int nonzeros, bool summed) {
  return summed ? sum_slab_rows(m, k, n, nonzeros) : slab_rows_of_width(panel_width(k));
}
inline int width_for(int m, int k, int n, int nonzeros, bool summed) {
  return summed ? sum_panel_width(m, k, n, nonzeros) : panel_width(k);
}
inline bool served(int k) { return panel_width(k) != 0 && k / panel_width(k) <= kMaxPanels; }

// The chunk table is the SpMM one with the mask's columns (n) in the role of k,
// cut at slab boundaries.  Topology only: a caller with a static mask runs it once.
template <int KV, int ROWS = default_slab_rows(KV)>
int plan(int m, int n, int slots, const int* row_indices, const int* row_offsets,
         const int* column_indices, int* table, int* row_ok, hipStream_t stream, int masks,
         int64_t mask_plan_ints) {
  using S = Slab<KV, ROWS>;
  if (masks > 1) {   // concatenated topologies: all masks' tables in one launch
    if (masks > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
    hipLaunchKernelGGL((spmm_chunk_table_masks_kernel<S::kRows>), dim3(ceil_div(slots, 4), masks),
                       dim3(256), 0, stream, m, n, slots, kSGroups * kSRows, ceil_div(n, S::kRows),
                       row_indices, row_offsets, column_indices, table, row_ok, mask_plan_ints);
    return launch_status();
  }
  hipLaunchKernelGGL((spmm_chunk_table_kernel<S::kRows>), dim3(ceil_div(slots, 4)),
                     dim3(256), 0, stream, m, n, slots, kSGroups * kSRows, ceil_div(n, S::kRows),
                     row_indices, row_offsets, column_indices, table, row_ok);
  return launch_status();
}

template <int KV>
int launch(int m, int k, int n, int nonzeros, int replicas, int slots, const int* row_indices,
           const int* row_offsets, const int* column_indices, const int* table,
           const int* row_ok, const float* lhs, int64_t lhs_stride, const float* rhs,
           int64_t rhs_stride, float* out, int64_t out_stride, int debug, hipStream_t stream,
           int mask_heads, int64_t mask_plan_ints) {
  using S = Slab<KV>;
  const int slabs = ceil_div(n, S::kRows);
  int st = 0;
  const int row_blocks = slots / (kSGroups * kSRows);
  if (row_blocks > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  // (more than one mask: their tables came from spmm_chunk_table_masks_kernel, which also
  // ranks them; the knob's word keeps its low bits)
  debug &= ~kMasksLargestFirst;
  if (mask_heads > 0 && replicas > mask_heads && replicas <= kMaxGridYZ && !(debug & 64))
    debug |= kMasksLargestFirst;
  for (int k0 = 0; k0 < k; k0 += S::kdim) {  // one launch per panel; later panels accumulate
    for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
      const int rz = min(replicas - r0, kMaxGridYZ);
      if constexpr (KV == 1) {   // quad form (debug bit 8: the 16-lanes-per-entry kernel; k = 128:
                                 // 8 chunks per lane and step ran at 128 against 79 us)
        if (!(debug & 8)) {
#define SPUTNIK_HIP_QUAD(ACC)                                                                     \
  hipLaunchKernelGGL((sddmm_quad_kernel<KV, S::kRows, float, float, ACC>),                        \
                     dim3(slabs, launch_rows_y(slabs, row_blocks, rz, debug, Quad<KV, S::kRows, float>::kBytes), rz), \
                     dim3(Quad<KV, S::kRows, float>::kThreads), 0,                                \
                     stream, m, n, nonzeros, slots, row_indices, row_offsets, column_indices,     \
                     table, row_ok, lhs + r0 * lhs_stride + k0, lhs_stride,                       \
                     rhs + r0 * rhs_stride + k0, rhs_stride, k, out + r0 * out_stride,            \
                     out_stride, 1, debug, mask_heads, mask_plan_ints, r0)
          if (k0 != 0) SPUTNIK_HIP_QUAD(true);
          else SPUTNIK_HIP_QUAD(false);
#undef SPUTNIK_HIP_QUAD
          st = launch_status();
          if (st != 0) return st;
          continue;
        }
      }
#define SPUTNIK_HIP_STAT(ACC)                                                                     \
  hipLaunchKernelGGL((sddmm_stationary_kernel<KV, ACC>),                                          \
                     dim3(slabs, launch_rows_y(slabs, row_blocks, rz, debug, S::kBytes), rz),                \
                     dim3(S::kThreads), 0, stream, m, n, nonzeros, slots, row_indices,            \
                     row_offsets, column_indices, table, row_ok, lhs + r0 * lhs_stride + k0,      \
                     lhs_stride, rhs + r0 * rhs_stride + k0, rhs_stride, k,                       \
                     out + r0 * out_stride, out_stride, 1, debug, mask_heads, mask_plan_ints, r0)
      if (k0 != 0) SPUTNIK_HIP_STAT(true);
      else SPUTNIK_HIP_STAT(false);
#undef SPUTNIK_HIP_STAT
      st = launch_status();
      if (st != 0) return st;
    }
  }
  return 0;
}

// Half operands (T = _Float16 / __bf16), output float or T.  W = width the plan was
// made for (its slab rows); the kernel's panels are at most 256 wide.
template <int KV, int ROWS, typename T, typename TO>
int launch_half(int m, int k, int n, int nonzeros, int replicas, int slots, const int* row_indices,
                const int* row_offsets, const int* column_indices, const int* table,
                const int* row_ok, const T* lhs, int64_t lhs_stride, const T* rhs,
                int64_t rhs_stride, TO* out, int64_t out_stride, int panels_z, int debug,
                hipStream_t stream, int mask_heads, int64_t mask_plan_ints) {
  using Q = Quad<KV, ROWS, T>;
  const int slabs = ceil_div(n, ROWS);
  const int row_blocks = slots / (kSGroups * kSRows);
  if (row_blocks > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (panels_z > 1) {   // every (replica, panel) pair at once, each into its own vector
    if (static_cast<int64_t>(replicas) * panels_z > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
    hipLaunchKernelGGL((sddmm_quad_kernel<KV, ROWS, T, TO, false>),
                       dim3(slabs, launch_rows_y(slabs, row_blocks, static_cast<int64_t>(replicas) * panels_z, debug, Q::kBytes),
                            replicas * panels_z),
                       dim3(Q::kThreads), 0, stream, m, n, nonzeros, slots, row_indices, row_offsets,
                       column_indices, table, row_ok, lhs, lhs_stride, rhs, rhs_stride, k, out,
                       out_stride, panels_z, debug, 0, int64_t{0}, 0);
    return launch_status();
  }
  for (int k0 = 0; k0 < k; k0 += Q::kdim) {
    for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
      const int rz = min(replicas - r0, kMaxGridYZ);
#define SPUTNIK_HIP_QUADH(ACC)                                                                    \
  hipLaunchKernelGGL((sddmm_quad_kernel<KV, ROWS, T, TO, ACC>),                                   \
                     dim3(slabs, launch_rows_y(slabs, row_blocks, rz, debug, Q::kBytes), rz),       \
                     dim3(Q::kThreads), 0, stream, m, n, nonzeros, slots, row_indices,            \
                     row_offsets, column_indices, table, row_ok, lhs + r0 * lhs_stride + k0,      \
                     lhs_stride, rhs + r0 * rhs_stride + k0, rhs_stride, k,                       \
                     out + r0 * out_stride, out_stride, 1, debug, mask_heads, mask_plan_ints, r0)
      if (k0 != 0) SPUTNIK_HIP_QUADH(true);
      else SPUTNIK_HIP_QUADH(false);
#undef SPUTNIK_HIP_QUADH
      const int st = launch_status();
      if (st != 0) return st;
    }
  }
  return 0;
}

template <typename T, typename TO>
int launch_half_width(int width, int rows /* of the plan's slabs */, int m, int k, int n, int nonzeros, int replicas, int slots,
                      const int* row_indices, const int* row_offsets, const int* column_indices,
                      const int* table, const int* row_ok, const void* lhs, int64_t lhs_stride,
                      const void* rhs, int64_t rhs_stride, void* out, int64_t out_stride,
                      int panels_z, int debug, hipStream_t stream, int mask_heads,
                      int64_t mask_plan_ints) {
#define SPUTNIK_HIP_SDH(KV, ROWS)                                                                  \
  return launch_half<KV, ROWS, T, TO>(m, k, n, nonzeros, replicas, slots, row_indices, row_offsets, \
                                      column_indices, table, row_ok, static_cast<const T*>(lhs),   \
                                      lhs_stride, static_cast<const T*>(rhs), rhs_stride,          \
                                      static_cast<TO*>(out), out_stride, panels_z, debug, stream,  \
                                      mask_heads, mask_plan_ints)
  switch (width) {
    case 64: SPUTNIK_HIP_SDH(1, 256);
    case 128: SPUTNIK_HIP_SDH(2, 128);
    case 256:
      if (rows == 80) SPUTNIK_HIP_SDH(4, 80);
      SPUTNIK_HIP_SDH(4, 128);
    case 512: SPUTNIK_HIP_SDH(4, 64);   // the plan's slabs (64 rows), panels of 256
    default: return SPUTNIK_HIP_INVALID_ARGUMENT;
  }
#undef SPUTNIK_HIP_SDH
}

// Every (replica, panel) pair in ONE launch, each writing its own [nonzeros]
// vector of `partials` (z = replica * panels + panel): for callers that sum the
// replicas anyway (the gradient of a weight shared by a batch), so that the
// panels need not run one after the other.
template <int KV, int ROWS = default_slab_rows(KV)>
int launch_partials(int m, int k, int n, int nonzeros, int replicas, int slots,
                    const int* row_indices, const int* row_offsets, const int* column_indices,
                    const int* table, const int* row_ok, const float* lhs, int64_t lhs_stride,
                    const float* rhs, int64_t rhs_stride, float* partials, int debug,
                    hipStream_t stream) {
  using S = Slab<KV, ROWS>;
  const int slabs = ceil_div(n, S::kRows);
  const int row_blocks = slots / (kSGroups * kSRows);
  const int panels = k / S::kdim;
  if (row_blocks > kMaxGridYZ || static_cast<int64_t>(replicas) * panels > kMaxGridYZ)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  hipLaunchKernelGGL((sddmm_stationary_kernel<KV, false, ROWS>),
                     dim3(slabs, launch_rows_y(slabs, row_blocks, static_cast<int64_t>(replicas) * panels, debug, S::kBytes),
                          replicas * panels),
                     dim3(S::kThreads), 0, stream, m, n, nonzeros, slots, row_indices, row_offsets,
                     column_indices, table, row_ok, lhs, lhs_stride, rhs, rhs_stride, k,
                     partials, static_cast<int64_t>(nonzeros), panels, debug, 0, int64_t{0}, 0);
  return launch_status();
}

}  // namespace

int sddmm_tiled_panel_width(int k) { return panel_width(k); }

int sddmm_tiled_panels(int m, int k, int n, int nonzeros) {
  return served(k) ? k / sum_panel_width(m, k, n, nonzeros) : 1;
}

int sddmm_tiled_launch_partials(int m, int k, int n, int nonzeros, int replicas,
                                const int* row_indices, const int* row_offsets,
                                const int* column_indices, const float* lhs, int64_t lhs_stride,
                                const float* rhs, int64_t rhs_stride, float* partials,
                                const void* workspace, hipStream_t stream) {
  const int debug = options().sddmm_debug & ~kMasksLargestFirst;
  const int slots = slots_of(m);
  const int* row_ok = static_cast<const int*>(workspace);
  const int* table =
      reinterpret_cast<const int*>(static_cast<const char*>(workspace) + row_ok_bytes(slots));
#define SPUTNIK_HIP_SD(KV)                                                                  \
  return launch_partials<KV>(m, k, n, nonzeros, replicas, slots, row_indices, row_offsets, \
                             column_indices, table, row_ok, lhs, lhs_stride, rhs,          \
                             rhs_stride, partials, debug, stream)
  switch (sum_panel_width(m, k, n, nonzeros)) {
    case 64: SPUTNIK_HIP_SD(1);
    case 128: SPUTNIK_HIP_SD(2);
    case 256:
      if (sum_slab_rows(m, k, n, nonzeros) == 80)
        return launch_partials<4, 80>(m, k, n, nonzeros, replicas, slots, row_indices, row_offsets,
                                      column_indices, table, row_ok, lhs, lhs_stride, rhs,
                                      rhs_stride, partials, debug, stream);
      SPUTNIK_HIP_SD(4);
    case 512: SPUTNIK_HIP_SD(8);
    default: return SPUTNIK_HIP_INVALID_ARGUMENT;
  }
#undef SPUTNIK_HIP_SD
}

bool sddmm_tiled_applicable(int m, int k, int n, int nonzeros, const float* lhs,
                            int64_t lhs_stride, const float* rhs, int64_t rhs_stride) {
  return served(k) && n >= 16 && m >= 16 && nonzeros >= 4 * static_cast<int64_t>(m) &&
         static_cast<int64_t>(n) * k * 4 < (int64_t{1} << 32) &&
         static_cast<int64_t>(m) * k * 4 < (int64_t{1} << 32) && nonzeros < (1 << 29) &&
         aligned_to(lhs, 16) && aligned_to(rhs, 16) && lhs_stride % 4 == 0 && rhs_stride % 4 == 0;
}

// `summed`: for sddmm_tiled_launch_partials (its panel width, hence its slabs and
// its chunk table, can differ from the plain product's).
namespace {
// Tables of the rhs-stationary kernels alone.
size_t table_bytes(int m, int k, int n, int nonzeros, bool summed) {
  if (!served(k) || n < 16 || m < 16 || nonzeros < 4 * static_cast<int64_t>(m)) return 0;
  const int rows = rows_for(m, k, n, nonzeros, summed);
  return row_ok_bytes(slots_of(m)) +
         sizeof(int) * static_cast<size_t>(ceil_div(n, rows) + 1) * slots_of(m);
}
// The plain (not summed) product of this shape can take the pair-flat kernel in SOME
// storage type (the workspace does not know the operands' type): its plan sits behind
// the tables.
bool flat_shape(int m, int k, int n, int nonzeros) {
  return sddmm_flat_applicable(m, k, n, nonzeros, 4) || sddmm_flat_applicable(m, k, n, nonzeros, 2);
}
size_t flat_plan_offset(int m, int k, int n, int nonzeros) {
  return (table_bytes(m, k, n, nonzeros, false) + 255) / 256 * 256;
}
}  // namespace

size_t sddmm_tiled_workspace_bytes(int m, int k, int n, int nonzeros, bool summed) {
  const size_t tables = table_bytes(m, k, n, nonzeros, summed);
  if (tables == 0 || summed || !flat_shape(m, k, n, nonzeros)) return tables;
  return flat_plan_offset(m, k, n, nonzeros) + sddmm_flat_plan_bytes(m, n, nonzeros);
}

// with_flat: also the pair-flat kernel's plan (sddmm_flat.hip) -- for plans that are KEPT
// (sputnik_hip_sddmm_plan: a static mask); a call that plans for itself and throws the
// plan away takes the tables only (one 5 us launch against a memset and four launches).
int sddmm_tiled_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                     const int* row_offsets, const int* column_indices, void* workspace,
                     hipStream_t stream, bool summed, bool with_flat, int masks,
                     int64_t mask_plan_ints) {
  if (with_flat && !summed && flat_shape(m, k, n, nonzeros)) {
    const int st = sddmm_flat_plan(m, n, nonzeros, row_indices, row_offsets, column_indices,
                                   static_cast<char*>(workspace) + flat_plan_offset(m, k, n, nonzeros),
                                   stream);
    if (st != 0) return st;
  }
  const int slots = slots_of(m);
  int* row_ok = static_cast<int*>(workspace);
  int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + row_ok_bytes(slots));
  switch (width_for(m, k, n, nonzeros, summed)) {
    case 64: return plan<1>(m, n, slots, row_indices, row_offsets, column_indices, table, row_ok, stream, masks, mask_plan_ints);
    case 128: return plan<2>(m, n, slots, row_indices, row_offsets, column_indices, table, row_ok, stream, masks, mask_plan_ints);
    case 256:
      if (rows_for(m, k, n, nonzeros, summed) == 80)
        return plan<4, 80>(m, n, slots, row_indices, row_offsets, column_indices, table, row_ok, stream, masks, mask_plan_ints);
      return plan<4>(m, n, slots, row_indices, row_offsets, column_indices, table, row_ok, stream, masks, mask_plan_ints);
    case 512: return plan<8>(m, n, slots, row_indices, row_offsets, column_indices, table, row_ok, stream, masks, mask_plan_ints);
    default: return SPUTNIK_HIP_INVALID_ARGUMENT;
  }
}

// Kernels only, on a planned workspace.
int sddmm_tiled_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                       const int* row_offsets, const int* column_indices, const float* lhs,
                       int64_t lhs_stride, const float* rhs, int64_t rhs_stride, float* out,
                       int64_t out_stride, const void* workspace, hipStream_t stream,
                       int mask_heads, int64_t mask_plan_ints, bool flat) {
  if (flat && mask_heads == 0 && sddmm_flat_applicable(m, k, n, nonzeros, 4))
    return sddmm_flat_launch(m, k, n, nonzeros, replicas, row_indices, lhs, lhs_stride, rhs,
                             rhs_stride, out, out_stride, SPUTNIK_HIP_F32, SPUTNIK_HIP_F32,
                             static_cast<const char*>(workspace) + flat_plan_offset(m, k, n, nonzeros),
                             stream);
  const int debug = options().sddmm_debug & ~kMasksLargestFirst;  // timing experiments only
  const int slots = slots_of(m);
  const int* row_ok = static_cast<const int*>(workspace);
  const int* table =
      reinterpret_cast<const int*>(static_cast<const char*>(workspace) + row_ok_bytes(slots));
#define SPUTNIK_HIP_SD(KV)                                                               \
  return launch<KV>(m, k, n, nonzeros, replicas, slots, row_indices, row_offsets,        \
                    column_indices, table, row_ok, lhs, lhs_stride, rhs, rhs_stride, out, \
                    out_stride, debug, stream, mask_heads, mask_plan_ints)
  switch (panel_width(k)) {
    case 64: SPUTNIK_HIP_SD(1);
    case 128: SPUTNIK_HIP_SD(2);
    case 256: SPUTNIK_HIP_SD(4);
    case 512: SPUTNIK_HIP_SD(8);
    default: return SPUTNIK_HIP_INVALID_ARGUMENT;
  }
#undef SPUTNIK_HIP_SD
}


// Native half operands (in_type SPUTNIK_HIP_F16 / BF16; out_type float or in_type).
bool sddmm_tiled_applicable_half(int m, int k, int n, int nonzeros, const void* lhs,
                                 int64_t lhs_stride, const void* rhs, int64_t rhs_stride) {
  return served(k) && n >= 16 && m >= 16 && nonzeros >= 4 * static_cast<int64_t>(m) &&
         static_cast<int64_t>(n) * k * 2 < (int64_t{1} << 32) &&
         static_cast<int64_t>(m) * k * 2 < (int64_t{1} << 32) && nonzeros < (1 << 29) &&
         aligned_to(lhs, 16) && aligned_to(rhs, 16) && lhs_stride % 8 == 0 && rhs_stride % 8 == 0;
}
// Kernel launches the plain product makes for one replica (more than one: later
// passes add into the output, which a half OUTPUT would round every time).
int sddmm_tiled_passes_half(int k) { return served(k) ? k / min(panel_width(k), 256) : 1; }

int sddmm_tiled_launch_half(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                            const int* row_offsets, const int* column_indices, const void* lhs,
                            int64_t lhs_stride, const void* rhs, int64_t rhs_stride, void* out,
                            int64_t out_stride, int in_type, int out_type, const void* workspace,
                            hipStream_t stream, int mask_heads, int64_t mask_plan_ints, bool flat) {
  if (flat && mask_heads == 0 && sddmm_flat_applicable(m, k, n, nonzeros, 2))
    return sddmm_flat_launch(m, k, n, nonzeros, replicas, row_indices, lhs, lhs_stride, rhs,
                             rhs_stride, out, out_stride, in_type, out_type,
                             static_cast<const char*>(workspace) + flat_plan_offset(m, k, n, nonzeros),
                             stream);
  const int debug = options().sddmm_debug & ~kMasksLargestFirst;
  const int slots = slots_of(m);
  const int* row_ok = static_cast<const int*>(workspace);
  const int* table =
      reinterpret_cast<const int*>(static_cast<const char*>(workspace) + row_ok_bytes(slots));
  const int width = panel_width(k);
#define SPUTNIK_HIP_SDT(T, TO)                                                                    \
  return launch_half_width<T, TO>(width, slab_rows_of_width(width), m, k, n, nonzeros, replicas, slots, row_indices, \
                                  row_offsets, column_indices, table, row_ok, lhs, lhs_stride,    \
                                  rhs, rhs_stride, out, out_stride, 1, debug, stream, mask_heads, \
                                  mask_plan_ints)
  if (in_type == SPUTNIK_HIP_F16 && out_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_SDT(_Float16, float);
  if (in_type == SPUTNIK_HIP_F16 && out_type == SPUTNIK_HIP_F16) SPUTNIK_HIP_SDT(_Float16, _Float16);
  if (in_type == SPUTNIK_HIP_BF16 && out_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_SDT(__bf16, float);
  if (in_type == SPUTNIK_HIP_BF16 && out_type == SPUTNIK_HIP_BF16) SPUTNIK_HIP_SDT(__bf16, __bf16);
#undef SPUTNIK_HIP_SDT
  return SPUTNIK_HIP_INVALID_ARGUMENT;
}

// Summed form on half operands: float partial vectors, one per (replica, panel).
// Served when the summed plan's panel width is at most 256 (always, up to k = 4096).
bool sddmm_tiled_sum_half_served(int m, int k, int n, int nonzeros) {
  return served(k) && sum_panel_width(m, k, n, nonzeros) <= 256;
}
int sddmm_tiled_launch_partials_half(int m, int k, int n, int nonzeros, int replicas,
                                     const int* row_indices, const int* row_offsets,
                                     const int* column_indices, const void* lhs,
                                     int64_t lhs_stride, const void* rhs, int64_t rhs_stride,
                                     int in_type, float* partials, const void* workspace,
                                     hipStream_t stream) {
  const int debug = options().sddmm_debug & ~kMasksLargestFirst;
  const int slots = slots_of(m);
  const int* row_ok = static_cast<const int*>(workspace);
  const int* table =
      reinterpret_cast<const int*>(static_cast<const char*>(workspace) + row_ok_bytes(slots));
  const int width = sum_panel_width(m, k, n, nonzeros);
  if (width > 256) return SPUTNIK_HIP_UNSUPPORTED;
  const int panels = k / width;
  if (in_type == SPUTNIK_HIP_F16)
    return launch_half_width<_Float16, float>(width, sum_slab_rows(m, k, n, nonzeros), m, k, n, nonzeros, replicas, slots, row_indices,
                                              row_offsets, column_indices, table, row_ok, lhs,
                                              lhs_stride, rhs, rhs_stride, partials,
                                              static_cast<int64_t>(nonzeros), panels, debug, stream,
                                              0, int64_t{0});
  if (in_type == SPUTNIK_HIP_BF16)
    return launch_half_width<__bf16, float>(width, sum_slab_rows(m, k, n, nonzeros), m, k, n, nonzeros, replicas, slots, row_indices,
                                            row_offsets, column_indices, table, row_ok, lhs,
                                            lhs_stride, rhs, rhs_stride, partials,
                                            static_cast<int64_t>(nonzeros), panels, debug, stream,
                                            0, int64_t{0});
  return SPUTNIK_HIP_INVALID_ARGUMENT;
}

}  // namespace sputnik_hip
