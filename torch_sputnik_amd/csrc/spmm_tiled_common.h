// Pieces shared by the LDS-tiled SpMM kernels (spmm_tiled.hip: 256-column
// tiles, spmm_tiled64.hip: 64-column tiles): the chunk-table pre-pass, the
// direct global->LDS copy, hand-counted waits and the DPP broadcast helpers.
#pragma once

#include <type_traits>
#include <utility>

#include "common.h"
#include "wave_utils.h"

namespace sputnik_hip {
namespace tiled {

// Workspace layout: [row_ok: one int per row slot][chunk table].
inline size_t row_ok_bytes(int slots) { return (sizeof(int) * static_cast<size_t>(slots) + 255) / 256 * 256; }

#define AS_GLOBAL(p) ((__attribute__((address_space(1))) void*)(p))
#define AS_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------
// Pre-pass: chunk table + per-row order check.  One wave per row slot.
// table[c * slots + slot], c in [0, nchunks]: index of the first nonzero of
// the row in slot `slot` (see dealt_index) whose column is >= c*BK (row end if
// none).  Padding slots get 0 everywhere, i.e. empty rows.
// ---------------------------------------------------------------------------
// Row slots are DEALT to workgroups: slot s holds entry
// (s % per) * (slots / per) + s / per of `row_indices`, so that every run of
// `per` consecutive slots (a workgroup's rows) takes every (slots/per)-th entry.
// Callers pass row_indices sorted by row length (modules/spmm.py:4-6,
// tests/sparse_matrix.py:22); contiguous blocks of that order would give the
// first workgroup the longest rows and the last one the shortest (+-10 % of
// work at 4096^2, density 0.1), and with about one workgroup per CU the launch
// lasts as long as the slowest (measured 427 -> 408 us).  Entries >= m are padding.
// Launch position -> work index such that CONSECUTIVE work indices run on the
// same XCD.  Workgroups are dealt to the 8 XCDs round-robin in launch order
// (x fastest, then y, then z), and each XCD has its own L2: a kernel that
// decodes its (tile, block, replica) from this index with the operand-sharing
// dimension fastest has all workgroups that read the same bytes behind ONE L2
// (otherwise every XCD fetches them from memory for itself).
// (64-bit: grid.x * grid.y * grid.z may exceed 2^31 for a huge batch)
__device__ __forceinline__ unsigned long long xcd_local_index() {
  using u64 = unsigned long long;
  const u64 total = static_cast<u64>(gridDim.x) * gridDim.y * gridDim.z;
  const u64 launch = (static_cast<u64>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  return total % 8 == 0 ? (launch % 8) * (total / 8) + launch / 8 : launch;
}

// "Many mask" launches (common.h, select_mask: replica r works under mask r / heads).  The
// masks of a batch differ in size -- tests/test_attention_many_masks.py:26-36 of the
// reference draws a sparsity per batch element -- and consecutive work indices, i.e. one
// XCD, would be the replicas of ONE mask: the XCD that draws the densest mask then works on
// alone (b = 8 x 8 heads with densities 0.1 / 0.2 / 0.05 / 0.5: the 0.5 XCDs carry 2.4
// times the mean).  Here XCD x takes the replicas r with r % 8 == x, whole replicas as
// before (their workgroups share operands: one L2), so the heads of every mask are dealt
// over all eight.  `per_replica` = workgroups of one replica, `replicas` the launch's;
// falls back to xcd_local_index when the replicas do not divide by 8.
__device__ __forceinline__ unsigned long long xcd_spread_replicas_index(unsigned per_replica,
                                                                        unsigned replicas) {
  using u64 = unsigned long long;
  const u64 v = xcd_local_index();
  if (replicas % 8 != 0) return v;
  const u64 per_xcd = static_cast<u64>(per_replica) * (replicas / 8);   // (the total divides by 8)
  const u64 xcd = v / per_xcd, local = v - xcd * per_xcd;
  const u64 nth = local / per_replica, inner = local - nth * per_replica;
  return (nth * 8 + xcd) * per_replica + inner;
}

// 32-bit form for the dense-output kernels, whose workgroup count is bounded by
// memory: every workgroup owns at least 64 x 64 output elements (16 KiB), so 2^31
// workgroups would be 32 TiB of output.  (Keeps 64-bit division out of the
// register-tight 128 x 512 kernel.)
__device__ __forceinline__ int xcd_local_index32() {
  const int total = gridDim.x * gridDim.y * gridDim.z;
  const int launch = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  return total % 8 == 0 ? (launch % 8) * (total / 8) + launch / 8 : launch;
}

__device__ __forceinline__ int dealt_index(int slot, int slots, int per) {
  return (slot % per) * (slots / per) + slot / per;
}

template <int BK>  // rows of B per chunk (any positive number; a power of two divides by shifting)
__device__ __forceinline__ void chunk_table_body(
    int m, int k, int slots, int per, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ table, int* __restrict__ row_ok) {
  const int lane = threadIdx.x % kWave;
  const int slot = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
  if (slot >= slots) return;
  const int entry = dealt_index(slot, slots, per);
  if (entry >= m) {
    for (int c = lane; c <= nchunks; c += kWave) table[static_cast<int64_t>(c) * slots + slot] = 0;
    if (lane == 0) row_ok[slot] = 1;
    return;
  }
  const int row = row_indices[entry];
  const int p0 = row_offsets[row];
  const int p1 = row_offsets[row + 1];
  bool ok = true;
  // (round 5: a row of w windows was w + 1 dependent round trips -- every window of 64
  // columns fetched when the one before it had been worked on, then the last column once
  // more; now the next window is in flight while this one is worked on and the last column
  // is taken from the window that holds it: 6.9 -> 6.0 us at 4096 rows of 410 entries,
  // tools/prepass_probe.py under rocprofv3 -- what is left is the launch and the three hops
  // row id -> bounds -> first window)
  int cur = 0, prev = -1;
  if (p0 + lane < p1) {
    cur = column_indices[p0 + lane];
    prev = lane > 0 ? column_indices[p0 + lane - 1] : -1;
  }
  int last_column = -1;
  for (int base = p0; base < p1; base += kWave) {
    const int p = base + lane;
    int cur_next = 0, prev_next = -1;
    if (p + kWave < p1) {
      cur_next = column_indices[p + kWave];
      prev_next = column_indices[p + kWave - 1];
    }
    if (p < p1) {
      if (cur <= prev || cur >= k) {
        ok = false;
      } else {
        const int cb = static_cast<int>(static_cast<unsigned>(cur) / BK);
        const int pb = (prev < 0) ? -1 : static_cast<int>(static_cast<unsigned>(prev) / BK);
        for (int c = pb + 1; c <= cb; ++c) table[static_cast<int64_t>(c) * slots + slot] = p;
      }
    }
    if (base + kWave >= p1)   // (wave-uniform: the row's last window; its last entry's lane)
      last_column = __builtin_amdgcn_readlane(cur, p1 - 1 - base);
    cur = cur_next;
    prev = prev_next;
  }
  int last = -1;
  if (p1 > p0) {
    const int lc = last_column;
    last = (lc >= 0 && lc < k) ? static_cast<int>(static_cast<unsigned>(lc) / BK) : nchunks;
  }
  for (int c = last + 1 + lane; c <= nchunks; c += kWave)
    table[static_cast<int64_t>(c) * slots + slot] = p1;
  // 1 = this row's columns ascend and are in range (the tiled kernels rely on
  // it); every slot writes its own word, so the array needs no initialisation.
  const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
  if (lane == 0) row_ok[slot] = wave_ok ? 1 : 0;
}

template <int BK>
__global__ __launch_bounds__(256) void spmm_chunk_table_kernel(
    int m, int k, int slots, int per, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ table, int* __restrict__ row_ok) {
  chunk_table_body<BK>(m, k, slots, per, nchunks, row_indices, row_offsets, column_indices, table,
                       row_ok);
}

// Where a many-mask plan keeps which mask starts nth (in ints from the workspace's start).
// A launch works through its masks largest first: the workgroups of the densest mask take
// several times the mean, and started last they run on alone while the chip empties.
__host__ __device__ __forceinline__ int64_t mask_start_word(int nth, int64_t mask_plan_ints) {
  return (static_cast<int64_t>(nth) + 1) * mask_plan_ints - 1;
}

// The same for the concatenated topologies of a "many mask" batch (common.h,
// select_mask) in ONE launch: blockIdx.y = mask, every mask's table and order words
// `mask_plan_ints` ints behind the previous mask's (round 4: one pre-pass launch per mask
// was 8 launches in front of the many-mask SDDMM).
template <int BK>
__global__ __launch_bounds__(256) void spmm_chunk_table_masks_kernel(
    int m, int k, int slots, int per, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ table, int* __restrict__ row_ok, int64_t mask_plan_ints) {
  const int mask = blockIdx.y;
  int first = 0;   // entries of the masks before this one
  for (int j = 0; j < mask; ++j) first += row_offsets[static_cast<int64_t>(j) * (m + 1) + m];
  // Start order of the masks, largest first (mask_start_word): this mask's rank among the
  // entry counts goes to the LAST word of region `rank` (the regions end in a spare word).
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int mine = row_offsets[static_cast<int64_t>(mask) * (m + 1) + m];
    int rank = 0;
    for (int j = 0; j < static_cast<int>(gridDim.y); ++j) {
      const int other = row_offsets[static_cast<int64_t>(j) * (m + 1) + m];
      rank += (other > mine || (other == mine && j < mask)) ? 1 : 0;
    }
    row_ok[mask_start_word(rank, mask_plan_ints)] = mask;
  }
  chunk_table_body<BK>(m, k, slots, per, nchunks, row_indices + static_cast<int64_t>(mask) * m,
                       row_offsets + static_cast<int64_t>(mask) * (m + 1), column_indices + first,
                       table + mask * mask_plan_ints, row_ok + mask * mask_plan_ints);
}

// Direct global->LDS copy of one 1 KiB row segment (64 lanes x 16 B):
// LDS destination = M0 + lane*16, global source = per-lane address.
// Written as inline asm on purpose: for the builtin form hipcc treats the
// copy as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of every
// following ds_read, which would serialise the prefetch of the next stage
// with the compute on the current one.  The waits are placed by hand instead
// (wait_vm<N>() before the barrier that publishes a stage).
__device__ __forceinline__ void lds_dma_row(const float* row_base /* wave-uniform */,
                                            unsigned lane_byte_offset, const float* lds_dst) {
  const unsigned lds_addr =
      static_cast<unsigned>(reinterpret_cast<uintptr_t>(AS_LDS(const_cast<float*>(lds_dst))));
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2"
      :
      : "s"(lds_addr), "v"(lane_byte_offset), "s"(row_base)
      : "memory", "m0");
}


// Vector load whose completion the compiler does not track (the main loop counts
// vmcnt by hand, see the kernel).
// RULE: the compiler takes the load as complete where it is issued.  Every
// register such a load writes must therefore be READ (tie_reg, or a "+v"
// operand of the wait) after the s_waitcnt that covers it, on EVERY path --
// also where the value is not wanted (the clamped requests past the last
// chunk).  A register that is never read again is free to the compiler at once:
// it may be given to something else while the load is still in flight, and the
// late data then lands on that something else (it was the epilogue's output
// address once: DESIGN.md section 3.1).
// (wave-uniform base in SGPRs + 32-bit per-lane byte offset: no 64-bit VGPR
// address arithmetic, no VGPR pairs to keep alive.)
__device__ __forceinline__ int untracked_load_i32(const int* base, unsigned byte_offset) {
  int v;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_offset), "s"(base) : "memory");
  return v;
}
__device__ __forceinline__ float untracked_load_f32(const float* base, unsigned byte_offset) {
  float v;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_offset), "s"(base) : "memory");
  return v;
}
// Wait until at most N vector-memory operations of this wave are in flight;
// the operands tie later uses of the loaded registers to the wait.
template <int N>
__device__ __forceinline__ void wait_vm(int& a, float& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// Marks a register as written "here": placed after a hand-counted wait it keeps
// the compiler from using an asm-loaded register before that wait.
__device__ __forceinline__ void tie_reg(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void tie_reg(float& v) { asm volatile("" : "+v"(v)); }

// ROWS consecutive table entries (stream positions of a wave's rows at one
// chunk boundary) straight into SGPRs: one scalar load instead of a per-lane
// vector load plus a v_readlane per use.  Untracked like the vector loads
// above: `wait_positions` must come before the first use.  (The scalar cache
// shares `lgkmcnt` with LDS; one extra operation in flight only makes the
// compiler's own counted LDS waits stricter, never looser.)
template <int ROWS>
struct Positions;
template <>
struct Positions<16> {
  using type = int __attribute__((ext_vector_type(16)));
  static __device__ __forceinline__ type load(const int* p /* wave-uniform */) {
    type v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
    return v;
  }
};
template <>
struct Positions<8> {
  using type = int __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ type load(const int* p /* wave-uniform */) {
    type v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
    return v;
  }
};
template <>
struct Positions<4> {
  using type = int __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ type load(const int* p /* wave-uniform */) {
    type v;
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
    return v;
  }
};
template <typename V>
__device__ __forceinline__ void wait_positions(V& a) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a) : : "memory");
}

// One nonzero against the staged tile.  A lane owns 4 consecutive columns of
// every 256-column piece of the tile row (VEC = 4: one piece, VEC = 8: two, the
// second 1 KiB further on), so a B row strip is one or two ds_read_b128.
#define SPUTNIK_HIP_FMA4(ACC, A, B)          \
  do {                                       \
    (ACC)[0] = fmaf((A), (B).x, (ACC)[0]);   \
    (ACC)[1] = fmaf((A), (B).y, (ACC)[1]);   \
    (ACC)[2] = fmaf((A), (B).z, (ACC)[2]);   \
    (ACC)[3] = fmaf((A), (B).w, (ACC)[3]);   \
  } while (0)
using v4f = float __attribute__((ext_vector_type(4)));
template <int VEC>
struct BStrip;
template <>
struct BStrip<4> {
  v4f p0;
  __device__ __forceinline__ void read(const char* __restrict__ at) {
    p0 = *reinterpret_cast<const v4f*>(at);
  }
  __device__ __forceinline__ void fma(float (&acc)[4], float a) { SPUTNIK_HIP_FMA4(acc, a, p0); }
};
template <>
struct BStrip<8> {
  v4f p0, p1;
  __device__ __forceinline__ void read(const char* __restrict__ at) {
    p0 = *reinterpret_cast<const v4f*>(at);
    p1 = *reinterpret_cast<const v4f*>(at + 1024);
  }
  __device__ __forceinline__ void fma(float (&acc)[8], float a) {
    // one `s_waitcnt` per entry instead of one per piece: the loop around this is
    // bound by scalar/issue slots, not by the LDS pipe
    asm("" : "+v"(p0), "+v"(p1));
    SPUTNIK_HIP_FMA4(acc, a, p0);
    SPUTNIK_HIP_FMA4(acc + 4, a, p1);
  }
};

// Compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>): a loop
// whose body indexes register arrays by the counter must not be left to
// `#pragma unroll` (a refused unroll turns the arrays into scratch memory).
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// DPP row_newbcast: every lane of a 16-lane row reads lane U of ITS row.  With
// the same 16 entries replicated in all four rows this is a wave-wide
// broadcast of entry U that costs one VALU op and no SGPR round trip.
template <int U>
__device__ __forceinline__ int row_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xF, 0xF, true);
}

// (offset, value) of an entry travel as ONE 64-bit register pair: a 64-bit DPP
// move broadcasts both halves for the price of a 32-bit one (measured:
// tools/ubench.hip, mov_dpp64 vs mov_dpp32), which saves one DPP operation per
// nonzero against broadcasting the two words separately.  (64-bit DPP only
// knows row_newbcast, so the rotation stays two 32-bit moves.)
using entry_pair = unsigned long long;
__device__ __forceinline__ entry_pair make_entry(int roff, float rval) {
  return static_cast<unsigned int>(roff) |
         (static_cast<entry_pair>(__builtin_bit_cast(unsigned int, rval)) << 32);
}
template <int U>
__device__ __forceinline__ entry_pair row_bcast_entry(entry_pair e) {
  return __builtin_amdgcn_update_dpp(entry_pair{0}, e, 0x150 + U, 0xF, 0xF, true);
}
__device__ __forceinline__ int entry_off(entry_pair e) { return static_cast<int>(e & 0xffffffffu); }
__device__ __forceinline__ float entry_val(entry_pair e) {
  return __builtin_bit_cast(float, static_cast<unsigned int>(e >> 32));
}
// Four nonzeros G..G+3 of the replicated 16-entry set (roff = byte offset of
// the B row inside the staged tile, rval = value; both per entry lane).
template <int G, int VEC>
__device__ __forceinline__ void dpp_group4(float (&acc)[VEC], int roff, float rval,
                                           const char* __restrict__ lane_base) {
  const entry_pair e = make_entry(roff, rval);
  const entry_pair e0 = row_bcast_entry<G + 0>(e), e1 = row_bcast_entry<G + 1>(e);
  const entry_pair e2 = row_bcast_entry<G + 2>(e), e3 = row_bcast_entry<G + 3>(e);
  BStrip<VEC> b0, b1, b2, b3;
  b0.read(lane_base + entry_off(e0));
  b1.read(lane_base + entry_off(e1));
  b2.read(lane_base + entry_off(e2));
  b3.read(lane_base + entry_off(e3));
  b0.fma(acc, entry_val(e0));
  b1.fma(acc, entry_val(e1));
  b2.fma(acc, entry_val(e2));
  b3.fma(acc, entry_val(e3));
}

// The same answer without LDS and without a barrier: every wave looks at all of
// the workgroup's row slots itself (a few hundred words from L2) and reduces with
// a ballot.  For the tile that needs all 160 KiB of the CU's LDS for B.
__device__ __forceinline__ bool block_rows_ok_wave(const int* __restrict__ row_ok, int block_slot0,
                                                   int rows) {
  int ok = 1;
  for (int s = threadIdx.x % kWave; s < rows; s += kWave) ok &= row_ok[block_slot0 + s];
  return __builtin_amdgcn_ballot_w64(ok == 0) == 0;
}

// True iff every one of the workgroup's `rows` row slots (starting at
// block_slot0) passed the pre-pass check.  Contains a workgroup barrier.
__device__ __forceinline__ bool block_rows_ok(const int* __restrict__ row_ok, int block_slot0,
                                              int rows) {
  int ok = 1;
  for (int s = threadIdx.x; s < rows; s += blockDim.x) ok &= row_ok[block_slot0 + s];
  return __syncthreads_and(ok) != 0;
}

// Order-independent path for ONE row strip of 4 columns per lane: walks the
// row's nonzeros in storage order and gathers B from global memory.  Used by
// the tiled kernels for row blocks whose columns do not ascend.
__device__ __forceinline__ float4 gather_row_strip(const float* __restrict__ values,
                                                   const int* __restrict__ column_indices,
                                                   int p0, int p1,
                                                   const float* __restrict__ dense_col, int n) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = p0; p < p1; ++p) {
    const float a = values[p];
    const float4 b =
        *reinterpret_cast<const float4*>(dense_col + static_cast<int64_t>(column_indices[p]) * n);
    acc.x = fmaf(a, b.x, acc.x);
    acc.y = fmaf(a, b.y, acc.y);
    acc.z = fmaf(a, b.z, acc.z);
    acc.w = fmaf(a, b.w, acc.w);
  }
  return acc;
}

// n16 (1..16) entries of a replicated 16-entry set, four at a time; the last
// group may hold up to three padded entries (zero value, tile row 0: the
// caller masks them).  Exact tails (pair/single groups at every position) were
// tried: the extra branches and the 2-3x larger unrolled code of the 16 rows
// cost more than the padded work they save (0.58 vs 0.50 ms at density 0.1).
template <int VEC>
__device__ __forceinline__ void dpp_entries(float (&acc)[VEC], int n16, int roff, float rval,
                                            const char* __restrict__ lane_base) {
  if (n16 > 0) dpp_group4<0>(acc, roff, rval, lane_base);
  if (n16 > 4) dpp_group4<4>(acc, roff, rval, lane_base);
  if (n16 > 8) dpp_group4<8>(acc, roff, rval, lane_base);
  if (n16 > 12) dpp_group4<12>(acc, roff, rval, lane_base);
}

// Exact variant for short segments: always work on entries 0..3 of the
// replicated set and rotate the set by four lanes (DPP row_ror) after each
// group, so every position is the static position 0: one group body instead of
// four, and the last 1-3 entries are processed exactly (no padded entry), at
// the price of two extra DPP moves per group of four.
// lane i of every 16-lane row <- lane (i + N) mod 16.  (DPP row_ror:R moves
// data towards HIGHER lanes, lane i <- lane i - R, so this is row_ror:16-N.)
template <int N>
__device__ __forceinline__ int row_rotate_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + (16 - N), 0xF, 0xF, true);
}
template <int N>
__device__ __forceinline__ float row_rotate_f(float v) {
  return __builtin_bit_cast(float, row_rotate_i<N>(__builtin_bit_cast(int, v)));
}
template <int N>
__device__ __forceinline__ entry_pair row_rotate_entry(entry_pair e) {
  return make_entry(row_rotate_i<N>(entry_off(e)), row_rotate_f<N>(entry_val(e)));
}

template <int COUNT, int VEC>
__device__ __forceinline__ void dpp_group_at0(float (&acc)[VEC], entry_pair e,
                                              const char* __restrict__ lane_base) {
  static_assert(COUNT == 1 || COUNT == 2 || COUNT == 4 || COUNT == 8, "");
  entry_pair b[COUNT];
  BStrip<VEC> strip[COUNT];
  static_for<COUNT>([&](auto i) { b[i] = row_bcast_entry<decltype(i)::value>(e); });
  static_for<COUNT>([&](auto i) { strip[i].read(lane_base + entry_off(b[i])); });
  static_for<COUNT>([&](auto i) { strip[i].fma(acc, entry_val(b[i])); });
}
template <int VEC>
__device__ __forceinline__ void dpp_entries_exact(float (&acc)[VEC], int n16, int roff, float rval,
                                                  const char* __restrict__ lane_base) {
  entry_pair e = make_entry(roff, rval);
  int left = n16;
  for (; left >= 4; left -= 4) {
    dpp_group_at0<4>(acc, e, lane_base);
    e = row_rotate_entry<4>(e);
  }
  if (left & 2) {
    dpp_group_at0<2>(acc, e, lane_base);
    e = row_rotate_entry<2>(e);
  }
  if (left & 1) dpp_group_at0<1>(acc, e, lane_base);
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

}  // namespace tiled
}  // namespace sputnik_hip
