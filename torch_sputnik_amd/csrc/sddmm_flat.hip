// Pair-flat SDDMM for attention-sized inner dimensions (rows of 128 or 256 bytes:
// k = 64 in float32, k = 64 / 128 in float16 / bfloat16), round 4:
//
//   out[p] = < lhs[i_p, 0:k], rhs[j_p, 0:k] >   for every stored (i_p, j_p)
//
// The quad kernel of sddmm_tiled.hip keeps one rhs slab in LDS and walks mask ROWS:
// a 16-lane group owns a row, the row's lhs fragment lives in registers, and every
// (row, slab) visit -- 25 entries at config 3 -- pays its own bookkeeping: window
// requests, bounds, masks, the lhs row's trip through an LDS line, 8 steps of 16 for
// 6.4 steps of work.  rocprofv3 counts 35 vector instructions per 16 entries where the
// arithmetic step has 17; 1024 workgroups stage 64 KiB each
// before they compute anything.
//
// Here NOTHING is per row.  An output entry needs one lhs row and one rhs row, and the
// output has no accumulator that outlives the entry, so the order of the entries is
// free.  A workgroup owns a block of lhs rows, RESIDENT in LDS, and walks the rhs operand
// in slabs through a double-buffered LDS stage.  For every (row block, slab) TILE a plan
// made from the topology alone holds a flat list of descriptors; a step of a wave is 16
// descriptors, one per QUAD:
//
//   descriptor (8 bytes) = a PAIR of CSR-adjacent entries (p, p + 1) of one mask row
//   whose columns both fall into the slab: lhs row in the block (8 bits), the two rhs
//   rows in the LDS stages (8 bits each: the stage rides in the top bit), a "single" flag
//   (the odd entry of a run), and p.
//
// Lane (quad q, t) reads the t-th quarter of the lhs row ONCE and of both rhs rows
// from LDS (3 C ds_read_b128 for two entries; C = 16-byte chunks of a row quarter),
// two quad_perm DPP adds close each dot product, and the pair leaves as one 8-byte
// store.  Per 32 entries of a wave: 3 C address adds (the quad broadcast of the row
// offset rides on them), 3 C LDS reads, 4 C v_pk_fma_f32 (v_dot2 pairs for the half
// types), 8 reduction / select instructions -- no window request that depends on a row
// bound, no mask, no branch that depends on the data.
//
// One wave of the workgroup does NOTHING BUT COPY: the next slab and the next tile's
// descriptor list go to LDS (direct global->LDS copies) while the compute waves work on
// the current ones.  The compute waves therefore issue no vector-memory LOAD at all:
// their descriptors come from LDS (lgkmcnt), their only vector-memory operations are the
// result stores, which nothing ever waits for.  (First form of this kernel, measured:
// descriptors loaded from memory by the compute waves -- `vmcnt` retires in order, so
// every descriptor wait also waited for the acknowledgement of the stores issued in
// between, a microsecond per window: 41.7 us against 45.0 for the quad kernel, with the
// arithmetic alone at 19.)
//
// Which quad computes an entry decides the order in which its 16-byte chunks are added
// (the rotation that keeps the quads off each other's LDS banks), so the low bits differ
// from sddmm_quad_kernel's; both are held to the oracle.  Columns need not ascend (pairs
// are CSR neighbours wherever they lie): no per-row order check, no fallback path.
// Replaces sputnik::CudaSddmm at /root/reference/src/sddmm_cuda.cu:46-53 for masks
// planned ahead of the call (sputnik_hip_sddmm_plan): the attention scores and their
// gradient, /root/reference/modules/sparse_attention.py:65-72.
#include <type_traits>

#include "options.h"
#include "sddmm_dot.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {

namespace {

using namespace tiled;

constexpr int kFStep = 16;       // descriptors (pairs) per wave step: one per quad
constexpr unsigned kFSingle = 0x80000000u;

// Geometry.  Block rows x slab rows x waves; a descriptor stage stores STAGE steps of
// which a round uses at most STAGE - 8 (windows of four steps are read ahead: a request
// may run up to 7 steps past a chunk); a tile with more steps is worked on in several
// rounds of the same slab.
//   Big:   256 x 128, 16 waves (15 compute): 64 + 2 x 32 KiB of rows (float32 k = 64) +
//          2 x 15 KiB of descriptors = 158 KiB, one workgroup per CU
//   Small: 128 x  64,  8 waves ( 7 compute): 32 + 2 x 16 + 2 x 8 = 80 KiB, two workgroups
//          per CU: one's prologue and rendezvous overlap the other's arithmetic
template <int BR, int SR, int WAVES, int COPY, int STAGE>
struct Geometry {
  static constexpr int kBlock = BR, kSlab = SR, kWaves = WAVES, kCopy = COPY, kCompute = WAVES - COPY;
  static constexpr int kThreads = WAVES * kWave;
  static constexpr int kCap = STAGE - 8;                 // steps of a round
  static constexpr int kDescBytes = STAGE * kFStep * 8;
  // written behind tile_start by the plan, checked by the kernel: a plan made under one
  // geometry is never walked with the other's offsets (the geometry follows a
  // developer knob that a test may reload between plan and product)
  static constexpr int kTag = (BR << 16) | (SR << 4) | 0x5;
  static_assert(BR <= 256 && 2 * SR <= 256, "row fields of a descriptor are 8 bits");
  static_assert(kDescBytes % 1024 == 0, "descriptor stages are copied in 1 KiB pieces");
};
using Big = Geometry<256, 128, 16, 1, 120>;
using Small = Geometry<128, 64, 8, 1, 64>;

template <typename G> inline int f_slots(int m) { return ceil_div(m, G::kBlock) * G::kBlock; }
template <typename G> inline int f_blocks(int m) { return ceil_div(m, G::kBlock); }
template <typename G> inline int f_slabs(int n) { return ceil_div(n, G::kSlab); }

// Plan layout: [tile_start: tiles + 1 ints, steps before a tile][seg: slots x slabs ints,
// descriptors of a row in a slab][desc: uint2].  Upper bound of the descriptors: every
// entry a single, every tile padded to a whole step; + one stage of slack (the copy of a
// tile's last round moves whole KiB).
struct FlatPlan {
  int slots, blocks, slabs, tiles;
  size_t start_off, seg_off, desc_off, bytes;
  int64_t desc_capacity;   // descriptors
};
template <typename G>
FlatPlan make_plan(int m, int n, int nonzeros) {
  FlatPlan p;
  p.slots = f_slots<G>(m);
  p.blocks = f_blocks<G>(m);
  p.slabs = f_slabs<G>(n);
  p.tiles = p.blocks * p.slabs;
  auto up = [](size_t v) { return (v + 1023) / 1024 * 1024; };
  p.start_off = 0;
  p.seg_off = up(sizeof(int) * (static_cast<size_t>(p.tiles) + 2));   // (+ the geometry tag)
  p.desc_off = up(p.seg_off + sizeof(int) * static_cast<size_t>(p.slots) * p.slabs);
  p.desc_capacity = static_cast<int64_t>(nonzeros) + static_cast<int64_t>(kFStep) * p.tiles +
                    G::kDescBytes / 8;
  p.bytes = up(p.desc_off + sizeof(uint2) * static_cast<size_t>(p.desc_capacity));
  return p;
}

// The steps of a round are dealt to the compute waves in contiguous chunks: chunk c has
// steps / W steps, the first steps % W chunks one more; chunk c of round i goes to wave
// (c + i) % W, so that the longer chunks go round.
template <int W>
struct Chunks {
  int base, rem;
  __device__ explicit Chunks(int steps) : base(steps / W), rem(steps % W) {}
  __device__ int count(int c) const { return base + (c < rem ? 1 : 0); }
  __device__ int start(int c) const { return c * base + (c < rem ? c : rem); }
};

// ---------------------------------------------------------------------------
// Plan, step 1: one thread per row slot walks its row in storage order and pairs CSR
// neighbours that fall into the same slab (greedy: (p, p + 1) if both columns lie in one
// slab, else p alone).  seg[(block * slabs + slab) * BR + r] = descriptors of row slot r
// of the block in that slab (zeroed by the caller).
// ---------------------------------------------------------------------------
template <typename G>
__device__ __forceinline__ int slot_row(int slot, int slots, int m, const int* __restrict__ row_indices) {
  const int entry = dealt_index(slot, slots, G::kBlock);
  return entry < m ? row_indices[entry] : -1;
}

template <typename G>
__global__ __launch_bounds__(G::kBlock) void sddmm_flat_count_kernel(
    int m, int n, int slots, int slabs, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    int* __restrict__ seg) {
  const int block = blockIdx.x, r = threadIdx.x;
  const int row = slot_row<G>(block * G::kBlock + r, slots, m, row_indices);
  if (row < 0) return;
  const int p1 = row_offsets[row + 1];
  int p = row_offsets[row];
  int run_slab = -1, run = 0;
  auto flush = [&]() {
    if (run > 0) seg[(static_cast<int64_t>(block) * slabs + run_slab) * G::kBlock + r] += run;
  };
  while (p < p1) {
    const unsigned c0 = static_cast<unsigned>(column_indices[p]);
    if (c0 >= static_cast<unsigned>(n)) {   // (invalid input: the entry is never written)
      ++p;
      continue;
    }
    const int s0 = static_cast<int>(c0) / G::kSlab;
    int step = 1;
    if (p + 1 < p1) {
      const unsigned c1 = static_cast<unsigned>(column_indices[p + 1]);
      if (c1 < static_cast<unsigned>(n) && static_cast<int>(c1) / G::kSlab == s0) step = 2;
    }
    if (s0 != run_slab) {
      flush();
      run_slab = s0;
      run = 0;
    }
    ++run;
    p += step;
  }
  flush();
}

// Plan, step 2: steps of every tile; exclusive scan over the tiles (one workgroup: a plan
// is made once per static mask).
template <typename G>
__global__ __launch_bounds__(G::kBlock) void sddmm_flat_tile_steps_kernel(
    const int* __restrict__ seg, int* __restrict__ tile_steps) {
  __shared__ int part[G::kBlock / kWave];
  const int tile = blockIdx.x, r = threadIdx.x;
  int v = seg[static_cast<int64_t>(tile) * G::kBlock + r];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (r % kWave == 0) part[r / kWave] = v;
  __syncthreads();
  if (r == 0) {
    int total = 0;
    for (int w = 0; w < G::kBlock / kWave; ++w) total += part[w];
    tile_steps[tile] = (total + kFStep - 1) / kFStep;
  }
}
__global__ __launch_bounds__(1024) void sddmm_flat_scan_kernel(int tiles, int* __restrict__ tile_start,
                                                              int geometry_tag) {
  // in place: tile_start[t] holds the tile's steps on entry, the steps before it on exit;
  // tile_start[tiles] = all steps
  __shared__ int partial[1024];
  const int t = threadIdx.x;
  const int per = (tiles + 1023) / 1024;
  const int i0 = min(t * per, tiles), i1 = min(i0 + per, tiles);
  int sum = 0;
  for (int i = i0; i < i1; ++i) sum += tile_start[i];
  partial[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off *= 2) {
    const int v = t >= off ? partial[t - off] : 0;
    __syncthreads();
    partial[t] += v;
    __syncthreads();
  }
  int run = partial[t] - sum;
  for (int i = i0; i < i1; ++i) {
    const int steps = tile_start[i];
    tile_start[i] = run;
    run += steps;
  }
  if (t == 1023) {
    tile_start[tiles] = partial[1023];
    tile_start[tiles + 1] = geometry_tag;
  }
}

// Plan, step 3: one workgroup per tile; row slot r's descriptors go behind those of the
// slots before it (exclusive scan of seg over the block), the tile's last step is padded
// with dummies (single, p = -1: computed on LDS row 0, never stored).  The rhs rows carry
// the LDS stage of the slab (top bit): the kernel never switches stage addresses.
template <typename G>
__global__ __launch_bounds__(G::kBlock) void sddmm_flat_emit_kernel(
    int m, int n, int slots, int slabs, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const int* __restrict__ seg, const int* __restrict__ tile_start, uint2* __restrict__ desc) {
  __shared__ int scan[G::kBlock];
  const int tile = blockIdx.x, r = threadIdx.x;
  const int block = tile / slabs, slab = tile % slabs;
  const int mine = seg[static_cast<int64_t>(tile) * G::kBlock + r];
  scan[r] = mine;
  __syncthreads();
  for (int off = 1; off < G::kBlock; off *= 2) {
    const int v = r >= off ? scan[r - off] : 0;
    __syncthreads();
    scan[r] += v;
    __syncthreads();
  }
  const int total = scan[G::kBlock - 1];
  uint2* __restrict__ out = desc + static_cast<int64_t>(tile_start[tile]) * kFStep;
  const int padded = (total + kFStep - 1) / kFStep * kFStep;
  for (int i = total + r; i < padded; i += G::kBlock) out[i] = make_uint2(kFSingle, 0xffffffffu);
  if (mine == 0) return;
  int at = scan[r] - mine;
  const int row = slot_row<G>(block * G::kBlock + r, slots, m, row_indices);
  const int p1 = row_offsets[row + 1];
  int p = row_offsets[row];
  int left = mine;
  const unsigned stage = static_cast<unsigned>(slab & 1) * G::kSlab;
  while (p < p1 && left > 0) {
    const unsigned c0 = static_cast<unsigned>(column_indices[p]);
    if (c0 >= static_cast<unsigned>(n)) {
      ++p;
      continue;
    }
    const int s0 = static_cast<int>(c0) / G::kSlab;
    int step = 1;
    unsigned c1 = 0;
    if (p + 1 < p1) {
      c1 = static_cast<unsigned>(column_indices[p + 1]);
      if (c1 < static_cast<unsigned>(n) && static_cast<int>(c1) / G::kSlab == s0) step = 2;
    }
    if (s0 == slab) {
      const unsigned a = stage + c0 - static_cast<unsigned>(slab) * G::kSlab;
      const unsigned b = step == 2 ? stage + c1 - static_cast<unsigned>(slab) * G::kSlab : a;
      out[at++] = make_uint2(static_cast<unsigned>(r) | (a << 8) | (b << 16) | (step == 2 ? 0u : kFSingle),
                             static_cast<unsigned>(p));
      --left;
    }
    p += step;
  }
}

// ---------------------------------------------------------------------------
// The kernel.  T: storage type of lhs / rhs, TO: of the output; C: 16-byte chunks of a
// row quarter (row bytes = 64 C).
// ---------------------------------------------------------------------------
// Rendezvous of the workgroup's waves.  LDS reads are drained first (a stage may be
// overwritten behind it) and the compiler must not move LDS accesses across it: the
// builtin alone is "no memory" to the compiler, and the copies that fill the stages are
// inline asm it knows nothing about.  (No vmcnt wait: that is the copying wave's business.)
__device__ __forceinline__ void rendezvous() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// v of the lane CTRL names inside the quad (quad_perm), in a form the compiler folds into
// the instruction that uses it.
template <int CTRL>
__device__ __forceinline__ float quad_swap(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <typename G, int C>
struct Flat {
  static constexpr int kRowBytes = 64 * C;
  static constexpr int kQuarter = 16 * C;
  static constexpr int kLhsBytes = G::kBlock * kRowBytes;
  static constexpr int kStageBytes = G::kSlab * kRowBytes;
  static constexpr int kRowsBytes = kLhsBytes + 2 * kStageBytes;
  static constexpr int kLdsBytes = kRowsBytes + 2 * G::kDescBytes;
  static constexpr int kRowsPerPiece = 1024 / kRowBytes;   // rows a 1 KiB wave copy covers
  static_assert(kLdsBytes <= 160 * 1024, "LDS of a CU");
};

template <typename G, typename T, typename TO, int C>
__global__ __launch_bounds__(G::kThreads) void sddmm_flat_kernel(
    int m, int n, int slots, int slabs, const int* __restrict__ row_indices,
    const T* __restrict__ lhs, int64_t lhs_stride, const T* __restrict__ rhs, int64_t rhs_stride,
    int ld /* elements between rows of lhs / rhs */, TO* __restrict__ out, int64_t out_stride,
    const int* __restrict__ tile_start, const uint2* __restrict__ desc, int debug) {
  using F = Flat<G, C>;
  using chunk = typename Dot<T>::chunk;
  constexpr int kRowBytes = F::kRowBytes;
  constexpr int W = G::kCompute;
  // lhs block | rhs stage 0 | rhs stage 1 | descriptor stage 0 | descriptor stage 1
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, t = i & 3;
  // (row block fastest: the blocks of a replica stage the same rhs slabs -- one XCD's L2)
  const unsigned long long work = xcd_local_index();
  const int block = static_cast<int>(work % gridDim.x);
  const int replica = static_cast<int>(work / gridDim.x);
  lhs += replica * lhs_stride;
  rhs += replica * rhs_stride;
  out += replica * out_stride;

  // a plan of the other geometry (or no plan at all) is not walked: nothing is written
  if (tile_start[static_cast<int64_t>(gridDim.x) * slabs + 1] != G::kTag) return;

  const unsigned ld_bytes = static_cast<unsigned>(ld) * static_cast<unsigned>(sizeof(T));
  const int piece_row = (lane * 16) / kRowBytes;               // row of a 1 KiB piece this lane copies
  const unsigned piece_byte = static_cast<unsigned>((lane * 16) % kRowBytes);
  const int* __restrict__ my_tiles = tile_start + static_cast<int64_t>(block) * slabs;

  // wave `share` of `waves` waves' part of: slab `slab` into its stage; a round's
  // descriptors (steps first .. first + count - 1 of the plan) into theirs
  auto stage_slab = [&](int slab, int share, int waves) {
    constexpr int kPieces = F::kStageBytes / 1024;
    char* dst = smem + F::kLhsBytes + (slab & 1) * F::kStageBytes;
    for (int piece = share; piece < kPieces; piece += waves) {
      const int row = min(slab * G::kSlab + piece * F::kRowsPerPiece + piece_row, n - 1);
      lds_dma_row(reinterpret_cast<const float*>(rhs), static_cast<unsigned>(row) * ld_bytes + piece_byte,
                  reinterpret_cast<const float*>(dst + piece * 1024));
    }
  };
  auto stage_desc = [&](int round, int first, int count, int share, int waves) {
    char* dst = smem + F::kRowsBytes + (round & 1) * G::kDescBytes;
    const int pieces = (count * kFStep * 8 + 1023) / 1024;   // (whole KiB: the plan has slack behind it)
    const char* src = reinterpret_cast<const char*>(desc + static_cast<int64_t>(first) * kFStep);
    for (int piece = share; piece < pieces; piece += waves)
      lds_dma_row(reinterpret_cast<const float*>(src), static_cast<unsigned>(piece) * 1024u + lane * 16u,
                  reinterpret_cast<const float*>(dst + piece * 1024));
  };

  // The rounds: slab after slab; a tile with more than kCap steps takes several rounds on
  // the same slab.  Every wave walks the same sequence (wave-uniform scalars).
  struct Round {
    int slab, first, count;   // steps first .. first + count - 1 of the plan
  };
  auto tile_at = [&](int slab) { return __builtin_amdgcn_readfirstlane(my_tiles[slab]); };
  auto round_at = [&](int slab, int off) {   // `off` steps into the slab's tile
    const int t0 = tile_at(slab), t1 = tile_at(slab + 1);
    return Round{slab, t0 + off, min(t1 - t0 - off, G::kCap)};
  };
  auto next_round = [&](const Round& r, int& off) {   // false: `r` was the last one
    if (r.first + r.count < tile_at(r.slab + 1)) {
      off += G::kCap;
      return true;
    }
    off = 0;
    return r.slab + 1 < slabs;
  };

  // prologue, all waves: the lhs block (row slot r of the block at smem + r * kRowBytes),
  // the first slab, the first round's descriptors
  int off = 0;
  Round cur = round_at(0, 0);
  if (!(debug & 16)) {   // (timing experiment: no prologue copies)
    constexpr int kPieces = F::kLhsBytes / 1024;
    for (int piece = wave; piece < kPieces; piece += G::kWaves) {
      const int slot = block * G::kBlock + piece * F::kRowsPerPiece + piece_row;
      const int entry = dealt_index(slot, slots, G::kBlock);
      const int row = row_indices[entry < m ? entry : 0];   // (padding slots: any valid row, never used)
      lds_dma_row(reinterpret_cast<const float*>(lhs), static_cast<unsigned>(row) * ld_bytes + piece_byte,
                  reinterpret_cast<const float*>(smem + piece * 1024));
    }
    stage_slab(0, wave, G::kWaves);
  }
  stage_desc(0, cur.first, cur.count, wave, G::kWaves);
  wait_vm<0>();
  __syncthreads();

  if (wave >= W) {
    // The copying waves, one round ahead: while round i is worked on it brings in round
    // i + 1's descriptors and, when that round starts a new slab, the slab -- both into the
    // stages that round i - 1 used, free since the rendezvous that ended it.  Its vmcnt
    // holds nothing else.
    for (int round = 0;; ++round) {
      Round nxt = cur;
      int noff = off;
      const bool more = next_round(cur, noff);
      if (more) {
        nxt = round_at(noff == 0 ? cur.slab + 1 : cur.slab, noff);
        if (nxt.slab != cur.slab && !(debug & 2)) stage_slab(nxt.slab, wave - W, G::kCopy);
        stage_desc(round + 1, nxt.first, nxt.count, wave - W, G::kCopy);
        wait_vm<0>();
      }
      rendezvous();
      if (!more) return;
      cur = nxt;
      off = noff;
    }
  }

  // this lane's chunks of a row, in sddmm_quad_kernel's rotated order (bank conflicts:
  // the quads of a 16-lane group read different rows in one instruction)
  const int rot = q + 4 * ((t * C) >> 4);
  int coff[C];
#pragma unroll
  for (int c = 0; c < C; ++c) coff[c] = t * F::kQuarter + 16 * ((c + rot) % C);

  const unsigned smem_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(AS_LDS(smem)));
  const unsigned stage_base = smem_base + F::kLhsBytes;   // (descriptors carry the stage: rows 0 .. 2 SR - 1)
  // descriptor of a window this lane holds: step t of the window, quad (g, q) of the wave
  const unsigned lane_desc = static_cast<unsigned>(t * kFStep + g * 4 + q) * 8u;

  for (int round = 0;; ++round) {
    // this wave's chunk of the round's steps
    const Chunks<W> chunks(cur.count);
    const int c = (wave + round) % W;
    const int s_begin = chunks.start(c), s_count = chunks.count(c);
    const unsigned dstage = smem_base + F::kRowsBytes + (round & 1) * G::kDescBytes + lane_desc +
                            static_cast<unsigned>(s_begin) * (kFStep * 8);
    // (a request past the chunk reads other waves' descriptors or stale bytes of the stage,
    // at most 7 steps behind the round's: never used)
    using u2 = unsigned __attribute__((ext_vector_type(2)));
    auto fetch = [&](int step0) {
      const u2 v = *reinterpret_cast<const __attribute__((address_space(3))) u2*>(
          dstage + static_cast<unsigned>(step0) * (kFStep * 8));
      return make_uint2(v.x, v.y);
    };
    // (Measured and not kept: the LDS reads of step s + 1 issued by hand in front of the
    // arithmetic of step s, two register sets, 122 registers -- 42.3 against 40.4 us at config
    // 3: with four waves per SIMD the reads of one wave already overlap the arithmetic of
    // another as far as the LDS lets them.)
    const int n_steps = (debug & 8) ? 0 : s_count;   // (8: timing experiment, no work)
    uint2 d_next = fetch(0);
    for (int pos = 0; pos < n_steps; pos += 4) {
      const uint2 d = d_next;
      d_next = fetch(pos + 4);
      const unsigned la = smem_base + (d.x & 0xffu) * kRowBytes;
      const unsigned ra = stage_base + ((d.x >> 8) & 0xffu) * kRowBytes;
      const unsigned rb = stage_base + ((d.x >> 16) & 0xffu) * kRowBytes;
      float res_a = 0.f, res_b = 0.f;
      static_for<4>([&](auto Sc) {
        constexpr int kS = decltype(Sc)::value;
        if (pos + kS >= n_steps || (debug & 1)) return;   // (wave-uniform)
        chunk a[C], ba[C], bb[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
          a[cc] = *reinterpret_cast<const __attribute__((address_space(3))) chunk*>(
              static_cast<unsigned>(quad_bcast_add<kS>(static_cast<int>(la), coff[cc])));
          ba[cc] = *reinterpret_cast<const __attribute__((address_space(3))) chunk*>(
              static_cast<unsigned>(quad_bcast_add<kS>(static_cast<int>(ra), coff[cc])));
          bb[cc] = *reinterpret_cast<const __attribute__((address_space(3))) chunk*>(
              static_cast<unsigned>(quad_bcast_add<kS>(static_cast<int>(rb), coff[cc])));
        }
        v2f acc_a = {0.f, 0.f}, acc_b = {0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
          Dot<T>::mac(acc_a, a[cc], ba[cc]);
          Dot<T>::mac(acc_b, a[cc], bb[cc]);
        }
        // (the quad reduction as v_add_f32_dpp: the DPP move folds into the add when its
        // "old" value is free -- every lane of a quad_perm has a source)
        float da = acc_a.x + acc_a.y, db = acc_b.x + acc_b.y;
        da += quad_swap<kDppQuadXor1>(da);
        db += quad_swap<kDppQuadXor1>(db);
        da += quad_swap<kDppQuadXor2>(da);
        db += quad_swap<kDppQuadXor2>(db);
        res_a = (t == kS) ? da : res_a;
        res_b = (t == kS) ? db : res_b;
      });
      // this lane's pair: step t of the window
      if (pos + t < n_steps && static_cast<int>(d.y) >= 0 && !(debug & 4)) {
        TO* dst = reinterpret_cast<TO*>(reinterpret_cast<char*>(out) +
                                        static_cast<unsigned>(d.y) * static_cast<unsigned>(sizeof(TO)));
        if (d.x & kFSingle) {
          *dst = static_cast<TO>(res_a);
        } else if constexpr (std::is_same_v<TO, float>) {
          typedef float v2f_a4 __attribute__((ext_vector_type(2), aligned(4)));
          *reinterpret_cast<v2f_a4*>(dst) = v2f_a4{res_a, res_b};   // 8 bytes, 4-byte aligned: the type says so
        } else {
          dst[0] = static_cast<TO>(res_a);
          dst[1] = static_cast<TO>(res_b);
        }
      }
    }
    int noff = off;
    const bool more = next_round(cur, noff);
    Round nxt = cur;
    if (more) nxt = round_at(noff == 0 ? cur.slab + 1 : cur.slab, noff);
    rendezvous();
    if (!more) return;
    cur = nxt;
    off = noff;
  }
}

template <typename G, typename T, typename TO, int C>
int launch_flat(int m, int n, int nonzeros, int replicas, const int* row_indices, const T* lhs,
                int64_t lhs_stride, const T* rhs, int64_t rhs_stride, int ld, TO* out,
                int64_t out_stride, const char* plan, hipStream_t stream) {
  using F = Flat<G, C>;
  const FlatPlan p = make_plan<G>(m, n, nonzeros);
  static const bool lds_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(sddmm_flat_kernel<G, T, TO, C>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, F::kLdsBytes) == hipSuccess;
  }();
  if (!lds_ok) return SPUTNIK_HIP_UNSUPPORTED;
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int rz = min(replicas - r0, kMaxGridYZ);
    hipLaunchKernelGGL((sddmm_flat_kernel<G, T, TO, C>), dim3(p.blocks, rz), dim3(G::kThreads),
                       F::kLdsBytes, stream, m, n, p.slots, p.slabs, row_indices,
                       lhs + r0 * lhs_stride, lhs_stride, rhs + r0 * rhs_stride, rhs_stride, ld,
                       out + r0 * out_stride, out_stride,
                       reinterpret_cast<const int*>(plan + p.start_off),
                       reinterpret_cast<const uint2*>(plan + p.desc_off), options().sddmm_debug >> 8);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

template <typename G>
int plan_flat(int m, int n, int nonzeros, const int* row_indices, const int* row_offsets,
              const int* column_indices, void* plan, hipStream_t stream) {
  const FlatPlan p = make_plan<G>(m, n, nonzeros);
  char* base = static_cast<char*>(plan);
  int* tile_start = reinterpret_cast<int*>(base + p.start_off);
  int* seg = reinterpret_cast<int*>(base + p.seg_off);
  uint2* desc = reinterpret_cast<uint2*>(base + p.desc_off);
  hipError_t e = hipMemsetAsync(seg, 0, sizeof(int) * static_cast<size_t>(p.slots) * p.slabs, stream);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(sddmm_flat_count_kernel<G>, dim3(p.blocks), dim3(G::kBlock), 0, stream, m, n,
                     p.slots, p.slabs, row_indices, row_offsets, column_indices, seg);
  hipLaunchKernelGGL(sddmm_flat_tile_steps_kernel<G>, dim3(p.tiles), dim3(G::kBlock), 0, stream, seg,
                     tile_start);
  hipLaunchKernelGGL(sddmm_flat_scan_kernel, dim3(1), dim3(1024), 0, stream, p.tiles, tile_start, G::kTag);
  hipLaunchKernelGGL(sddmm_flat_emit_kernel<G>, dim3(p.tiles), dim3(G::kBlock), 0, stream, m, n,
                     p.slots, p.slabs, row_indices, row_offsets, column_indices, seg, tile_start, desc);
  return launch_status();
}

// SPUTNIK_HIP_SDDMM_FLAT: 0 off, 1 the big geometry, 2 the small one.
inline bool small_geometry() { return options().sddmm_flat == 2; }

}  // namespace

// Row bytes of the operands this kernel serves: 128 or 256 (k = 64 float32; k = 64 / 128
// float16 / bfloat16); at least half a row block and one slab; 32-bit byte offsets.
bool sddmm_flat_applicable(int m, int k, int n, int nonzeros, int elem_bytes) {
  const int64_t row_bytes = static_cast<int64_t>(k) * elem_bytes;
  if (options().sddmm_flat == 0) return false;
  return (row_bytes == 128 || row_bytes == 256) && m >= 128 && n >= 128 &&
         nonzeros >= 4 * static_cast<int64_t>(m) && nonzeros < (1 << 29) &&
         static_cast<int64_t>(m) * row_bytes < (int64_t{1} << 32) &&
         static_cast<int64_t>(n) * row_bytes < (int64_t{1} << 32) &&
         static_cast<int64_t>(f_blocks<Small>(m)) * f_slabs<Small>(n) < (1 << 22);
}

size_t sddmm_flat_plan_bytes(int m, int n, int nonzeros) {
  return small_geometry() ? make_plan<Small>(m, n, nonzeros).bytes : make_plan<Big>(m, n, nonzeros).bytes;
}

int sddmm_flat_plan(int m, int n, int nonzeros, const int* row_indices, const int* row_offsets,
                    const int* column_indices, void* plan, hipStream_t stream) {
  return small_geometry()
             ? plan_flat<Small>(m, n, nonzeros, row_indices, row_offsets, column_indices, plan, stream)
             : plan_flat<Big>(m, n, nonzeros, row_indices, row_offsets, column_indices, plan, stream);
}

// in_type / out_type: SPUTNIK_HIP_F32 / F16 / BF16 (out float32 or in_type).
int sddmm_flat_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                      const void* lhs, int64_t lhs_stride, const void* rhs, int64_t rhs_stride,
                      void* out, int64_t out_stride, int in_type, int out_type, const void* plan,
                      hipStream_t stream) {
  const char* base = static_cast<const char*>(plan);
#define SPUTNIK_HIP_FLAT_G(G, T, TO, C)                                                              \
  return launch_flat<G, T, TO, C>(m, n, nonzeros, replicas, row_indices, static_cast<const T*>(lhs), \
                                  lhs_stride, static_cast<const T*>(rhs), rhs_stride, k,             \
                                  static_cast<TO*>(out), out_stride, base, stream)
#define SPUTNIK_HIP_FLAT(T, TO, C)                             \
  do {                                                         \
    if (small_geometry()) SPUTNIK_HIP_FLAT_G(Small, T, TO, C); \
    SPUTNIK_HIP_FLAT_G(Big, T, TO, C);                         \
  } while (0)
  if (in_type == SPUTNIK_HIP_F32 && out_type == SPUTNIK_HIP_F32 && k == 64) SPUTNIK_HIP_FLAT(float, float, 4);
  if (in_type == SPUTNIK_HIP_F16 && k == 64) {
    if (out_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_FLAT(_Float16, float, 2);
    if (out_type == SPUTNIK_HIP_F16) SPUTNIK_HIP_FLAT(_Float16, _Float16, 2);
  }
  if (in_type == SPUTNIK_HIP_F16 && k == 128) {
    if (out_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_FLAT(_Float16, float, 4);
    if (out_type == SPUTNIK_HIP_F16) SPUTNIK_HIP_FLAT(_Float16, _Float16, 4);
  }
  if (in_type == SPUTNIK_HIP_BF16 && k == 64) {
    if (out_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_FLAT(__bf16, float, 2);
    if (out_type == SPUTNIK_HIP_BF16) SPUTNIK_HIP_FLAT(__bf16, __bf16, 2);
  }
  if (in_type == SPUTNIK_HIP_BF16 && k == 128) {
    if (out_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_FLAT(__bf16, float, 4);
    if (out_type == SPUTNIK_HIP_BF16) SPUTNIK_HIP_FLAT(__bf16, __bf16, 4);
  }
#undef SPUTNIK_HIP_FLAT
#undef SPUTNIK_HIP_FLAT_G
  return SPUTNIK_HIP_INVALID_ARGUMENT;
}

}  // namespace sputnik_hip
