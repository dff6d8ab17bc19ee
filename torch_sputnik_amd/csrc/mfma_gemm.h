// The matrix-core tile kernel of the half-storage extension (round 5): ONE template serves
// every product of a sparse layer on half-stored operands (sddmm_mfma.hip, spmm_mfma.hip,
// sparse_linear_half.hip say which):
//
//   C[m, n] = sum over (replica, k) of A[m, k] * B[k, n]          float32 accumulation
//
// on v_mfma_f32_32x32x16_{f16,bf16}, 128 x 128 tiles, four waves of 64 x 64, K in steps of
// 64 through two LDS stages filled by direct global->LDS copies.
//
// Operand layouts (each operand on its own, so that a product reads the tensors as the
// caller has them -- no transposed copies):
//   k-contiguous  (KM = false): element (r, k) at r * ld + k.  A step's tile is 128 rows of
//       128 bytes (8 rows per wave copy); 16-byte slot g of row r lands in slot
//       g ^ ((r >> 1) & 7), fragments are ds_read_b128.
//   k-major       (KM = true):  element (r, k) at k * ld + r.  A step's tile is 64 rows (k)
//       of 256 bytes (4 rows per wave copy); chunk c of row q lands in chunk
//       c ^ (((q & 3) << 2) | ((q >> 2) & 3)), and a fragment -- 8 consecutive k of one r --
//       is two ds_read_b64_tr_b16 (the hardware's transposing 4 x 16 read).
//   Both swizzles sit in the per-lane SOURCE address (the LDS image of such a copy is
//   lane-linear) and make the reads bank-conflict free (SQ_LDS_BANK_CONFLICT = 0,
//   profiles/r5a_pmc_sq_c5_fp16_step.json).
//
// Planes: an operand that arrived as float32 is given as PA / PB half planes whose (scaled)
// sum is the value (split_planes_kernel / densify_kernel); a step multiplies every pair of
// planes whose order a + b stays below the longer plane count.  float16 keeps its low plane
// scaled by 2^11 (out of the subnormals) and accumulates the order-1 products in a second
// tile (ACCS = 2) that enters the result times 2^-11.
//
// Epilogues: kDense (float32 tile, bias / ReLU), kDenseHalf (rounded to T), kSampled (the
// tile goes to LDS and the entries of a CSR mask that fall into it are stored: SDDMM).
//
// Where the time of a launch goes (config 5's forward pass, one plane pair, 34.4 dense GFLOP;
// SPUTNIK_HIP_MFMA_DEBUG switches parts off, tools/half_linear_bench.py): 40.4 us, of which
// 19.6 us remain with copies, fragment reads and MFMAs all switched off -- launch, the first
// tiles' latency, 32 rendezvous and 33.5 MB of output stores; the steps themselves take the
// other ~21 us = 1.6 PFLOP/s (dense).  Two rules of the epilogues come from that accounting:
// vmcnt retires in order, so NO load may sit between the stores (a per-row bias load in front
// of every store pair made each store wait for all earlier ones: 28 us of skeleton instead of
// 19.6), and the steps' tile origins advance by scalar additions (a division per step cost
// as much as the step's arithmetic).
#pragma once

#include <type_traits>

#include "mfma_tiles.h"
#include "options.h"

namespace sputnik_hip {
namespace mfma_tiles {

typedef short s4v __attribute__((__vector_size__(8)));

constexpr int kDense = 0, kSampled = 1, kDenseHalf = 2;

// The (A plane, B plane) pairs a step multiplies.
template <int PA, int PB>
struct Passes {
  static constexpr int kLimit = (PA > PB ? PA : PB) - 1;
  static constexpr int count() {
    int c = 0;
    for (int a = 0; a < PA; ++a)
      for (int b = 0; b < PB; ++b) c += a + b <= kLimit ? 1 : 0;
    return c;
  }
  static constexpr int a_of(int i) {
    int c = 0;
    for (int a = 0; a < PA; ++a)
      for (int b = 0; b < PB; ++b)
        if (a + b <= kLimit && c++ == i) return a;
    return 0;
  }
  static constexpr int b_of(int i) {
    int c = 0;
    for (int a = 0; a < PA; ++a)
      for (int b = 0; b < PB; ++b)
        if (a + b <= kLimit && c++ == i) return b;
    return 0;
  }
};

struct GemmOperand {
  const void* base;        // plane 0, replica 0
  int64_t ld;              // elements between consecutive rows (k-contiguous) / k (k-major)
  int64_t replica_stride;  // elements
  int64_t plane_stride;    // elements
};

struct GemmOut {
  // dense epilogues
  void* dense;             // [outer][m][ld]
  int64_t ld;
  int64_t outer_stride;
  const float* bias;       // [m] or null
  int relu;
  // sampled epilogue
  float* sampled;          // [outer][nonzeros]
  const int* row_offsets;
  const int* column_indices;
  const int* plan;         // sddmm_mfma_plan_kernel's table, or null
  int nonzeros;
  int vector_columns;
  // SPUTNIK_HIP_MFMA_DEBUG (timing experiments, wrong results): 1 no MFMAs, 2 no copies after
  // the prologue's, 4 no fragment reads
  int debug;
};

// One operand's tile of a step: R rows (R = 128 or 256: the extent of m or n it covers) by 64
// k, copied by W waves (W * PIECES pieces of 1 KiB).  Per-lane source byte offsets of this
// wave's pieces, the pieces' LDS offsets, and the fragment read addresses of a wave's 64-row
// block (all relative to the operand's place in a stage).
template <bool KM, int R, int W>
struct TileMap {
  static constexpr int kBytes = R * kStep * 2;             // 16 / 32 KiB
  static constexpr int kPieces = kBytes / 1024 / W;        // per wave
  static constexpr int kRowBytes = KM ? R * 2 : kStep * 2; // 256 / 512 (k-major), 128 (k-contiguous)
  // LDS offset of this wave's piece j
  static __device__ __forceinline__ int piece_offset(int wave, int j) { return (wave + W * j) * 1024; }
  // rows: extent of the r dimension (m or n); r0: first r of the tile
  static __device__ __forceinline__ void copy_offsets(unsigned (&off)[kPieces], int wave, int lane, int r0,
                                                      int rows, int64_t ld) {
    if constexpr (!KM) {
#pragma unroll
      for (int j = 0; j < kPieces; ++j) {
        const int row = 8 * (wave + W * j) + (lane >> 3);
        const unsigned slot = static_cast<unsigned>((lane & 7) ^ ((row >> 1) & 7)) * 16u;
        // (rows beyond the matrix are clamped onto its last row: never stored / sampled)
        off[j] = static_cast<unsigned>(min(r0 + row, rows - 1) - r0) * static_cast<unsigned>(ld) * 2u + slot;
      }
    } else {
      const int last_chunk = (rows - r0) / 8 - 1;   // (rows is a multiple of 8)
      constexpr int kPerRow = kRowBytes / 16;        // 16-byte chunks per k row: 16 / 32
      constexpr int kRowsPerPiece = 64 / kPerRow;    // 4 / 2
#pragma unroll
      for (int j = 0; j < kPieces; ++j) {
        const int in_piece = lane / kPerRow;                       // k row inside the piece
        const int krow = kRowsPerPiece * (wave + W * j) + in_piece;
        const int at = lane % kPerRow;                             // LDS chunk inside the row
        // (a 512-byte row is two 256-byte bank rows, swizzled each on its own)
        const int chunk = (at & ~15) | ((at & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3)));
        off[j] = static_cast<unsigned>(krow) * static_cast<unsigned>(ld) * 2u +
                 static_cast<unsigned>(min(chunk, last_chunk)) * 16u;
      }
    }
  }
  // global address of the tile's first element at k-step offset k0 (elements)
  static __device__ __forceinline__ int64_t tile_origin(int r0, int64_t k0, int64_t ld) {
    return KM ? k0 * ld + r0 : static_cast<int64_t>(r0) * ld + k0;
  }
  // fragment addresses of the wave's two 32-row blocks at rows w * 64 of the tile:
  //   k-contiguous: addr[i][0] (k slot 0; slot 2 ks + (lane >> 5) is an XOR with ks * 32)
  //   k-major:      addr[i][t] for the two transposing reads (k 0-3, 4-7) at ks = 0; + ks * 16 rows
  static __device__ __forceinline__ void fragment_addresses(unsigned (&addr)[2][2], int w, int lane) {
    if constexpr (!KM) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = w * 64 + i * 32 + (lane & 31);
        addr[i][0] = static_cast<unsigned>(r * 128 + (((lane >> 5) ^ ((r >> 1) & 7)) * 16));
        addr[i][1] = 0;
      }
    } else {
      // lane 4q + p of a 16-lane group supplies row q, columns 4p .. 4p + 3 of a 4 x 16 block
      // and receives column (lane & 15) of its four rows
      const int q = (lane & 15) >> 2, p = lane & 3, h = lane >> 5, g1 = (lane >> 4) & 1;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = 8 * h + 4 * t + q;
          const int chunk = w * 8 + i * 4 + 2 * g1 + (p >> 1);
          const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
          addr[i][t] = static_cast<unsigned>(row * kRowBytes + ((chunk & ~15) | ((chunk & 15) ^ swz)) * 16 +
                                             8 * (p & 1));
        }
    }
  }
  template <typename frag>
  static __device__ __forceinline__ frag read(const char* stage_operand, const unsigned (&addr)[2], int ks) {
    if constexpr (!KM) {
      return *reinterpret_cast<const frag*>(stage_operand + (addr[0] ^ (ks * 32u)));
    } else {
      typedef short s8v __attribute__((__vector_size__(16)));
      const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s4v*)(stage_operand + (addr[0] + ks * (16u * kRowBytes))));
      const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s4v*)(stage_operand + (addr[1] + ks * (16u * kRowBytes))));
      const s8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(frag, both);
    }
  }
};

// Geometry of a workgroup's tile: TM rows (128: four waves, two LDS stages, two workgroups
// per CU; 256: eight waves, three stages -- the copies of two steps in flight -- one
// workgroup per CU, 87 flop per staged byte instead of 64) by 128 columns.
template <int TM, int EPI>
struct TileGeometry {
  static constexpr int kWaves = TM / 32;
  static constexpr int kThreads = 64 * kWaves;
  static constexpr int kStages = TM == 256 ? 3 : 2;
  static constexpr int kABytes = TM * kStep * 2, kBBytes = kTile * kStep * 2;
  static constexpr int kStage = kABytes + kBBytes;
  static constexpr int kTileFloats = TM * kPitch;                 // the sampled epilogue's tile
  static constexpr int kBounds = 4 * (TM + 4);
  static constexpr int kSmem = EPI == kSampled && kTileFloats * 4 + kBounds > kStages * kStage
                                   ? kTileFloats * 4 + kBounds : kStages * kStage;
};

template <typename T, int TM, bool AKM, bool BKM, int PA, int PB, int ACCS, int EPI>
__global__ __launch_bounds__((TileGeometry<TM, EPI>::kThreads), (TM == 256 ? 1 : 2)) void mfma_gemm_kernel(
    int m, int n, int k, int tiles_m, int tiles_n, int steps_per_replica, int total_steps,
    int outers, int outer_is_split, GemmOperand a_op, GemmOperand b_op, GemmOut out,
    float low_scale) {
  using H = Half8<T>;
  using frag = typename H::type;
  using P = Passes<PA, PB>;
  using G = TileGeometry<TM, EPI>;
  using MapA = TileMap<AKM, TM, G::kWaves>;
  using MapB = TileMap<BKM, kTile, G::kWaves>;
  constexpr int NP = P::count();
  __shared__ __attribute__((aligned(16))) char smem[G::kSmem];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (an SGPR: M0 takes it)
  const int wr = wave >> 1, wc = wave & 1;
  // Consecutive work indices run behind one L2.  Sampled (split workgroups of one output):
  // the splits of a tile, then the row tiles of one tile column (same B panel).  Dense
  // (independent outputs): the row tiles of one column tile, then the column tiles of one
  // output.
  const int work = xcd_local_index32();
  int outer, rt, ct;
  if (outer_is_split) {
    outer = work % outers;
    const int tile = work / outers;
    rt = tile % tiles_m;
    ct = tile / tiles_m;
  } else {
    rt = work % tiles_m;
    ct = (work / tiles_m) % tiles_n;
    outer = work / (tiles_m * tiles_n);
  }
  const int r0 = rt * TM, c0 = ct * kTile;
  // the (replica, k step) pairs this workgroup reduces
  const int s_begin = outer_is_split ? static_cast<int>(static_cast<int64_t>(total_steps) * outer / outers)
                                     : outer * steps_per_replica;
  const int s_end = outer_is_split ? static_cast<int>(static_cast<int64_t>(total_steps) * (outer + 1) / outers)
                                   : (outer + 1) * steps_per_replica;

  unsigned a_off[MapA::kPieces], b_off[MapB::kPieces];
  MapA::copy_offsets(a_off, wave, lane, r0, m, a_op.ld);
  MapB::copy_offsets(b_off, wave, lane, c0, n, b_op.ld);
  // The operands' tile origins step by step, advanced by scalar additions (a division per
  // step -- which replica, which k -- cost as much as the step's arithmetic).
  struct StepAt {
    const T* a;
    const T* b;
    int kk;   // k step inside the replica
  };
  const int64_t a_step = AKM ? kStep * a_op.ld : kStep, b_step = BKM ? kStep * b_op.ld : kStep;
  const int64_t a_jump = a_op.replica_stride - (steps_per_replica - 1) * a_step;
  const int64_t b_jump = b_op.replica_stride - (steps_per_replica - 1) * b_step;
  auto advanced = [&](StepAt at) {
    if (++at.kk == steps_per_replica) {
      at.kk = 0;
      at.a += a_jump;
      at.b += b_jump;
    } else {
      at.a += a_step;
      at.b += b_step;
    }
    return at;
  };
  StepAt at0;
  {
    const int replica = s_begin / steps_per_replica;
    at0.kk = s_begin - replica * steps_per_replica;
    at0.a = uniform_ptr(static_cast<const T*>(a_op.base) + replica * a_op.replica_stride +
                        MapA::tile_origin(r0, static_cast<int64_t>(at0.kk) * kStep, a_op.ld));
    at0.b = uniform_ptr(static_cast<const T*>(b_op.base) + replica * b_op.replica_stride +
                        MapB::tile_origin(c0, static_cast<int64_t>(at0.kk) * kStep, b_op.ld));
  }
  auto stage = [&](const StepAt& at, int pass, unsigned buffer_offset, bool first) {
    if ((out.debug & 2) && !first) return;
    const T* a = at.a + P::a_of(pass) * a_op.plane_stride;
    const T* b = at.b + P::b_of(pass) * b_op.plane_stride;
    const char* dst = smem + buffer_offset;
#pragma unroll
    for (int j = 0; j < MapA::kPieces; ++j) copy_piece(a, a_off[j], dst + MapA::piece_offset(wave, j));
#pragma unroll
    for (int j = 0; j < MapB::kPieces; ++j)
      copy_piece(b, b_off[j], dst + G::kABytes + MapB::piece_offset(wave, j));
  };

  unsigned fa[2][2], fb[2][2];
  MapA::fragment_addresses(fa, wr, lane);
  MapB::fragment_addresses(fb, wc, lane);

  f32x16 acc[ACCS][2][2];
#pragma unroll
  for (int z = 0; z < ACCS; ++z)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[z][i][j] = f32x16{};

  // one tile product: fragments from the stage at `offset`, 16 MFMAs
  auto multiply = [&](unsigned offset, bool low) {
    const char* sa = smem + offset;
    const char* sb = sa + G::kABytes;
    frag a[2][4], b[2][4];
    if (out.debug & 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i][ks] = frag{};
          b[i][ks] = frag{};
          asm volatile("" : "+v"(a[i][ks]), "+v"(b[i][ks]));
        }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i][ks] = MapA::template read<frag>(sa, fa[i], ks);
          b[i][ks] = MapB::template read<frag>(sb, fb[i], ks);
        }
    }
    if (out.debug & 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(a[i][ks]), "v"(b[i][ks]));
      return;
    }
    constexpr int kOne = ACCS == 2 ? 1 : 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (low) acc[kOne][i][j] = H::mfma(a[i][ks], b[j][ks], acc[kOne][i][j]);
          else acc[0][i][j] = H::mfma(a[i][ks], b[j][ks], acc[0][i][j]);
        }
  };

  if constexpr (G::kStages == 2) {
    if (s_begin < s_end) {
      stage(at0, 0, 0, true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    unsigned stage_base = 0;
    StepAt at = at0;
    for (int s = s_begin; s < s_end; ++s) {
      const StepAt next = advanced(at);
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        const unsigned other = stage_base == 0 ? G::kStage : 0;
        if (pass + 1 < NP) {
          stage(at, pass + 1, other, s == s_begin);
        } else if (s + 1 < s_end) {
          stage(next, 0, other, false);
        }
        multiply(stage_base, P::a_of(pass) + P::b_of(pass) > 0);
        // the next tiles have landed (this wave's copies), and every wave is done with these
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage_base = other;
      }
      at = next;
    }
  } else {
    // Three stages: the copies of the NEXT TWO tile products are in flight while one is
    // multiplied.  Per product q: wait for this wave's copies of q (all but the youngest
    // stage's -- counted, the copies are the loop's only vector-memory operations),
    // rendezvous (every wave's copies of q have landed, and every wave is done reading the
    // stage that q + 2 goes into: it held q - 1), issue q + 2, multiply q.
    constexpr int kCopies = MapA::kPieces + MapB::kPieces;   // per wave and stage
    const int products = (s_end - s_begin) * NP;
    // (q = (s - s_begin) * NP + pass: product q + d belongs to step s + (pass + d) / NP)
    StepAt at[3];
    at[0] = at0;
    at[1] = advanced(at[0]);
    at[2] = advanced(at[1]);
    if (products > 0) stage(at[0], 0, 0, true);
    if (products > 1) stage(at[1 / NP], 1 % NP, G::kStage, NP > 1);
    unsigned cur = 0, nxt = G::kStage, nxt2 = 2 * G::kStage;
    int q = 0;
    for (int s = s_begin; s < s_end; ++s) {
#pragma unroll
      for (int pass = 0; pass < NP; ++pass, ++q) {
        if (q + 1 < products) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kCopies) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (q + 2 < products) stage(at[(pass + 2) / NP], (pass + 2) % NP, nxt2, false);
        multiply(cur, P::a_of(pass) + P::b_of(pass) > 0);
        const unsigned t = cur;
        cur = nxt;
        nxt = nxt2;
        nxt2 = t;
      }
      at[0] = at[1];
      at[1] = at[2];
      at[2] = advanced(at[2]);
    }
    __syncthreads();   // (the epilogue below writes over the stages)
  }

  auto value = [&](int i, int j, int reg) {
    return ACCS == 2 ? fmaf(acc[ACCS - 1][i][j][reg], low_scale, acc[0][i][j][reg]) : acc[0][i][j][reg];
  };

  if constexpr (EPI != kSampled) {
    // ---- the accumulators' 32 x 32 blocks (column = lane & 31: 128- / 64-byte runs) ----
    // (vmcnt retires in order: a load between the stores would make every store wait for
    // the ones before it -- the bias values are all fetched BEFORE the first store, and
    // the path without a bias holds no load at all)
    using TO = typename std::conditional<EPI == kDenseHalf, T, float>::type;
    TO* __restrict__ o = static_cast<TO*>(out.dense) + outer * out.outer_stride;
    const int row_base = r0 + wr * 64 + 4 * (lane >> 5);
    const int col_base = c0 + wc * 64 + (lane & 31);
    auto store_all = [&](auto bias_of) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = row_base + i * 32 + (reg & 3) + 8 * (reg >> 2);
          if (row < m) {
            TO* __restrict__ orow = o + static_cast<int64_t>(row) * out.ld;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int col = col_base + j * 32;
              float v = value(i, j, reg) + bias_of(i, reg);
              if (out.relu) v = fmaxf(v, 0.f);
              if (col < n) orow[col] = static_cast<TO>(v);
            }
          }
        }
    };
    if (out.bias == nullptr) {
      store_all([](int, int) { return 0.f; });
    } else {
      float bv[2][16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          bv[i][reg] = out.bias[min(row_base + i * 32 + (reg & 3) + 8 * (reg >> 2), m - 1)];
      store_all([&](int i, int reg) { return bv[i][reg]; });
    }
  } else {
    // ---- the tile to LDS, then the mask rows' entries that fall into it ----
    float* tile_lds = reinterpret_cast<float*>(smem);
    int* bounds = reinterpret_cast<int*>(smem + G::kTileFloats * 4);   // [TM + 1]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
          const int col = wc * 64 + j * 32 + (lane & 31);
          tile_lds[row * kPitch + col] = value(i, j, reg);
        }
    float* __restrict__ o = out.sampled + static_cast<int64_t>(outer) * out.nonzeros;
    const int* __restrict__ column_indices = out.column_indices;
    // With a plan (sddmm_mfma_plan: where every row's entries cross the tile columns) whose
    // rows all have ascending columns, a row's entries inside this tile are one known run.
    bool by_table = out.plan != nullptr;
    if (by_table) {
      const int ok = threadIdx.x < TM ? out.plan[min(r0 + static_cast<int>(threadIdx.x), m - 1)] : 1;
      by_table = __syncthreads_and(ok) != 0;
    } else {
      __syncthreads();
    }
    if (by_table) {
      const int* table = out.plan + plan_rows(m);
      constexpr int kGroups = G::kThreads / 16;
      const int group = threadIdx.x >> 4, l16 = threadIdx.x & 15;
      int from[8], to[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // (all bounds requested before the first is used)
        const int row = r0 + group + kGroups * j;
        const int* run = table + static_cast<int64_t>(min(row, m - 1)) * (tiles_n + 1) + ct;
        from[j] = run[0];
        to[j] = row < m ? run[1] : run[0];
      }
      // vmcnt retires in order, so a column load behind a store waits for that store: the
      // first 32 entries of every row's run -- all of them at layer densities -- have their
      // columns fetched BEFORE the first store; longer runs (dense masks) go on below
      int cols[8][2];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int p = from[j] + l16 + 16 * e;
          cols[j][e] = p < to[j] ? column_indices[p] : c0;
        }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* tile_row = tile_lds + (group + kGroups * j) * kPitch - c0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int p = from[j] + l16 + 16 * e;
          if (p < to[j]) o[p] = tile_row[cols[j][e]];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* tile_row = tile_lds + (group + kGroups * j) * kPitch - c0;
        for (int p = from[j] + l16 + 32; p < to[j]; p += 16) o[p] = tile_row[column_indices[p]];
      }
      return;
    }
    // No plan, or a row whose columns do not ascend: the workgroup walks the CSR entries of
    // its 128 rows FLAT (they are contiguous in column_indices), sixteen bytes per lane,
    // and stores those whose column lies in the tile.
    if (threadIdx.x <= TM)
      bounds[threadIdx.x] = out.row_offsets[min(r0 + static_cast<int>(threadIdx.x), m)];
    __syncthreads();
    const int first = bounds[0], last = bounds[TM];
    // a lane takes four consecutive entries per round; its row moves forward only
    int row = 0;
    const int start = (first & ~3) + 4 * static_cast<int>(threadIdx.x);
    {  // first row whose end lies behind `start` (binary search over the 128 bounds)
      int lo = 0, hi = TM;   // answer in [lo, hi]
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bounds[mid + 1] > start) hi = mid; else lo = mid + 1;
      }
      row = lo;
    }
    for (int p = start; p < last; p += 4 * G::kThreads) {
      int cols[4];
      if (out.vector_columns && p + 3 < out.nonzeros) {
        const int4 v = *reinterpret_cast<const int4*>(column_indices + p);
        cols[0] = v.x; cols[1] = v.y; cols[2] = v.z; cols[3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) cols[e] = p + e < out.nonzeros ? column_indices[p + e] : -1;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = p + e;
        while (row < TM && bounds[row + 1] <= q) ++row;
        const unsigned c = static_cast<unsigned>(cols[e] - c0);
        if (q >= first && q < last && c < static_cast<unsigned>(kTile)) o[q] = tile_lds[row * kPitch + c];
      }
    }
  }
}

// 256-row tiles (one workgroup of eight waves per CU, three stages: 87 flop per staged byte
// instead of 64) are built and parity-tested, and NOT taken by default: measured at config 5
// (tools/half_linear_bench.py, us, 128 / 256 rows): forward 40.4 / 41.7, weight gradient
// 75.3 / 80.6, input gradient 72.4 / 79.4 (float32 values: 90.3 / 105.2) -- with the steps'
// loop at 1.6 PFLOP/s either way, what is left of a launch at this size is its skeleton
// (launch, first tiles, the output's stores), and one workgroup per CU hides less of it than
// two.  SPUTNIK_HIP_MFMA_TILE=256 takes them wherever the output has 256 rows.
inline bool wide_tile(int m, int n, int64_t independent_outputs) {
  (void)n;
  (void)independent_outputs;
  return options().mfma_tile == 256 && m >= 256;
}

// Host side of a launch.  outers: independent outputs (dense) or workgroups that share a
// tile of ONE summed output (sampled, outer_is_split).
template <typename T, int TM, bool AKM, bool BKM, int PA, int PB, int ACCS, int EPI>
inline int launch_mfma_gemm(int m, int n, int k, int replicas, int outers, bool outer_is_split,
                            const GemmOperand& a, const GemmOperand& b, const GemmOut& out,
                            float low_scale, hipStream_t stream) {
  using G = TileGeometry<TM, EPI>;
  const int tiles_m = ceil_div(m, TM), tiles_n = ceil_div(n, kTile);
  const int steps_per_replica = k / kStep;
  const int64_t total_steps = static_cast<int64_t>(replicas) * steps_per_replica;
  const int64_t blocks = static_cast<int64_t>(tiles_m) * tiles_n * outers;
  if (total_steps >= (int64_t{1} << 31) || blocks >= (int64_t{1} << 31) || outers < 1)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  GemmOut out_dbg = out;
  out_dbg.debug = options().mfma_debug;
  auto kernel = mfma_gemm_kernel<T, TM, AKM, BKM, PA, PB, ACCS, EPI>;
  // (G::kSmem is static LDS: up to the CU's 160 KiB launches as it is)
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(G::kThreads), 0, stream, m, n, k,
                     tiles_m, tiles_n, steps_per_replica, static_cast<int>(total_steps), outers,
                     outer_is_split ? 1 : 0, a, b, out_dbg, low_scale);
  return launch_status();
}

// Dispatch over the plane counts a storage type has: float16 (1 | 2 planes, two tiles as
// soon as there is a low plane), bfloat16 (1 | 3 planes, one tile; 3 x 3 is not built).
// Returns SPUTNIK_HIP_UNSUPPORTED for a combination that is not instantiated.
template <bool AKM, bool BKM, int EPI>
inline int launch_mfma_gemm_typed(int tile_type, int pa, int pb, int m, int n, int k, int replicas,
                                  int outers, bool outer_is_split, const GemmOperand& a,
                                  const GemmOperand& b, const GemmOut& out, hipStream_t stream,
                                  bool wide) {
  const float low = 1.f / kLowPlaneScale;
#define SPUTNIK_HIP_GEMM(T, PA, PB, ACCS, LOW)                                                    \
  return wide ? launch_mfma_gemm<T, 256, AKM, BKM, PA, PB, ACCS, EPI>(m, n, k, replicas, outers,  \
                                                                      outer_is_split, a, b, out,  \
                                                                      LOW, stream)                \
              : launch_mfma_gemm<T, 128, AKM, BKM, PA, PB, ACCS, EPI>(m, n, k, replicas, outers,  \
                                                                      outer_is_split, a, b, out,  \
                                                                      LOW, stream)
  if (tile_type == SPUTNIK_HIP_F16) {
    if (pa == 1 && pb == 1) SPUTNIK_HIP_GEMM(_Float16, 1, 1, 1, 1.f);
    if (pa == 2 && pb == 1) SPUTNIK_HIP_GEMM(_Float16, 2, 1, 2, low);
    if (pa == 1 && pb == 2) SPUTNIK_HIP_GEMM(_Float16, 1, 2, 2, low);
    if (pa == 2 && pb == 2) SPUTNIK_HIP_GEMM(_Float16, 2, 2, 2, low);
  } else if (tile_type == SPUTNIK_HIP_BF16) {
    if (pa == 1 && pb == 1) SPUTNIK_HIP_GEMM(__bf16, 1, 1, 1, 1.f);
    if (pa == 3 && pb == 1) SPUTNIK_HIP_GEMM(__bf16, 3, 1, 1, 1.f);
    if (pa == 1 && pb == 3) SPUTNIK_HIP_GEMM(__bf16, 1, 3, 1, 1.f);
  }
#undef SPUTNIK_HIP_GEMM
  return SPUTNIK_HIP_UNSUPPORTED;
}

}  // namespace mfma_tiles
}  // namespace sputnik_hip
