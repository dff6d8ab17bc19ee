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
// What bounds it: a 128 x 128 x 64 step moves 32 KiB from L2 to LDS for 2.1 MFLOP -- 64
// flop per byte; the chip delivers ~18 TB/s into LDS (MI355X_MICROARCH.md, "Indexed rows"),
// i.e. ~1.15 PFLOP/s for this tile whatever the MFMA peak.  Measured 0.87 (config 5).
#pragma once

#include <type_traits>

#include "mfma_tiles.h"

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
};

// Per-lane source byte offsets of this wave's four copy pieces of one operand's tile, and
// its fragment read addresses inside a stage (relative to the operand's 16 KiB).
template <bool KM>
struct TileMap {
  // rows: extent of the r dimension (m or n); r0: first r of the tile
  static __device__ __forceinline__ void copy_offsets(unsigned (&off)[4], int wave, int lane, int r0,
                                                      int rows, int64_t ld) {
    if constexpr (!KM) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 8 * wave + 32 * j + (lane >> 3);
        const unsigned slot = static_cast<unsigned>((lane & 7) ^ ((row >> 1) & 7)) * 16u;
        // (rows beyond the matrix are clamped onto its last row: never stored / sampled)
        off[j] = static_cast<unsigned>(min(r0 + row, rows - 1) - r0) * static_cast<unsigned>(ld) * 2u + slot;
      }
    } else {
      const int last_chunk = (rows - r0) / 8 - 1;   // (rows is a multiple of 8)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int krow = 4 * (wave + 4 * j) + (lane >> 4);
        const int chunk = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | wave);
        off[j] = static_cast<unsigned>(krow) * static_cast<unsigned>(ld) * 2u +
                 static_cast<unsigned>(min(chunk, last_chunk)) * 16u;
      }
    }
  }
  // global address of the tile's first element at k-step offset k0 (elements)
  static __device__ __forceinline__ int64_t tile_origin(int r0, int64_t k0, int64_t ld) {
    return KM ? k0 * ld + r0 : static_cast<int64_t>(r0) * ld + k0;
  }
  // fragment addresses of the wave's two 32-row blocks at `r_in_tile` = w * 64:
  //   k-contiguous: addr[i][0] (k slot 0; slot 2 ks + (lane >> 5) is an XOR with ks * 32)
  //   k-major:      addr[i][t] for the two transposing reads (k 0-3, 4-7) at ks = 0; + ks * 4096
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
          addr[i][t] = static_cast<unsigned>(row * 256 + ((chunk ^ swz) * 16) + 8 * (p & 1));
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
          (__attribute__((address_space(3))) s4v*)(stage_operand + (addr[0] + ks * 4096u)));
      const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s4v*)(stage_operand + (addr[1] + ks * 4096u)));
      const s8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(frag, both);
    }
  }
};

template <typename T, bool AKM, bool BKM, int PA, int PB, int ACCS, int EPI>
__global__ __launch_bounds__(256, 2) void mfma_gemm_kernel(
    int m, int n, int k, int tiles_m, int tiles_n, int steps_per_replica, int total_steps,
    int outers, int outer_is_split, GemmOperand a_op, GemmOperand b_op, GemmOut out,
    float low_scale) {
  using H = Half8<T>;
  using frag = typename H::type;
  using P = Passes<PA, PB>;
  constexpr int NP = P::count();
  constexpr int kSmem = EPI == kSampled ? kLdsBytes : 2 * kStageBytes;
  __shared__ __attribute__((aligned(16))) char smem[kSmem];

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
  const int r0 = rt * kTile, c0 = ct * kTile;
  // the (replica, k step) pairs this workgroup reduces
  const int s_begin = outer_is_split ? static_cast<int>(static_cast<int64_t>(total_steps) * outer / outers)
                                     : outer * steps_per_replica;
  const int s_end = outer_is_split ? static_cast<int>(static_cast<int64_t>(total_steps) * (outer + 1) / outers)
                                   : (outer + 1) * steps_per_replica;

  unsigned a_off[4], b_off[4];
  TileMap<AKM>::copy_offsets(a_off, wave, lane, r0, m, a_op.ld);
  TileMap<BKM>::copy_offsets(b_off, wave, lane, c0, n, b_op.ld);
  const T* a_base = static_cast<const T*>(a_op.base);
  const T* b_base = static_cast<const T*>(b_op.base);
  auto stage = [&](int s, int pass, int buffer) {
    const int replica = s / steps_per_replica;
    const int64_t k0 = static_cast<int64_t>(s - replica * steps_per_replica) * kStep;
    const T* a = uniform_ptr(a_base + replica * a_op.replica_stride + P::a_of(pass) * a_op.plane_stride +
                             TileMap<AKM>::tile_origin(r0, k0, a_op.ld));
    const T* b = uniform_ptr(b_base + replica * b_op.replica_stride + P::b_of(pass) * b_op.plane_stride +
                             TileMap<BKM>::tile_origin(c0, k0, b_op.ld));
    const char* dst = smem + buffer * kStageBytes + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) copy_piece(a, a_off[j], dst + j * 4096);
#pragma unroll
    for (int j = 0; j < 4; ++j) copy_piece(b, b_off[j], dst + kOperandBytes + j * 4096);
  };

  unsigned fa[2][2], fb[2][2];
  TileMap<AKM>::fragment_addresses(fa, wr, lane);
  TileMap<BKM>::fragment_addresses(fb, wc, lane);

  f32x16 acc[ACCS][2][2];
#pragma unroll
  for (int z = 0; z < ACCS; ++z)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[z][i][j] = f32x16{};

  if (s_begin < s_end) {
    stage(s_begin, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  unsigned stage_base = 0;
  for (int s = s_begin; s < s_end; ++s) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int other = stage_base == 0 ? 1 : 0;
      if (pass + 1 < NP) {
        stage(s, pass + 1, other);
      } else if (s + 1 < s_end) {
        stage(s + 1, 0, other);
      }
      const char* sa = smem + stage_base;
      const char* sb = sa + kOperandBytes;
      frag a[2][4], b[2][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i][ks] = TileMap<AKM>::template read<frag>(sa, fa[i], ks);
          b[i][ks] = TileMap<BKM>::template read<frag>(sb, fb[i], ks);
        }
      constexpr int kOne = ACCS == 2 ? 1 : 0;
      const bool low = P::a_of(pass) + P::b_of(pass) > 0;   // (a compile-time constant once unrolled)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (low) acc[kOne][i][j] = H::mfma(a[i][ks], b[j][ks], acc[kOne][i][j]);
            else acc[0][i][j] = H::mfma(a[i][ks], b[j][ks], acc[0][i][j]);
          }
      // the next tiles have landed (this wave's copies), and every wave is done with these
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stage_base ^= static_cast<unsigned>(kStageBytes);
    }
  }

  auto value = [&](int i, int j, int reg) {
    return ACCS == 2 ? fmaf(acc[ACCS - 1][i][j][reg], low_scale, acc[0][i][j][reg]) : acc[0][i][j][reg];
  };

  if constexpr (EPI != kSampled) {
    // ---- the accumulators' 32 x 32 blocks (column = lane & 31: 128- / 64-byte runs) ----
    using TO = typename std::conditional<EPI == kDenseHalf, T, float>::type;
    TO* __restrict__ o = static_cast<TO*>(out.dense) + outer * out.outer_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < m) {
          const float bv = out.bias != nullptr ? out.bias[row] : 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = c0 + wc * 64 + j * 32 + (lane & 31);
            float v = value(i, j, reg) + bv;
            if (out.relu) v = fmaxf(v, 0.f);
            if (col < n) o[static_cast<int64_t>(row) * out.ld + col] = static_cast<TO>(v);
          }
        }
      }
  } else {
    // ---- the tile to LDS, then the mask rows' entries that fall into it ----
    float* tile_lds = reinterpret_cast<float*>(smem);
    int* bounds = reinterpret_cast<int*>(smem + kTileBytes);   // [kTile + 1]
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
      const int ok = threadIdx.x < kTile ? out.plan[min(r0 + static_cast<int>(threadIdx.x), m - 1)] : 1;
      by_table = __syncthreads_and(ok) != 0;
    } else {
      __syncthreads();
    }
    if (by_table) {
      const int* table = out.plan + plan_rows(m);
      const int group = threadIdx.x >> 4, l16 = threadIdx.x & 15;
      int from[8], to[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // (all bounds requested before the first is used)
        const int row = r0 + group + 16 * j;
        const int* run = table + static_cast<int64_t>(min(row, m - 1)) * (tiles_n + 1) + ct;
        from[j] = run[0];
        to[j] = row < m ? run[1] : run[0];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* tile_row = tile_lds + (group + 16 * j) * kPitch - c0;
        for (int p = from[j] + l16; p < to[j]; p += 16) o[p] = tile_row[column_indices[p]];
      }
      return;
    }
    // No plan, or a row whose columns do not ascend: the workgroup walks the CSR entries of
    // its 128 rows FLAT (they are contiguous in column_indices), sixteen bytes per lane,
    // and stores those whose column lies in the tile.
    if (threadIdx.x <= kTile)
      bounds[threadIdx.x] = out.row_offsets[min(r0 + static_cast<int>(threadIdx.x), m)];
    __syncthreads();
    const int first = bounds[0], last = bounds[kTile];
    // a lane takes four consecutive entries per round; its row moves forward only
    int row = 0;
    const int start = (first & ~3) + 4 * static_cast<int>(threadIdx.x);
    {  // first row whose end lies behind `start` (binary search over the 128 bounds)
      int lo = 0, hi = kTile;   // answer in [lo, hi]
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bounds[mid + 1] > start) hi = mid; else lo = mid + 1;
      }
      row = lo;
    }
    for (int p = start; p < last; p += 4 * 256) {
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
        while (row < kTile && bounds[row + 1] <= q) ++row;
        const unsigned c = static_cast<unsigned>(cols[e] - c0);
        if (q >= first && q < last && c < static_cast<unsigned>(kTile)) o[q] = tile_lds[row * kPitch + c];
      }
    }
  }
}

// Host side of a launch.  outers: independent outputs (dense) or workgroups that share a
// tile of ONE summed output (sampled, outer_is_split).
template <typename T, bool AKM, bool BKM, int PA, int PB, int ACCS, int EPI>
inline int launch_mfma_gemm(int m, int n, int k, int replicas, int outers, bool outer_is_split,
                            const GemmOperand& a, const GemmOperand& b, const GemmOut& out,
                            float low_scale, hipStream_t stream) {
  const int tiles_m = ceil_div(m, kTile), tiles_n = ceil_div(n, kTile);
  const int steps_per_replica = k / kStep;
  const int64_t total_steps = static_cast<int64_t>(replicas) * steps_per_replica;
  const int64_t blocks = static_cast<int64_t>(tiles_m) * tiles_n * outers;
  if (total_steps >= (int64_t{1} << 31) || blocks >= (int64_t{1} << 31) || outers < 1)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  hipLaunchKernelGGL((mfma_gemm_kernel<T, AKM, BKM, PA, PB, ACCS, EPI>),
                     dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, m, n, k, tiles_m,
                     tiles_n, steps_per_replica, static_cast<int>(total_steps), outers,
                     outer_is_split ? 1 : 0, a, b, out, low_scale);
  return launch_status();
}

// Dispatch over the plane counts a storage type has: float16 (1 | 2 planes, two tiles as
// soon as there is a low plane), bfloat16 (1 | 3 planes, one tile; 3 x 3 is not built).
// Returns SPUTNIK_HIP_UNSUPPORTED for a combination that is not instantiated.
template <bool AKM, bool BKM, int EPI>
inline int launch_mfma_gemm_typed(int tile_type, int pa, int pb, int m, int n, int k, int replicas,
                                  int outers, bool outer_is_split, const GemmOperand& a,
                                  const GemmOperand& b, const GemmOut& out, hipStream_t stream) {
  const float low = 1.f / kLowPlaneScale;
#define SPUTNIK_HIP_GEMM(T, PA, PB, ACCS, LOW)                                                   \
  return launch_mfma_gemm<T, AKM, BKM, PA, PB, ACCS, EPI>(m, n, k, replicas, outers,             \
                                                          outer_is_split, a, b, out, LOW, stream)
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
