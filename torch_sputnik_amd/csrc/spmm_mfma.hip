// left_spmm on HALF-storage operands through the matrix cores (round 5).
//
//   out_r[m, n] = A[m, k] * B_r[k, n]      A sparse (CSR, shared by the replicas), B_r dense
//
// i.e. sputnik::CudaSpmm in the host loop of src/left_replicated_spmm.cu:32-41 -- the forward
// pass and the input gradient of modules/sparse_linear.py:28,60-65.  For float32 operands
// that is the vector kernels of spmm_*.hip (the north star: no MFMA for the five float32
// operators).  float16 / bfloat16 dense operands are this library's extension; their
// products are exact in float32, and a layer weight at density 0.2 occupies every 128 x 64
// tile, so the product is a DENSE contraction with four zeros in five: the densified
// weight on v_mfma_f32_32x32x16_{f16,bf16} costs 1 / density times the sparse flops on a
// unit sixteen times faster than the packed-float32 vector pipe (config 5: 34.4 dense
// GFLOP per pass against 6.9 sparse ones at 40-50 TFLOP/s).
//
// Two launches per call (plus a memset):
//   1. densify: the CSR values scattered into a zeroed [m rounded up to 128, k] half image
//      (one wave per row).  float32 values are NOT rounded to the storage type: they leave
//      as half planes whose (scaled) sum is the value, as the float32 operand of the
//      weight gradient does (sddmm_mfma.hip, split_planes_kernel) -- float16: two planes,
//      the low one scaled by 2^11 and accumulated in a tile of its own; bfloat16: three.
//      A float32 DENSE operand (the incoming gradient in the backward pass) is split the
//      same way into the workspace.
//   2. the tile kernel: one workgroup (4 waves) = one 128 x 128 tile of one replica's
//      product, K walked in steps of 64 with one tile product per (A plane, B plane) pair
//      that matters (for two planes each: hi*hi, hi*lo, lo*hi -- lo*lo is 2^-22 of the
//      result).  A's tile as in sddmm_mfma.hip (k-contiguous rows of 128 B, XOR-swizzled at
//      the source, ds_read_b128 fragments).  B is [k][n], n contiguous -- the layout the
//      reference's left_spmm takes -- so its 64 x 128 tile lands as 256-byte rows (direct
//      global->LDS copies, 4 rows per wave instruction, chunk c of row r in slot
//      c ^ (((r & 3) << 2) | ((r >> 2) & 3))) and the MFMA's B fragment -- 8 consecutive k
//      of one column -- comes out of two ds_read_b64_tr_b16 (the hardware's 4 x 16
//      transposing read), conflict-free under that swizzle.  Double buffered, one
//      rendezvous per tile product.  The float32 tile leaves from the accumulators with
//      the bias / ReLU epilogue of sputnik_hip_spmm_bias_batched.
#include "mfma.h"
#include "mfma_tiles.h"
#include "options.h"

namespace sputnik_hip {
namespace {

using namespace mfma_tiles;

typedef short s4v __attribute__((__vector_size__(8)));
typedef short s8v __attribute__((__vector_size__(16)));

// The (A plane, B plane) pairs a step multiplies: every pair whose order a + b stays
// below the longer operand's plane count (one plane each: the one product).
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

// ACCS = 2 (float16 planes): products of order 1 (one low plane, kept scaled by 2^11)
// accumulate in a second tile that enters the result times low_scale = 2^-11.
template <typename T, int PA, int PB, int ACCS>
__global__ __launch_bounds__(256, 2) void spmm_mfma_kernel(
    int m, int n, int k, int steps, int tiles_m, int tiles_n, const T* __restrict__ a_planes,
    int64_t a_plane_stride, const T* __restrict__ dense, int64_t dense_stride,
    int64_t dense_plane_stride, const float* __restrict__ bias, int relu, float* __restrict__ out,
    int64_t out_stride, float low_scale) {
  using H = Half8<T>;
  using frag = typename H::type;
  using P = Passes<PA, PB>;
  constexpr int NP = P::count();
  __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (an SGPR: M0 takes it)
  const int wr = wave >> 1, wc = wave & 1;
  // consecutive work indices run behind one L2: the row tiles of one column tile (they
  // stage the same B panel), then the column tiles of one replica
  const int work = xcd_local_index32();
  const int rt = work % tiles_m;
  const int ct = (work / tiles_m) % tiles_n;
  const int replica = work / (tiles_m * tiles_n);
  const int r0 = rt * kTile, c0 = ct * kTile;

  // per-lane source offsets: A's four pieces (8 rows x 128 B each; the image has whole
  // tiles of rows), B's four pieces (4 rows x 256 B each; chunks beyond the matrix's last
  // column are clamped onto its last chunk: their products are never stored)
  unsigned a_off[4], b_off[4];
  const int last_chunk = (n - c0) / 8 - 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int arow = 8 * wave + 32 * j + (lane >> 3);
    a_off[j] = static_cast<unsigned>(arow) * static_cast<unsigned>(k) * 2u +
               static_cast<unsigned>((lane & 7) ^ ((arow >> 1) & 7)) * 16u;
    const int brow = 4 * (wave + 4 * j) + (lane >> 4);
    const int chunk = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | wave);
    b_off[j] = static_cast<unsigned>(brow) * static_cast<unsigned>(n) * 2u +
               static_cast<unsigned>(min(chunk, last_chunk)) * 16u;
  }
  const T* a_tile = uniform_ptr(a_planes + static_cast<int64_t>(r0) * k);
  const T* b_tile = uniform_ptr(dense + replica * dense_stride + c0);
  auto stage = [&](int s, int pass, int buffer) {
    const T* a = a_tile + P::a_of(pass) * a_plane_stride + s * kStep;
    const T* b = b_tile + P::b_of(pass) * dense_plane_stride + static_cast<int64_t>(s) * kStep * n;
    const char* dst = smem + buffer * kStageBytes + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) copy_piece(a, a_off[j], dst + j * 4096);
#pragma unroll
    for (int j = 0; j < 4; ++j) copy_piece(b, b_off[j], dst + kOperandBytes + j * 4096);
  };

  // A fragment addresses (k slot 0; slot 2 * ks + (lane >> 5) is an XOR with ks * 32)
  unsigned fa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wr * 64 + i * 32 + (lane & 31);
    fa[i] = static_cast<unsigned>(ra * 128 + (((lane >> 5) ^ ((ra >> 1) & 7)) * 16));
  }
  // B: transposing reads.  Lane 4q + p of a 16-lane group supplies row q, columns 4p .. 4p + 3
  // of a 4 x 16 block and receives column (lane & 15) of its four rows; block rows
  // 16 ks + 8 (lane >> 5) + 4 t + q (t = 0, 1: the fragment's k 0-3 and 4-7), block columns
  // wc * 64 + 32 j + 16 ((lane >> 4) & 1) + ...: fb[t][j] at ks = 0, + ks * 4096
  unsigned fb[2][2];
  {
    const int q = (lane & 15) >> 2, p = lane & 3, h = lane >> 5, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 8 * h + 4 * t + q;
        const int chunk = wc * 8 + j * 4 + 2 * g1 + (p >> 1);
        const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
        fb[t][j] = static_cast<unsigned>(kOperandBytes + row * 256 + ((chunk ^ swz) * 16) + 8 * (p & 1));
      }
  }

  f32x16 acc[ACCS][2][2];
#pragma unroll
  for (int z = 0; z < ACCS; ++z)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[z][i][j] = f32x16{};

  stage(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned stage_base = 0;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int other = stage_base == 0 ? 1 : 0;
      if (pass + 1 < NP) {
        stage(s, pass + 1, other);
      } else if (s + 1 < steps) {
        stage(s + 1, 0, other);
      }
      frag a[2][4], b[2][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i][ks] = *reinterpret_cast<const frag*>(smem + ((fa[i] ^ (ks * 32u)) + stage_base));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s4v*)(smem + (fb[0][j] + ks * 4096u + stage_base)));
          const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s4v*)(smem + (fb[1][j] + ks * 4096u + stage_base)));
          b[j][ks] = __builtin_bit_cast(frag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
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

  // ---- epilogue: the accumulators' 32 x 32 blocks (column = lane & 31: 128-byte runs) ----
  float* __restrict__ o = out + replica * out_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = r0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (row < m) {
        const float bv = bias != nullptr ? bias[row] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = c0 + wc * 64 + j * 32 + (lane & 31);
          float v = ACCS == 2 ? fmaf(acc[ACCS - 1][i][j][reg], low_scale, acc[0][i][j][reg])
                              : acc[0][i][j][reg];
          v += bv;
          if (relu) v = fmaxf(v, 0.f);
          if (col < n) o[static_cast<int64_t>(row) * n + col] = v;
        }
      }
    }
}

// The CSR values scattered into the zeroed dense image(s) [PLANES][rows][k]: one wave per
// row.  TV = float: split into planes as split_planes_kernel does (sddmm_mfma.hip); TV = T:
// as they are (PLANES = 1).  A column outside [0, k) is skipped.
template <typename T, typename TV, int PLANES>
__global__ __launch_bounds__(256) void densify_kernel(int m, int k, const int* __restrict__ row_offsets,
                                                      const int* __restrict__ column_indices,
                                                      const TV* __restrict__ values, T* __restrict__ image,
                                                      int64_t plane_stride, float scale1) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  T* dst = image + static_cast<int64_t>(row) * k;
  const int p1 = row_offsets[row + 1];
  for (int p = row_offsets[row] + lane; p < p1; p += 64) {
    const unsigned col = static_cast<unsigned>(column_indices[p]);
    if (col >= static_cast<unsigned>(k)) continue;
    float rest = static_cast<float>(values[p]);
#pragma unroll
    for (int z = 0; z < PLANES; ++z) {
      const float scaled = z == 1 ? rest * scale1 : rest;
      const T h = PLANES == 1 && sizeof(TV) == 2 ? static_cast<T>(values[p]) : static_cast<T>(scaled);
      dst[z * plane_stride + col] = h;
      rest -= z == 1 ? static_cast<float>(h) / scale1 : static_cast<float>(h);
    }
  }
}

int64_t padded_rows(int m) { return static_cast<int64_t>(ceil_div(m, kTile)) * kTile; }

int planes_of(int operand_type, int tile_type) {
  return operand_type == SPUTNIK_HIP_F32 ? sddmm_mfma_planes_of(tile_type) : 1;
}

int passes_of(int pa, int pb) {
  int c = 0;
  const int limit = (pa > pb ? pa : pb) - 1;
  for (int a = 0; a < pa; ++a)
    for (int b = 0; b < pb; ++b) c += a + b <= limit ? 1 : 0;
  return c;
}

}  // namespace

// tile_type: the half type of the tiles (a half operand's own type).
bool spmm_mfma_shape(int m, int k, int n, int nonzeros, int replicas, int values_type,
                     int dense_type, int tile_type) {
  const int forced = options().spmm_kernel;
  if (forced != 0 && forced != 4) return false;   // a vector kernel was asked for by name
  if (tile_type != SPUTNIK_HIP_F16 && tile_type != SPUTNIK_HIP_BF16) return false;
  if (values_type != SPUTNIK_HIP_F32 && values_type != tile_type) return false;
  if (dense_type != SPUTNIK_HIP_F32 && dense_type != tile_type) return false;
  if (m <= 0 || k <= 0 || n <= 0 || nonzeros <= 0 || replicas <= 0) return false;
  if (k % kStep != 0 || n % 8 != 0) return false;
  if (static_cast<int64_t>(m) * n >= (int64_t{1} << 31) || static_cast<int64_t>(k) * n >= (int64_t{1} << 30) ||
      padded_rows(m) * k >= (int64_t{1} << 31))
    return false;
  const int64_t tiles = static_cast<int64_t>(ceil_div(m, kTile)) * ceil_div(n, kTile) * replicas;
  if (tiles >= (int64_t{1} << 30)) return false;
  const int pa = planes_of(values_type, tile_type), pb = planes_of(dense_type, tile_type);
  if (tile_type == SPUTNIK_HIP_BF16 && pa > 1 && pb > 1) return false;   // six products per step
  if (forced == 4) return true;                       // "mfma": every shape the kernel serves
  // One tile product per pass costs 1 / density times the sampled flops, at several hundred
  // TFLOP/s against the vector kernels' 40-50: from density ~0.06 per pass; and the grid
  // should fill the chip (a tile per CU).
  const double density = static_cast<double>(nonzeros) / (static_cast<double>(m) * k);
  return m >= kTile && n >= kTile && tiles >= 192 && density >= 0.06 * passes_of(pa, pb);
}

size_t spmm_mfma_workspace_bytes(int m, int k, int n, int replicas, int values_type, int dense_type,
                                 int tile_type) {
  const int pa = planes_of(values_type, tile_type), pb = planes_of(dense_type, tile_type);
  size_t bytes = (static_cast<size_t>(pa) * padded_rows(m) * k * 2 + 255) / 256 * 256;
  if (dense_type == SPUTNIK_HIP_F32) bytes += static_cast<size_t>(pb) * replicas * k * n * 2;
  return bytes;
}

int spmm_mfma_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_offsets,
                     const int* column_indices, const void* values, int values_type,
                     const void* dense, int dense_type, int64_t dense_stride, int tile_type,
                     const float* bias, int relu, float* out, int64_t out_stride, void* workspace,
                     hipStream_t stream) {
  const int pa = planes_of(values_type, tile_type), pb = planes_of(dense_type, tile_type);
  const int64_t a_plane = padded_rows(m) * k;
  const size_t a_bytes = (static_cast<size_t>(pa) * a_plane * 2 + 255) / 256 * 256;
  hipError_t e = hipMemsetAsync(workspace, 0, static_cast<size_t>(pa) * a_plane * 2, stream);
  if (e != hipSuccess) return static_cast<int>(e);
  const dim3 rows_grid(ceil_div(m, 4));
#define SPUTNIK_HIP_DENSIFY(T, TV, PLANES, SCALE)                                                  \
  hipLaunchKernelGGL((densify_kernel<T, TV, PLANES>), rows_grid, dim3(256), 0, stream, m, k,       \
                     row_offsets, column_indices, static_cast<const TV*>(values),                  \
                     static_cast<T*>(workspace), a_plane, SCALE)
  if (tile_type == SPUTNIK_HIP_F16) {
    if (values_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_DENSIFY(_Float16, float, 2, kLowPlaneScale);
    else SPUTNIK_HIP_DENSIFY(_Float16, _Float16, 1, 1.f);
  } else {
    if (values_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_DENSIFY(__bf16, float, 3, 1.f);
    else SPUTNIK_HIP_DENSIFY(__bf16, __bf16, 1, 1.f);
  }
#undef SPUTNIK_HIP_DENSIFY
  int st = launch_status();
  if (st != 0) return st;
  const void* b = dense;
  int64_t b_stride = dense_stride, b_plane = 0;
  if (dense_type == SPUTNIK_HIP_F32) {
    // (the replicas of a float32 operand are split in one flat pass: back to back only)
    if (replicas > 1 && dense_stride != static_cast<int64_t>(k) * n) return SPUTNIK_HIP_UNSUPPORTED;
    char* planes = static_cast<char*>(workspace) + a_bytes;
    b_plane = static_cast<int64_t>(replicas) * k * n;
    st = sddmm_mfma_split_planes(b_plane, static_cast<const float*>(dense), tile_type, planes, stream);
    if (st != 0) return st;
    b = planes;
    b_stride = static_cast<int64_t>(k) * n;
  }
  const int tiles_m = ceil_div(m, kTile), tiles_n = ceil_div(n, kTile);
  const dim3 grid(static_cast<unsigned>(static_cast<int64_t>(tiles_m) * tiles_n * replicas));
  const bool two_tiles = tile_type == SPUTNIK_HIP_F16 && (pa > 1 || pb > 1);
  const float low_scale = two_tiles ? 1.f / kLowPlaneScale : 1.f;
#define SPUTNIK_HIP_MF(T, PA, PB, ACCS)                                                            \
  hipLaunchKernelGGL((spmm_mfma_kernel<T, PA, PB, ACCS>), grid, dim3(256), 0, stream, m, n, k,     \
                     k / kStep, tiles_m, tiles_n, static_cast<const T*>(workspace), a_plane,       \
                     static_cast<const T*>(b), b_stride, b_plane, bias, relu, out, out_stride,     \
                     low_scale)
  if (tile_type == SPUTNIK_HIP_F16) {
    if (pa == 1 && pb == 1) SPUTNIK_HIP_MF(_Float16, 1, 1, 1);
    else if (pa == 2 && pb == 1) SPUTNIK_HIP_MF(_Float16, 2, 1, 2);
    else if (pa == 1 && pb == 2) SPUTNIK_HIP_MF(_Float16, 1, 2, 2);
    else SPUTNIK_HIP_MF(_Float16, 2, 2, 2);
  } else {
    if (pa == 1 && pb == 1) SPUTNIK_HIP_MF(__bf16, 1, 1, 1);
    else if (pa == 3 && pb == 1) SPUTNIK_HIP_MF(__bf16, 3, 1, 1);
    else if (pa == 1 && pb == 3) SPUTNIK_HIP_MF(__bf16, 1, 3, 1);
    else return SPUTNIK_HIP_UNSUPPORTED;
  }
#undef SPUTNIK_HIP_MF
  return launch_status();
}

}  // namespace sputnik_hip
