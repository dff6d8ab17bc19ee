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
// Two launches per call:
//   1. densify: the [m rounded up to 128, k] half image of the CSR matrix (one wave per row
//      assembles the row in LDS and writes it whole: no memset, no 2-byte stores).  float32 values are NOT rounded to the storage type: they leave
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
#include "mfma_gemm.h"
#include "options.h"

namespace sputnik_hip {
namespace {

using namespace mfma_tiles;

// The weight's dense image(s) [PLANES][rows][k] of the tile type from its CSR form: one wave
// per row assembles the row in LDS (zeros, then the row's entries scattered in) and writes
// it out whole in 16-byte pieces -- every byte of the image is written exactly once, no
// memset pass, no 2-byte stores to memory (k a multiple of 8; rows beyond m come out zero;
// k beyond 2048 in segments of 2048 columns, each walking the row's entries again).
// TV = float: split into planes as split_planes_kernel does (sddmm_mfma.hip); TV = T: as
// they are (PLANES = 1).  A column outside [0, k) is skipped.
template <typename T, typename TV, int PLANES>
__global__ __launch_bounds__(256) void densify_kernel(int m, int rows, int k,
                                                      const int* __restrict__ row_offsets,
                                                      const int* __restrict__ column_indices,
                                                      const TV* __restrict__ values, T* __restrict__ image,
                                                      int64_t plane_stride, float scale1) {
  constexpr int kSeg = 2048;
  __shared__ __attribute__((aligned(16))) T segment[4][PLANES][kSeg];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int p0 = row < m ? row_offsets[row] : 0, p1 = row < m ? row_offsets[row + 1] : 0;
  T* dst = image + static_cast<int64_t>(row) * k;
  for (int c0 = 0; c0 < k; c0 += kSeg) {
    const int width = min(kSeg, k - c0);
    // (a wave's LDS operations execute in order: zeros, then the entries, then the reads)
    for (int i = lane * 8; i < width; i += 512)
#pragma unroll
      for (int z = 0; z < PLANES; ++z) *reinterpret_cast<uint4*>(&segment[wave][z][i]) = uint4{0, 0, 0, 0};
    for (int p = p0 + lane; p < p1; p += 64) {
      const unsigned col = static_cast<unsigned>(column_indices[p] - c0);
      if (col >= static_cast<unsigned>(width)) continue;
      float rest = static_cast<float>(values[p]);
#pragma unroll
      for (int z = 0; z < PLANES; ++z) {
        const float scaled = z == 1 ? rest * scale1 : rest;
        const T h = PLANES == 1 && sizeof(TV) == 2 ? static_cast<T>(values[p]) : static_cast<T>(scaled);
        segment[wave][z][col] = h;
        rest -= z == 1 ? static_cast<float>(h) / scale1 : static_cast<float>(h);
      }
    }
    for (int i = lane * 8; i < width; i += 512)
#pragma unroll
      for (int z = 0; z < PLANES; ++z)
        *reinterpret_cast<uint4*>(dst + z * plane_stride + c0 + i) =
            *reinterpret_cast<const uint4*>(&segment[wave][z][i]);
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

// The image [planes][rows_padded][k] of the tile type (one launch: every byte written once).
int densify_into(int m, int k, const int* row_offsets, const int* column_indices, const void* values,
                 int values_type, int tile_type, void* image, int64_t rows_padded, hipStream_t stream) {
  if (k % 8 != 0 || !aligned_to(image, 16)) return SPUTNIK_HIP_INVALID_ARGUMENT;
  const int64_t a_plane = rows_padded * k;
  const dim3 rows_grid(static_cast<unsigned>(ceil_div64(rows_padded, 4)));
#define SPUTNIK_HIP_DENSIFY(T, TV, PLANES, SCALE)                                                  \
  hipLaunchKernelGGL((densify_kernel<T, TV, PLANES>), rows_grid, dim3(256), 0, stream, m,          \
                     static_cast<int>(rows_padded), k, row_offsets, column_indices,                \
                     static_cast<const TV*>(values), static_cast<T*>(image), a_plane, SCALE)
  if (tile_type == SPUTNIK_HIP_F16) {
    if (values_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_DENSIFY(_Float16, float, 2, kLowPlaneScale);
    else SPUTNIK_HIP_DENSIFY(_Float16, _Float16, 1, 1.f);
  } else {
    if (values_type == SPUTNIK_HIP_F32) SPUTNIK_HIP_DENSIFY(__bf16, float, 3, 1.f);
    else SPUTNIK_HIP_DENSIFY(__bf16, __bf16, 1, 1.f);
  }
#undef SPUTNIK_HIP_DENSIFY
  return launch_status();
}

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
  int st = densify_into(m, k, row_offsets, column_indices, values, values_type, tile_type, workspace,
                        padded_rows(m), stream);
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
  const GemmOperand a{workspace, k, 0, a_plane};
  const GemmOperand bop{b, n, b_stride, b_plane};
  GemmOut o{};
  o.dense = out;
  o.ld = n;
  o.outer_stride = out_stride;
  o.bias = bias;
  o.relu = relu;
  // the densified weight is k-contiguous, the dense operand [k][n] k-major (the layout
  // src/left_replicated_spmm.cu:36 takes): its fragments are transposing reads
  return launch_mfma_gemm_typed<false, true, kDense>(tile_type, pa, pb, m, n, k, replicas, replicas,
                                                     /*outer_is_split=*/false, a, bop, o, stream,
                                                     wide_tile(m, n, replicas));
}

}  // namespace sputnik_hip
