// Summed SDDMM on HALF-storage operands through the matrix cores (round 5).
//
//   out[p] = sum_r < lhs_r[i_p, 0:k], rhs_r[j_p, 0:k] >        for every stored (i_p, j_p)
//
// i.e. the gradient of sparse weights shared by a batch: the reference computes the
// [R, nnz] products with sputnik::CudaSddmm in a host loop (src/sddmm_cuda.cu:45-54, called
// from modules/sparse_linear.py:44-49) and lets autograd add them up.  For float32
// operands that is the vector kernel of sddmm_tiled.hip (the north star: no MFMA for the
// five float32 operators).  float16 / bfloat16 operands are this library's extension, their
// products are exact in float32, and with a reduction of R * k elements over a mask that
// occupies every 128 x 128 tile (density 0.2 at config 5) the product is a sampled DENSE
// contraction: the dense tile on v_mfma_f32_32x32x16_{f16,bf16} costs 1 / density times the
// sparse flops on a unit sixteen times faster than the packed-float32 vector pipe.
//
// One workgroup (4 waves) = one 128 x 128 tile of lhs * rhs^T over a contiguous range of
// the (replica, 64-element k step) pairs:
//   * both operands are k-contiguous, so a step's 128 x 64 tiles go to LDS by direct
//     global->LDS copies (1 KiB = 8 rows x 128 B per wave instruction), double buffered,
//     one rendezvous per step: the copies of step s + 1 fly under the MFMAs of step s;
//   * the LDS image is lane-linear (the copy's rule), the XOR swizzle that keeps the
//     fragment reads off each other's banks sits in the per-lane SOURCE address: 16-byte
//     slot g of row r lands in slot g ^ ((r >> 1) & 7) (ds_read_b128 serves 16 lanes at a
//     time -- rows {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of a fragment -- and with
//     128-byte rows two rows share a bank row: the 16 lanes hit 16 different slots);
//   * a wave owns a 64 x 64 quarter as 2 x 2 accumulators of 32 x 32 (64 registers);
//     per step 16 ds_read_b128 and 16 MFMAs;
//   * epilogue: the float tile goes to LDS (it reuses the stages), and the workgroup walks
//     the CSR entries of its 128 rows FLAT -- they are contiguous in column_indices --
//     sixteen bytes per lane, storing those whose column lies in the tile.  No table, no
//     pre-pass, no assumption on the order of a row's columns.
// `splits` workgroups share a tile (each writes its own partial vector, added in index
// order by sum_partials_kernel of sddmm.hip: deterministic), so that a 2048 x 2048 mask
// -- 256 tiles -- puts two workgroups on every CU.
//
// Bytes and flops per launch (config 5: 2048^2 mask at density 0.2, k = 512, 8 replicas):
// dense 2 * 2048^2 * 4096 = 34.4 GFLOP for 6.9 GFLOP of sampled products; operands
// 2 * 16.8 MB once from memory, 2048 / 128 = 16 times through L2 (one 128-row panel per
// tile row / tile column).
#include "mfma.h"
#include "mfma_gemm.h"
#include "options.h"

namespace sputnik_hip {
namespace {

using namespace mfma_tiles;

// float32 -> PLANES planes of the half type T whose sum is the value: p0 = round(v), p1 =
// round((v - p0) * scale1), p2 = round(v - p0 - p1) -- the products p * x are exact in
// float32, so a float32 operand (the incoming gradient of modules/sparse_linear.py:44-49,
// which no storage type rounds) keeps its bits on the matrix cores at PLANES times the tiles.
//   float16: 2 planes, the low one scaled by 2^11 (scale1; the kernel accumulates it in a
//     tile of its own and adds it times 2^-11): 22 bits of every value down to ~1e-6 --
//     unscaled, the low plane of a value below 0.1 would sit in float16's subnormals;
//   bfloat16: 3 planes as they are (float32's exponent range): 24 bits.
template <typename T, int PLANES>
__global__ __launch_bounds__(256) void split_planes_kernel(int64_t quads /* of 4 elements */,
                                                           const float* __restrict__ in,
                                                           T* __restrict__ out, int64_t plane_stride,
                                                           float scale1) {
  using T4 = T __attribute__((ext_vector_type(4)));
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < quads;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const float4 v4 = reinterpret_cast<const float4*>(in)[i];
    float rest[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
      T4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float scaled = p == 1 ? rest[e] * scale1 : rest[e];
        h[e] = static_cast<T>(scaled);
        rest[e] -= p == 1 ? static_cast<float>(h[e]) / scale1 : static_cast<float>(h[e]);
      }
      reinterpret_cast<T4*>(out + p * plane_stride)[i] = h;
    }
  }
}

// The plan: per row, where its entry stream crosses the boundaries of the 128-column
// tiles -- table[row][c] = first entry with column >= 128 c, c = 0 .. tiles_n: a tile's run
// of the row is [table[row][ct], table[row][ct + 1]) -- and whether the row's
// columns ascend (row_ok; every row writes its own word, nothing is initialised).  One
// wave per row, topology only.
__global__ __launch_bounds__(256) void sddmm_mfma_plan_kernel(int m, int n, int tiles_n,
                                                              const int* __restrict__ row_offsets,
                                                              const int* __restrict__ column_indices,
                                                              int* __restrict__ plan) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  int* table = plan + plan_rows(m) + static_cast<int64_t>(row) * (tiles_n + 1);
  const int p0 = row_offsets[row], p1 = row_offsets[row + 1];
  bool ok = true;
  for (int base = p0; base < p1; base += 64) {
    const int p = base + lane;
    if (p < p1) {
      const int cur = column_indices[p];
      const int prev = p > p0 ? column_indices[p - 1] : -1;
      if (cur <= prev || cur >= n) {
        ok = false;
      } else {
        const int tc = cur / kTile, tp = prev < 0 ? -1 : prev / kTile;
        for (int c = tp + 1; c <= tc; ++c) table[c] = p;
      }
    }
  }
  // the tile columns behind the row's last entry (all of them for an empty row)
  const int last_tile = p1 > p0 ? min(max(column_indices[p1 - 1], 0), n - 1) / kTile : -1;
  for (int c = last_tile + 1 + lane; c <= tiles_n; c += 64) table[c] = p1;
  const bool wave_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
  if (lane == 0) plan[row] = wave_ok ? 1 : 0;
}

}  // namespace

bool sddmm_mfma_shape(int m, int k, int n, int nonzeros, int replicas) {
  const int forced = options().sddmm_kernel;
  if (forced == 1 || forced == 2) return false;   // "tiled" / "wave": the vector kernels
  if (k <= 0 || k % kStep != 0 || m <= 0 || n <= 0 || nonzeros <= 0 || replicas <= 0) return false;
  if (static_cast<int64_t>(k) * 2 * kTile >= (int64_t{1} << 31)) return false;
  const int64_t tiles = static_cast<int64_t>(ceil_div(m, kTile)) * ceil_div(n, kTile);
  if (tiles * 64 > (int64_t{1} << 30)) return false;
  if (forced == 3) return true;                   // "mfma": every shape the kernel serves
  // The dense tiles cost 1 / density times the sampled flops; the vector kernels run the
  // sampled flops at 25-27 TFLOP/s on half operands, the tiles at several hundred.
  const double density = static_cast<double>(nonzeros) / (static_cast<double>(m) * n);
  const int64_t reduction = static_cast<int64_t>(replicas) * k;
  return m >= kTile && n >= kTile && density >= 0.05 && reduction >= 1024;
}

bool sddmm_mfma_applicable(int m, int k, int n, int nonzeros, int replicas, const void* lhs,
                           int64_t lhs_stride, const void* rhs, int64_t rhs_stride) {
  return sddmm_mfma_shape(m, k, n, nonzeros, replicas) && aligned_to(lhs, 16) &&
         aligned_to(rhs, 16) && lhs_stride % 8 == 0 && rhs_stride % 8 == 0;
}

int sddmm_mfma_splits(int m, int k, int n, int replicas, int planes) {
  const bool wide = wide_tile(m, n, 0);
  const int64_t tiles = static_cast<int64_t>(ceil_div(m, wide ? 256 : kTile)) * ceil_div(n, kTile);
  const int64_t steps = static_cast<int64_t>(replicas) * (k / kStep);
  // 128-row tiles: two workgroups per CU (68 KiB of LDS each); 256-row tiles: one; at
  // least eight tile products per workgroup
  int64_t splits = ceil_div64(wide ? 256 : 512, tiles);
  if (splits > steps * planes / 8) splits = steps * planes / 8;
  if (splits > steps) splits = steps;
  if (splits > 8) splits = 8;
  return splits < 1 ? 1 : static_cast<int>(splits);
}

size_t sddmm_mfma_plan_bytes(int m, int n) {
  return sizeof(int) * static_cast<size_t>(plan_rows(m) + static_cast<int64_t>(m) * (ceil_div(n, kTile) + 1));
}

int sddmm_mfma_plan(int m, int n, const int* row_offsets, const int* column_indices, void* plan,
                    hipStream_t stream) {
  hipLaunchKernelGGL(sddmm_mfma_plan_kernel, dim3(ceil_div(m, 4)), dim3(256), 0, stream, m, n,
                     ceil_div(n, kTile), row_offsets, column_indices, static_cast<int*>(plan));
  return launch_status();
}

int sddmm_mfma_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_offsets,
                      const int* column_indices, const void* lhs, int64_t lhs_stride,
                      const void* rhs, int64_t rhs_stride, int in_type, float* partials,
                      int splits, const void* plan, hipStream_t stream, int planes,
                      int64_t lhs_plane_stride, int64_t rhs_plane_stride) {
  if (planes < 1 || planes > 3 || splits < 1) return SPUTNIK_HIP_INVALID_ARGUMENT;
  // (the planes belong to the operand whose plane stride is given)
  const int pa = rhs_plane_stride == 0 ? planes : 1, pb = rhs_plane_stride == 0 ? 1 : planes;
  const GemmOperand a{lhs, k, lhs_stride, lhs_plane_stride};
  const GemmOperand b{rhs, k, rhs_stride, rhs_plane_stride};
  GemmOut out{};
  out.sampled = partials;
  out.row_offsets = row_offsets;
  out.column_indices = column_indices;
  out.plan = static_cast<const int*>(plan);
  out.nonzeros = nonzeros;
  out.vector_columns = aligned_to(column_indices, 16) ? 1 : 0;
  // lhs [m, k] and rhs [n, k] are both k-contiguous (src/sddmm_cuda.cu:48-53's layout)
  return launch_mfma_gemm_typed<false, false, kSampled>(in_type, pa, pb, m, n, k, replicas, splits,
                                                        /*outer_is_split=*/true, a, b, out, stream,
                                                        wide_tile(m, n, 0));
}

int sddmm_mfma_planes_of(int half_type) { return half_type == SPUTNIK_HIP_BF16 ? 3 : 2; }

int sddmm_mfma_split_planes(int64_t count, const float* in, int half_type, void* planes,
                            hipStream_t stream) {
  if (count % 4 != 0 || !aligned_to(in, 16) || !aligned_to(planes, 8))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (count == 0) return 0;
  const int64_t quads = count / 4;
  const int64_t want = ceil_div64(quads, 256);
  const unsigned blocks = static_cast<unsigned>(want < 8192 ? want : 8192);
  if (half_type == SPUTNIK_HIP_F16) {
    hipLaunchKernelGGL((split_planes_kernel<_Float16, 2>), dim3(blocks), dim3(256), 0, stream, quads,
                       in, static_cast<_Float16*>(planes), count, kLowPlaneScale);
  } else if (half_type == SPUTNIK_HIP_BF16) {
    hipLaunchKernelGGL((split_planes_kernel<__bf16, 3>), dim3(blocks), dim3(256), 0, stream, quads, in,
                       static_cast<__bf16*>(planes), count, 1.f);
  } else {
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  }
  return launch_status();
}

}  // namespace sputnik_hip
