// SpMM  C[m,n] = A_csr[m,k] * B[k,n]  for gfx950 -- dispatcher + the
// workspace-free "row gather" kernel.
//
// Replaces sputnik::CudaSpmm as driven by the reference's host wrappers
// (src/spmm_cuda.cu:48-57, src/left_replicated_spmm.cu:32-41).
//
// Row-gather kernel: 1-D row splitting.  A group of LPR lanes owns one output
// row strip of LPR*VEC columns; groups take rows in `row_indices` order so a
// 256-thread workgroup holds rows of similar length.  The group streams its
// row's (column, value) pairs with coalesced loads, broadcasts one pair at a
// time (v_readlane for full-wave groups) and gathers the matching row of B
// with one VEC-wide load per lane: every B read is a contiguous
// LPR*VEC*4-byte segment.  It needs no workspace and places no requirement
// on the order of column indices inside a row.
#include "common.h"
#include "options.h"
#include "wave_utils.h"

namespace sputnik_hip {

int spmm_tiled_plan(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                    const int* row_offsets, const int* column_indices, void* workspace,
                    size_t workspace_bytes, hipStream_t stream, bool* planned);
int spmm_tiled_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                    const float* values, int64_t values_stride, const int* row_offsets,
                    const int* column_indices, const float* dense, int64_t dense_stride,
                    float* out, int64_t out_stride, const void* workspace,
                    size_t workspace_bytes, hipStream_t stream, Epilogue epi, bool* handled);
size_t spmm_tiled_workspace_bytes(int m, int k, int n, int nonzeros);
int spmm_tiled_choice(int m, int k, int n, int nonzeros, int replicas);
const char* spmm_tiled_kernel_name(int m, int k, int n, int nonzeros, int replicas);
bool spmm_panel_applicable(int m, int k, int n, int nonzeros, const float* dense,
                           int64_t dense_stride, const float* out, int64_t out_stride);
struct GroupProblemHost {
  const int* row_indices;
  const int* row_offsets;
  const int* column_indices;
  const float* values;
  const int* value_permutation;
  const float* dense;
  float* out;
  int nonzeros;
};
bool spmm_panel_group_supported(int m, int k, int n, int count, int block_rows, bool accumulate);
int spmm_panel_group_launch(int m, int k, int n, int replicas, int count,
                            const GroupProblemHost* problems, int64_t dense_stride,
                            int64_t out_stride, int block_rows, bool accumulate,
                            hipStream_t stream);

int spmm_panel_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                      const float* values, int64_t values_stride, const int* row_offsets,
                      const int* column_indices, const float* dense, int64_t dense_stride,
                      float* out, int64_t out_stride, hipStream_t stream, Epilogue epi,
                      const int* value_permutation = nullptr, int block_rows = 0,
                      int mask_heads = 0);

namespace {

constexpr int kBlock = 256;

template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void spmm_rowgather_kernel(
    int m, int n, const int* __restrict__ row_indices, const float* __restrict__ values,
    int64_t values_stride, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, const float* __restrict__ dense,
    int64_t dense_stride, float* __restrict__ out, int64_t out_stride, Epilogue epi) {
  constexpr int kRowsPerBlock = kBlock / LPR;
  const int sub = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const int slot = blockIdx.x * kRowsPerBlock + sub;
  const int replica = blockIdx.z;
  const int c0 = (blockIdx.y * LPR + l) * VEC;

  values += replica * values_stride;
  dense += replica * dense_stride;
  out += replica * out_stride;

  const bool row_ok = slot < m;
  const bool col_ok = c0 < n;  // n % VEC == 0, so the whole vector is in range
  const int row = row_ok ? row_indices[slot] : 0;
  int p = row_ok ? row_offsets[row] : 0;
  const int p_end = row_ok ? row_offsets[row + 1] : 0;

  const float* __restrict__ b_col = dense + (col_ok ? c0 : 0);
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

  for (; p < p_end; p += LPR) {
    const int q = p + l;
    int j = 0;
    float a = 0.f;
    if (q < p_end) {
      j = column_indices[q];
      a = values[q];
    }
    const int cnt = min(LPR, p_end - p);
#pragma unroll 4
    for (int t = 0; t < cnt; ++t) {
      const int jj = group_broadcast<LPR>(j, t);
      const float aa = group_broadcast<LPR>(a, t);
      float b[VEC];
      load_vec<VEC>(b, b_col + static_cast<int64_t>(jj) * n);
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = fmaf(aa, b[v], acc[v]);
    }
  }

  if (row_ok && col_ok) {
    if (epi.bias != nullptr || epi.relu) {
      const float b = epi.bias != nullptr ? epi.bias[row] : 0.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = epilogue_scalar(acc[v], b, epi.relu);
    }
    store_vec<VEC>(out + static_cast<int64_t>(row) * n + c0, acc);
  }
}

template <int VEC, int LPR>
int launch_rowgather(int m, int n, int replicas, const int* row_indices, const float* values,
                     int64_t values_stride, const int* row_offsets, const int* column_indices,
                     const float* dense, int64_t dense_stride, float* out, int64_t out_stride,
                     hipStream_t stream, Epilogue epi) {
  constexpr int kRowsPerBlock = kBlock / LPR;
  const int gx = ceil_div(m, kRowsPerBlock);
  const int gy = ceil_div(n, LPR * VEC);
  if (gy > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int rz = min(replicas - r0, kMaxGridYZ);
    hipLaunchKernelGGL((spmm_rowgather_kernel<VEC, LPR>), dim3(gx, gy, rz), dim3(kBlock), 0,
                       stream, m, n, row_indices, values + r0 * values_stride, values_stride,
                       row_offsets, column_indices, dense + r0 * dense_stride, dense_stride,
                       out + r0 * out_stride, out_stride, epi);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

template <int VEC>
int launch_rowgather_vec(int m, int n, int replicas, const int* row_indices, const float* values,
                         int64_t values_stride, const int* row_offsets,
                         const int* column_indices, const float* dense, int64_t dense_stride,
                         float* out, int64_t out_stride, hipStream_t stream, Epilogue epi) {
  const int lanes_needed = ceil_div(n, VEC);
#define SPUTNIK_HIP_RG(LPR)                                                                  \
  return launch_rowgather<VEC, LPR>(m, n, replicas, row_indices, values, values_stride,      \
                                    row_offsets, column_indices, dense, dense_stride, out,   \
                                    out_stride, stream, epi)
  if (lanes_needed <= 8) SPUTNIK_HIP_RG(8);
  if (lanes_needed <= 16) SPUTNIK_HIP_RG(16);
  if (lanes_needed <= 32) SPUTNIK_HIP_RG(32);
  SPUTNIK_HIP_RG(64);
#undef SPUTNIK_HIP_RG
}

}  // namespace

int spmm_rowgather_launch(int m, int n, int replicas, const int* row_indices,
                          const float* values, int64_t values_stride, const int* row_offsets,
                          const int* column_indices, const float* dense, int64_t dense_stride,
                          float* out, int64_t out_stride, hipStream_t stream, Epilogue epi) {
  int vec = vector_width(dense, n, dense_stride);
  vec = min(vec, vector_width(out, n, out_stride));
  switch (vec) {
    case 4:
      return launch_rowgather_vec<4>(m, n, replicas, row_indices, values, values_stride,
                                     row_offsets, column_indices, dense, dense_stride, out,
                                     out_stride, stream, epi);
    case 2:
      return launch_rowgather_vec<2>(m, n, replicas, row_indices, values, values_stride,
                                     row_offsets, column_indices, dense, dense_stride, out,
                                     out_stride, stream, epi);
    default:
      return launch_rowgather_vec<1>(m, n, replicas, row_indices, values, values_stride,
                                     row_offsets, column_indices, dense, dense_stride, out,
                                     out_stride, stream, epi);
  }
}

}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

size_t sputnik_hip_spmm_workspace_bytes(int m, int k, int n, int nonzeros) {
  return spmm_tiled_workspace_bytes(m, k, n, nonzeros);
}

int sputnik_hip_spmm_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                          const int* row_offsets, const int* column_indices, void* workspace,
                          size_t workspace_bytes, sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || n == 0) return 0;
  bool planned = false;  // replica count not known yet: plan for either tiled kernel
  return spmm_tiled_plan(m, k, n, nonzeros, -1, row_indices, row_offsets, column_indices,
                         workspace, workspace_bytes, stream, &planned);
}

namespace {

// The panel-resident kernel (spmm_panel.hip, k <= 512: B panel copied to LDS once,
// no pre-pass) is taken where the chunked kernels would be mostly skeleton: when
// they would pick the 64-column kernel, or nothing although the call is not
// tiny.  Wide shapes (many 256 / 512-column tiles) stay with the wide kernels,
// whose inner step does twice the FMAs per broadcast.  Knob: "panel" forces it.
bool takes_panel(int m, int k, int n, int nonzeros, int replicas, const float* dense,
                 int64_t dense_stride, const float* out, int64_t out_stride) {
  const int forced = options().spmm_kernel;
  if (forced != 0 && forced != 3) return false;
  if (nonzeros == 0 ||
      !spmm_panel_applicable(m, k, n, nonzeros, dense, dense_stride, out, out_stride))
    return false;
  if (forced == 3) return true;
  // More than one panel (k > 512): every pass walks (part of) the rows' streams
  // again, which pays for two panels (1024^2 x 64 x 64 replicas: density 0.1 36 vs
  // 47 us, 0.3 84 vs 90 us; four panels, 2048^2: 83 vs 60 us).
  if (k > 1024 || (k > 512 && nonzeros > 320 * static_cast<int64_t>(m))) return false;
  // (two panels on a grid that does not fill the chip: 2048 x 1024 x 1024, one
  // replica, 128 workgroups: 33.0 vs 30.9 us for the chunked kernel)
  if (k > 512 && static_cast<int64_t>((m + 255) / 256) * ((n + 63) / 64) * replicas < 192)
    return false;
  const int choice = spmm_tiled_choice(m, k, n, nonzeros, replicas);
  const int64_t work = static_cast<int64_t>(nonzeros) * n * replicas;
  return choice == 2 || choice == 3 || (choice == 0 && work >= (int64_t{1} << 24));
}

int spmm_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
              const float* values, int64_t values_stride, const int* row_offsets,
              const int* column_indices, const float* dense, int64_t dense_stride, float* out,
              int64_t out_stride, const void* workspace, size_t workspace_bytes,
              hipStream_t stream, Epilogue epi) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || n == 0 || replicas == 0) return 0;
  if (takes_panel(m, k, n, nonzeros, replicas, dense, dense_stride, out, out_stride))
    return spmm_panel_launch(m, k, n, nonzeros, replicas, row_indices, values, values_stride,
                             row_offsets, column_indices, dense, dense_stride, out, out_stride,
                             stream, epi);
  bool handled = false;
  const int st = spmm_tiled_exec(m, k, n, nonzeros, replicas, row_indices, values, values_stride,
                                 row_offsets, column_indices, dense, dense_stride, out,
                                 out_stride, workspace, workspace_bytes, stream, epi, &handled);
  if (st != 0 || handled) return st;
  return spmm_rowgather_launch(m, n, replicas, row_indices, values, values_stride, row_offsets,
                               column_indices, dense, dense_stride, out, out_stride, stream, epi);
}

}  // namespace

int sputnik_hip_spmm_batched_planned(int m, int k, int n, int nonzeros, int replicas,
                                     const int* row_indices, const float* values,
                                     int64_t values_stride, const int* row_offsets,
                                     const int* column_indices, const float* dense,
                                     int64_t dense_stride, float* out, int64_t out_stride,
                                     const void* workspace, size_t workspace_bytes,
                                     sputnik_hip_stream_t stream) {
  return spmm_exec(m, k, n, nonzeros, replicas, row_indices, values, values_stride, row_offsets,
                   column_indices, dense, dense_stride, out, out_stride, workspace,
                   workspace_bytes, stream, Epilogue{});
}

int sputnik_hip_spmm_batched(int m, int k, int n, int nonzeros, int replicas,
                             const int* row_indices, const float* values,
                             int64_t values_stride, const int* row_offsets,
                             const int* column_indices, const float* dense,
                             int64_t dense_stride, float* out, int64_t out_stride,
                             void* workspace, size_t workspace_bytes,
                             sputnik_hip_stream_t stream) {
  return sputnik_hip_spmm_bias_batched(m, k, n, nonzeros, replicas, row_indices, values,
                                       values_stride, row_offsets, column_indices, dense,
                                       dense_stride, nullptr, 0, out, out_stride, workspace,
                                       workspace_bytes, stream);
}

int sputnik_hip_spmm_bias_batched(int m, int k, int n, int nonzeros, int replicas,
                                  const int* row_indices, const float* values,
                                  int64_t values_stride, const int* row_offsets,
                                  const int* column_indices, const float* dense,
                                  int64_t dense_stride, const float* bias, int relu, float* out,
                                  int64_t out_stride, void* workspace, size_t workspace_bytes,
                                  sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || n == 0 || replicas == 0) return 0;
  Epilogue epi;
  epi.bias = bias;
  epi.relu = relu != 0;
  if (takes_panel(m, k, n, nonzeros, replicas, dense, dense_stride, out, out_stride))
    return spmm_panel_launch(m, k, n, nonzeros, replicas, row_indices, values, values_stride,
                             row_offsets, column_indices, dense, dense_stride, out, out_stride,
                             stream, epi);   // no pre-pass
  bool planned = false;
  const int st = spmm_tiled_plan(m, k, n, nonzeros, replicas, row_indices, row_offsets,
                                 column_indices, workspace, workspace_bytes, stream, &planned);
  if (st != 0) return st;
  return spmm_exec(m, k, n, nonzeros, replicas, row_indices, values, values_stride, row_offsets,
                   column_indices, dense, dense_stride, out, out_stride, workspace,
                   workspace_bytes, stream, epi);
}

const char* sputnik_hip_spmm_kernel_name(int m, int k, int n, int nonzeros, int replicas) {
  if (m <= 0 || n <= 0 || replicas <= 0) return "none";
  if (takes_panel(m, k, n, nonzeros, replicas, nullptr, 0, nullptr, 0)) return "spmm_panel64_kernel";
  return spmm_tiled_kernel_name(m, k, n, nonzeros, replicas);
}

int sputnik_hip_spmm_permuted_supported(int m, int k, int n, int nonzeros) {
  // (operand alignment is checked by the call itself)
  return m > 0 && nonzeros > 0 && k <= 512 &&
         spmm_panel_applicable(m, k, n, nonzeros, nullptr, 0, nullptr, 0) ? 1 : 0;
}

int sputnik_hip_spmm_permuted_batched(int m, int k, int n, int nonzeros, int replicas,
                                      const int* row_indices, const float* values,
                                      int64_t values_stride, const int* value_permutation,
                                      const int* row_offsets, const int* column_indices,
                                      const float* dense, int64_t dense_stride, float* out,
                                      int64_t out_stride, sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0 || value_permutation == nullptr)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || n == 0 || replicas == 0) return 0;
  if (nonzeros == 0 ||
      !spmm_panel_applicable(m, k, n, nonzeros, dense, dense_stride, out, out_stride))
    return SPUTNIK_HIP_UNSUPPORTED;
  return spmm_panel_launch(m, k, n, nonzeros, replicas, row_indices, values, values_stride,
                           row_offsets, column_indices, dense, dense_stride, out, out_stride,
                           stream, Epilogue{}, value_permutation);
}

namespace {
// Row blocks the transposing store handles: a multiple of 64 that divides the
// workgroup's 256 rows or is a multiple of them, and divides m.
bool block_rows_ok(int m, int block_rows) {
  return block_rows >= 64 && block_rows % 64 == 0 && m % block_rows == 0 &&
         (256 % block_rows == 0 || block_rows % 256 == 0);
}
}  // namespace

int sputnik_hip_spmm_transposed_out_supported(int m, int k, int n, int nonzeros, int block_rows) {
  return m > 0 && nonzeros > 0 && k <= 1024 && block_rows_ok(m, block_rows) &&
         spmm_panel_applicable(m, k, n, nonzeros, nullptr, 0, nullptr, 0) ? 1 : 0;
}

int sputnik_hip_spmm_transposed_out_batched(int m, int k, int n, int nonzeros, int replicas,
                                            const float* values, int64_t values_stride,
                                            const int* value_permutation, const int* row_offsets,
                                            const int* column_indices, const float* dense,
                                            int64_t dense_stride, const float* bias, int relu,
                                            int block_rows, float* out, int64_t out_stride,
                                            sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0 || block_rows <= 0)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || n == 0 || replicas == 0) return 0;
  if (nonzeros == 0 || !block_rows_ok(m, block_rows) ||
      !spmm_panel_applicable(m, k, n, nonzeros, dense, dense_stride, out, out_stride))
    return SPUTNIK_HIP_UNSUPPORTED;
  Epilogue epi;
  epi.bias = bias;
  epi.relu = relu != 0;
  return spmm_panel_launch(m, k, n, nonzeros, replicas, /*row_indices=*/nullptr, values,
                           values_stride, row_offsets, column_indices, dense, dense_stride, out,
                           out_stride, stream, epi, value_permutation, block_rows);
}

int sputnik_hip_spmm_group_supported(int m, int k, int n, int count, int block_rows,
                                     int accumulate) {
  if (block_rows > 0 && !block_rows_ok(m, block_rows)) return 0;
  return spmm_panel_group_supported(m, k, n, count, block_rows, accumulate != 0) ? 1 : 0;
}

static_assert(sizeof(sputnik_hip_spmm_problem) == sizeof(GroupProblemHost),
              "the C struct and the launcher's view of it are one layout");

int sputnik_hip_spmm_group_batched(int m, int k, int n, int replicas, int count,
                                   const sputnik_hip_spmm_problem* problems,
                                   int64_t dense_stride, int64_t out_stride, int block_rows,
                                   int accumulate, sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || replicas < 0 || count < 1 || problems == nullptr || block_rows < 0)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || n == 0 || replicas == 0) return 0;
  if (block_rows > 0 && !block_rows_ok(m, block_rows)) return SPUTNIK_HIP_UNSUPPORTED;
  return spmm_panel_group_launch(m, k, n, replicas, count,
                                 reinterpret_cast<const GroupProblemHost*>(problems),
                                 dense_stride, out_stride, block_rows, accumulate != 0, stream);
}

int sputnik_hip_spmm(int m, int k, int n, int nonzeros, const int* row_indices,
                     const float* values, const int* row_offsets, const int* column_indices,
                     const float* dense, float* out, sputnik_hip_stream_t stream) {
  return sputnik_hip_spmm_batched(m, k, n, nonzeros, 1, row_indices, values, 0, row_offsets,
                                  column_indices, dense, 0, out, 0, nullptr, 0, stream);
}

}  // extern "C"
