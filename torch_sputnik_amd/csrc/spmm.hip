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
#include <type_traits>

#include "common.h"
#include "mfma.h"
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

bool spmm_panel_applicable_typed(int m, int k, int n, int nonzeros, const void* dense,
                                 int dense_type, int64_t dense_stride, const float* out,
                                 int64_t out_stride);
int spmm_panel_launch_typed(int m, int k, int n, int nonzeros, int replicas,
                            const int* row_indices, const void* values, int values_type,
                            int64_t values_stride, const int* row_offsets,
                            const int* column_indices, const void* dense, int dense_type,
                            int64_t dense_stride, float* out, int64_t out_stride,
                            hipStream_t stream, Epilogue epi);

namespace {

constexpr int kBlock = 256;

// VEC elements of storage type T, widened to float.
template <int VEC, typename T>
__device__ __forceinline__ void load_widened(float (&dst)[VEC], const T* __restrict__ src) {
  if constexpr (std::is_same_v<T, float>) {
    load_vec<VEC>(dst, src);
  } else if constexpr (VEC == 1) {
    dst[0] = static_cast<float>(*src);
  } else {
    using V = T __attribute__((ext_vector_type(VEC)));
    const V v = *reinterpret_cast<const V*>(src);
#pragma unroll
    for (int e = 0; e < VEC; ++e) dst[e] = static_cast<float>(v[e]);
  }
}

// TV / TB: storage type of the values / the dense operand (float; _Float16 / __bf16 =
// native half operands, round 3: widened in registers, half the gather traffic).
template <int VEC, int LPR, typename TV = float, typename TB = float>
__global__ __launch_bounds__(kBlock) void spmm_rowgather_kernel(
    int m, int n, const int* __restrict__ row_indices, const TV* __restrict__ values,
    int64_t values_stride, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, const TB* __restrict__ dense,
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

  const TB* __restrict__ b_col = dense + (col_ok ? c0 : 0);
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

  // A row is a chain of dependent memory round trips -- its window of LPR (column, value)
  // pairs, then the rows of B they name -- and with four gathers in flight and the window
  // fetched on demand an entry took 0.19 us (round 5, tools/spmm_dispatch_sweep.py: 40 us
  // for rows of 205 entries whatever the size of the call).  Now the NEXT window is in
  // flight while this one is worked on, and its gathers go out eight at a time (an entry
  // past the row's end gathers nothing and multiplies a zero by a zero).
  constexpr int kBatch = LPR < 8 ? LPR : 8;
  int j = 0;
  float a = 0.f;
  if (p + l < p_end) {
    j = column_indices[p + l];
    a = static_cast<float>(values[p + l]);
  }
  for (; p < p_end; p += LPR) {
    const int q_next = p + LPR + l;
    int j_next = 0;
    float a_next = 0.f;
    if (q_next < p_end) {
      j_next = column_indices[q_next];
      a_next = static_cast<float>(values[q_next]);
    }
    const int cnt = min(LPR, p_end - p);
    for (int t0 = 0; t0 < cnt; t0 += kBatch) {
      float b[kBatch][VEC], aa[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int jj = group_broadcast<LPR>(j, t0 + u);
        aa[u] = group_broadcast<LPR>(a, t0 + u);
#pragma unroll
        for (int v = 0; v < VEC; ++v) b[u][v] = 0.f;
        if (t0 + u < cnt) load_widened<VEC, TB>(b[u], b_col + static_cast<int64_t>(jj) * n);
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const float au = t0 + u < cnt ? aa[u] : 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(au, b[u][v], acc[v]);
      }
    }
    j = j_next;
    a = a_next;
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

template <int VEC, int LPR, typename TV = float, typename TB = float>
int launch_rowgather(int m, int n, int replicas, const int* row_indices, const TV* values,
                     int64_t values_stride, const int* row_offsets, const int* column_indices,
                     const TB* dense, int64_t dense_stride, float* out, int64_t out_stride,
                     hipStream_t stream, Epilogue epi) {
  constexpr int kRowsPerBlock = kBlock / LPR;
  const int gx = ceil_div(m, kRowsPerBlock);
  const int gy = ceil_div(n, LPR * VEC);
  if (gy > kMaxGridYZ) return SPUTNIK_HIP_INVALID_ARGUMENT;
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int rz = min(replicas - r0, kMaxGridYZ);
    hipLaunchKernelGGL((spmm_rowgather_kernel<VEC, LPR, TV, TB>), dim3(gx, gy, rz), dim3(kBlock), 0,
                       stream, m, n, row_indices, values + r0 * values_stride, values_stride,
                       row_offsets, column_indices, dense + r0 * dense_stride, dense_stride,
                       out + r0 * out_stride, out_stride, epi);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

template <int VEC, typename TV = float, typename TB = float>
int launch_rowgather_vec(int m, int n, int replicas, const int* row_indices, const TV* values,
                         int64_t values_stride, const int* row_offsets,
                         const int* column_indices, const TB* dense, int64_t dense_stride,
                         float* out, int64_t out_stride, hipStream_t stream, Epilogue epi) {
  const int lanes_needed = ceil_div(n, VEC);
#define SPUTNIK_HIP_RG(LPR)                                                                       \
  return launch_rowgather<VEC, LPR, TV, TB>(m, n, replicas, row_indices, values, values_stride,   \
                                            row_offsets, column_indices, dense, dense_stride,     \
                                            out, out_stride, stream, epi)
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

namespace {
// count8 pieces of 8 elements (16 bytes in, 32 out) + a scalar tail.
template <typename T>
__global__ __launch_bounds__(256) void widen_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                    int64_t count) {
  using raw8 = T __attribute__((ext_vector_type(8)));
  using f8v = float __attribute__((ext_vector_type(8)));
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 8;
  if (i + 8 <= count) {
    *reinterpret_cast<f8v*>(out + i) = __builtin_convertvector(*reinterpret_cast<const raw8*>(in + i), f8v);
  } else {
    for (int64_t j = i; j < count; ++j) out[j] = static_cast<float>(in[j]);
  }
}
int widen_to(const void* in, int type, float* out, int64_t count, hipStream_t stream) {
  if (count == 0) return 0;
  const int64_t blocks = ceil_div64(ceil_div64(count, 8), 256);
  if (blocks > 0x7fffffff) return SPUTNIK_HIP_INVALID_ARGUMENT;
  // (16-byte vector loads: a source that is not 16-byte aligned goes element by element)
  if (!aligned_to(in, 16)) return SPUTNIK_HIP_UNSUPPORTED;
  if (type == SPUTNIK_HIP_F16)
    hipLaunchKernelGGL(widen_kernel<_Float16>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       stream, static_cast<const _Float16*>(in), out, count);
  else
    hipLaunchKernelGGL(widen_kernel<__bf16>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       stream, static_cast<const __bf16*>(in), out, count);
  return launch_status();
}
size_t align256(size_t v) { return (v + 255) / 256 * 256; }

template <typename TV, typename TB>
int rowgather_typed(int m, int n, int replicas, const int* row_indices, const void* values,
                    int64_t values_stride, const int* row_offsets, const int* column_indices,
                    const void* dense, int64_t dense_stride, float* out, int64_t out_stride,
                    hipStream_t stream, Epilogue epi) {
  const TV* v = static_cast<const TV*>(values);
  const TB* d = static_cast<const TB*>(dense);
  // widest vector of ELEMENTS both the dense rows and the output rows support
  int vec = vector_width(out, n, out_stride);
  if (!(n % 4 == 0 && dense_stride % 4 == 0 && aligned_to(dense, 4 * sizeof(TB)))) vec = min(vec, 2);
  if (!(n % 2 == 0 && dense_stride % 2 == 0 && aligned_to(dense, 2 * sizeof(TB)))) vec = 1;
  switch (vec) {
    case 4:
      return launch_rowgather_vec<4, TV, TB>(m, n, replicas, row_indices, v, values_stride,
                                             row_offsets, column_indices, d, dense_stride, out,
                                             out_stride, stream, epi);
    case 2:
      return launch_rowgather_vec<2, TV, TB>(m, n, replicas, row_indices, v, values_stride,
                                             row_offsets, column_indices, d, dense_stride, out,
                                             out_stride, stream, epi);
    default:
      return launch_rowgather_vec<1, TV, TB>(m, n, replicas, row_indices, v, values_stride,
                                             row_offsets, column_indices, d, dense_stride, out,
                                             out_stride, stream, epi);
  }
}
}  // namespace

int spmm_rowgather_launch_typed(int m, int n, int replicas, const int* row_indices,
                                const void* values, int values_type, int64_t values_stride,
                                const int* row_offsets, const int* column_indices,
                                const void* dense, int dense_type, int64_t dense_stride, float* out,
                                int64_t out_stride, hipStream_t stream, Epilogue epi) {
#define SPUTNIK_HIP_RGT(TV, TB)                                                                  \
  return rowgather_typed<TV, TB>(m, n, replicas, row_indices, values, values_stride, row_offsets, \
                                 column_indices, dense, dense_stride, out, out_stride, stream, epi)
  const int F = SPUTNIK_HIP_F32, H = SPUTNIK_HIP_F16, B = SPUTNIK_HIP_BF16;
  if (values_type == H && dense_type == H) SPUTNIK_HIP_RGT(_Float16, _Float16);
  if (values_type == B && dense_type == B) SPUTNIK_HIP_RGT(__bf16, __bf16);
  if (values_type == H && dense_type == F) SPUTNIK_HIP_RGT(_Float16, float);
  if (values_type == B && dense_type == F) SPUTNIK_HIP_RGT(__bf16, float);
  if (values_type == F && dense_type == H) SPUTNIK_HIP_RGT(float, _Float16);
  if (values_type == F && dense_type == B) SPUTNIK_HIP_RGT(float, __bf16);
#undef SPUTNIK_HIP_RGT
  return SPUTNIK_HIP_UNSUPPORTED;
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
  // (round 5: with the rows cut at the panel boundary the long rows that round 2's masked walk
  // lost on -- 320 entries and more -- are won back where the product is one or two column
  // tiles wide: 1024^2 at density 0.5 x 64 columns x 64 replicas 119 against 142 us for the
  // 64-column kernel, x 128 columns 228 against 251; at 256 columns and more the chunked
  // kernels keep their 12-15 %)
  if (k > 1024 || (k > 512 && nonzeros > 320 * static_cast<int64_t>(m) && n > 128)) return false;
  // (two panels on a grid that does not fill the chip: 2048 x 1024 x 1024, one
  // replica, 128 workgroups: 33.0 vs 30.9 us for the chunked kernel)
  if (k > 512 && static_cast<int64_t>((m + 255) / 256) * ((n + 63) / 64) * replicas < 192)
    return false;
  const int choice = spmm_tiled_choice(m, k, n, nonzeros, replicas);
  const int64_t work = static_cast<int64_t>(nonzeros) * n * replicas;
  // (round 5, tools/spmm_dispatch_sweep.py: ONE resident panel against up to four column
  // tiles also beats the 256-column kernel -- 512^2 x 256 x 64 replicas at density 0.3: 74
  // against 87 us, at 0.1: 38 against 44; at 512 columns the wide kernels lead again)
  // (against the row gather the panel kernel needs workgroups: one per 256 rows and 64
  // columns -- 512^2 x 256 at density 0.3 is 8 of them and 31 us against 16; with 128,
  // 512^2 x 512 x 8 replicas at 0.1, it is 18 against 24)
  const int64_t workgroups = static_cast<int64_t>((m + 255) / 256) * ((n + 63) / 64) * replicas;
  return choice == 2 || choice == 3 ||
         (choice == 0 && work >= (int64_t{1} << 24) && workgroups >= 96) ||
         (choice == 1 && k <= 512 && n <= 256);
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
  // One panel, or (round 5) two with short rows: since round 4 the two-panel kernel CUTS its
  // rows at the panel boundary and walks every entry once, so a value is gathered once --
  // config 3's transposed attention products (1024^2 at density 0.1 x 64 x 64 replicas): 51.8
  // us against 60.2 for the banded permutation + the product, bit-identical
  // (tools/spmm_c3_bench.py --permuted).  (Round 2 measured 162 against 41 + 50 on the masked
  // walk, which gathered every value twice; rows whose columns do not ascend still take it.)
  // (SPUTNIK_HIP_SPMM_DEBUG bit 0x20000: one panel only, the rule of rounds 2-4 -- A/B runs)
  const bool panels_ok = k <= 512 || (k <= 1024 && (nonzeros <= 320 * static_cast<int64_t>(m) || n <= 128) &&
                                      !(options().spmm_debug & 0x20000));   // (as takes_panel)
  return m > 0 && nonzeros > 0 && panels_ok &&
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

size_t sputnik_hip_left_spmm_half_tiles_workspace_bytes(int m, int k, int n, int nonzeros,
                                                        int replicas, int values_type,
                                                        int dense_type, int tile_type) {
  if (!spmm_mfma_shape(m, k, n, nonzeros, replicas, values_type, dense_type, tile_type)) return 0;
  return spmm_mfma_workspace_bytes(m, k, n, replicas, values_type, dense_type, tile_type);
}

int sputnik_hip_left_spmm_half_tiles(int m, int k, int n, int nonzeros, int replicas,
                                     const int* row_offsets, const int* column_indices,
                                     const void* values, int values_type, const void* dense,
                                     int dense_type, int64_t dense_stride, int tile_type,
                                     const float* bias, int relu, float* out, int64_t out_stride,
                                     void* workspace, size_t workspace_bytes,
                                     sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (!spmm_mfma_shape(m, k, n, nonzeros, replicas, values_type, dense_type, tile_type))
    return SPUTNIK_HIP_UNSUPPORTED;
  const size_t need = spmm_mfma_workspace_bytes(m, k, n, replicas, values_type, dense_type, tile_type);
  if (workspace == nullptr || !aligned_to(workspace, 256) || workspace_bytes < need ||
      !aligned_to(dense, 16) || !aligned_to(out, 4) || dense_stride % 8 != 0 ||
      !aligned_to(values, values_type == SPUTNIK_HIP_F32 ? 4 : 2))
    return SPUTNIK_HIP_UNSUPPORTED;
  return spmm_mfma_launch(m, k, n, nonzeros, replicas, row_offsets, column_indices, values,
                          values_type, dense, dense_type, dense_stride, tile_type, bias, relu, out,
                          out_stride, workspace, stream);
}

// Workspace of sputnik_hip_spmm_typed: the float form's, plus room for float copies of
// the half operands for the shapes the half-reading kernels do not serve well.
size_t sputnik_hip_spmm_typed_workspace_bytes(int m, int k, int n, int nonzeros, int replicas,
                                              int values_type, int64_t values_stride,
                                              int dense_type) {
  if (m <= 0 || k <= 0 || n <= 0 || replicas <= 0) return 0;
  // (shapes the half-reading kernels take need none: see sputnik_hip_spmm_typed)
  const bool panel_shape = n % 4 == 0 && n >= 64 && m >= 16 &&
                           (k <= 512 || (k <= 1024 && nonzeros <= 320 * static_cast<int64_t>(m)));
  const bool big = static_cast<int64_t>(nonzeros) * n * replicas >= (int64_t{1} << 28);
  // (the matrix-core route of a half dense operand against shared values: spmm_mfma.hip)
  const size_t tiles_bytes =
      values_stride == 0 && spmm_mfma_shape(m, k, n, nonzeros, replicas, values_type, dense_type, dense_type)
          ? spmm_mfma_workspace_bytes(m, k, n, replicas, values_type, dense_type, dense_type) : 0;
  if (panel_shape || !big || (values_type == SPUTNIK_HIP_F32 && dense_type == SPUTNIK_HIP_F32))
    return tiles_bytes;
  if (tiles_bytes != 0) return tiles_bytes;
  size_t bytes = align256(spmm_tiled_workspace_bytes(m, k, n, nonzeros));
  if (dense_type != SPUTNIK_HIP_F32)
    bytes += align256(sizeof(float) * static_cast<size_t>(replicas) * k * n);
  if (values_type != SPUTNIK_HIP_F32)
    bytes += align256(sizeof(float) * static_cast<size_t>(nonzeros) * (values_stride == 0 ? 1 : replicas));
  return bytes;
}

int sputnik_hip_spmm_typed(int m, int k, int n, int nonzeros, int replicas,
                           const int* row_indices, const void* values, int values_type,
                           int64_t values_stride, const int* row_offsets,
                           const int* column_indices, const void* dense, int dense_type,
                           int64_t dense_stride, const float* bias, int relu, float* out,
                           int64_t out_stride, void* workspace, size_t workspace_bytes,
                           sputnik_hip_stream_t stream) {
  const auto known = [](int t) {
    return t == SPUTNIK_HIP_F32 || t == SPUTNIK_HIP_F16 || t == SPUTNIK_HIP_BF16;
  };
  if (!known(values_type) || !known(dense_type)) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (values_type == SPUTNIK_HIP_F32 && dense_type == SPUTNIK_HIP_F32)
    return sputnik_hip_spmm_bias_batched(m, k, n, nonzeros, replicas, row_indices,
                                         static_cast<const float*>(values), values_stride,
                                         row_offsets, column_indices,
                                         static_cast<const float*>(dense), dense_stride, bias, relu,
                                         out, out_stride, workspace, workspace_bytes, stream);
  if (values_type != SPUTNIK_HIP_F32 && dense_type != SPUTNIK_HIP_F32 && values_type != dense_type)
    return SPUTNIK_HIP_UNSUPPORTED;   // float16 against bfloat16
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || n == 0 || replicas == 0) return 0;
  if (!aligned_to(values, values_type == SPUTNIK_HIP_F32 ? 4 : 2) ||
      !aligned_to(dense, dense_type == SPUTNIK_HIP_F32 ? 4 : 2) || !aligned_to(out, 4))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  // Shared values at layer density against a half dense operand: a dense contraction on
  // the matrix cores (spmm_mfma.hip) where the shape is served and the workspace is there.
  if (values_stride == 0 && dense_type != SPUTNIK_HIP_F32) {
    const int st = sputnik_hip_left_spmm_half_tiles(
        m, k, n, nonzeros, replicas, row_offsets, column_indices, values, values_type, dense,
        dense_type, dense_stride, dense_type, bias, relu, out, out_stride, workspace, workspace_bytes,
        stream);
    if (st != SPUTNIK_HIP_UNSUPPORTED) return st;
  }
  Epilogue epi;
  epi.bias = bias;
  epi.relu = relu != 0;
  // Panel-resident kernel (the half panel is widened on its way into LDS) wherever it
  // applies and is not hopeless (at most four panels, i.e. k <= 2048; forced by the
  // "panel" knob up to its limit), otherwise the row-gather kernel, which reads half
  // rows of B straight from L2.  Neither needs the workspace.
  // (measured, tools/half_bench.py: attention P.V 1024^2 x 64 x 64 replicas 42.8 us native
  // against 61 for cast + float kernels, projection 19 against 36; 2048^2 x 512 x 8, four
  // panels: 366 against 189 -- so beyond two panels the operands are widened into the
  // workspace by one pass of this library and the chunked float kernels run)
  const int forced = options().spmm_kernel;
  const bool panel_ok = nonzeros > 0 && (forced == 0 || forced == 3) &&
                        spmm_panel_applicable_typed(m, k, n, nonzeros, dense, dense_type,
                                                    dense_stride, out, out_stride) &&
                        (forced == 3 || k <= 512 ||
                         (k <= 1024 && (nonzeros <= 320 * static_cast<int64_t>(m) || n <= 128))) &&
                        static_cast<int64_t>(nonzeros) * n * replicas >= (int64_t{1} << 22) &&
                        // (round 5, tools/spmm_dispatch_sweep.py --half: as for float operands the
                        // panel kernel needs workgroups against the row gather -- 1024^2 x 256,
                        // one replica, 16 of them: 35 against 23 us at density 0.1, 82 against 57
                        // at 0.3)
                        (forced == 3 ||
                         static_cast<int64_t>((m + 255) / 256) * ((n + 63) / 64) * replicas >= 96);
  if (panel_ok)
    return spmm_panel_launch_typed(m, k, n, nonzeros, replicas, row_indices, values, values_type,
                                   values_stride, row_offsets, column_indices, dense, dense_type,
                                   dense_stride, out, out_stride, stream, epi);
  const bool big = static_cast<int64_t>(nonzeros) * n * replicas >= (int64_t{1} << 28);
  if (forced == 0 && big && workspace != nullptr && aligned_to(workspace, 256) &&
      dense_stride == static_cast<int64_t>(k) * n && (values_stride == 0 || values_stride == nonzeros) &&
      sputnik_hip_spmm_typed_workspace_bytes(m, k, n, nonzeros, replicas, values_type,
                                             values_stride, dense_type) != 0 &&
      workspace_bytes >= sputnik_hip_spmm_typed_workspace_bytes(m, k, n, nonzeros, replicas,
                                                                values_type, values_stride,
                                                                dense_type)) {
    char* at = static_cast<char*>(workspace);
    void* float_ws = at;
    const size_t float_ws_bytes = align256(spmm_tiled_workspace_bytes(m, k, n, nonzeros));
    at += float_ws_bytes;
    const float* dense_f = static_cast<const float*>(dense);
    const float* values_f = static_cast<const float*>(values);
    int st = 0;
    if (dense_type != SPUTNIK_HIP_F32) {
      const int64_t count = static_cast<int64_t>(replicas) * k * n;
      st = widen_to(dense, dense_type, reinterpret_cast<float*>(at), count, stream);
      dense_f = reinterpret_cast<const float*>(at);
      at += align256(sizeof(float) * static_cast<size_t>(count));
    }
    if (st == 0 && values_type != SPUTNIK_HIP_F32) {
      const int64_t count = static_cast<int64_t>(nonzeros) * (values_stride == 0 ? 1 : replicas);
      st = widen_to(values, values_type, reinterpret_cast<float*>(at), count, stream);
      values_f = reinterpret_cast<const float*>(at);
    }
    if (st == 0)
      return sputnik_hip_spmm_bias_batched(m, k, n, nonzeros, replicas, row_indices, values_f,
                                           values_stride, row_offsets, column_indices, dense_f,
                                           dense_stride, bias, relu, out, out_stride, float_ws,
                                           float_ws_bytes, stream);
    if (st != SPUTNIK_HIP_UNSUPPORTED) return st;
  }
  return spmm_rowgather_launch_typed(m, n, replicas, row_indices, values, values_type,
                                     values_stride, row_offsets, column_indices, dense, dense_type,
                                     dense_stride, out, out_stride, stream, epi);
}

int sputnik_hip_spmm(int m, int k, int n, int nonzeros, const int* row_indices,
                     const float* values, const int* row_offsets, const int* column_indices,
                     const float* dense, float* out, sputnik_hip_stream_t stream) {
  return sputnik_hip_spmm_batched(m, k, n, nonzeros, 1, row_indices, values, 0, row_offsets,
                                  column_indices, dense, 0, out, 0, nullptr, 0, stream);
}

}  // extern "C"
