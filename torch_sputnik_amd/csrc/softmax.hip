// Sparse softmax over the stored entries of each CSR row, for gfx950.
//
// Replaces sputnik::SparseSoftmax as driven by src/softmax_cuda.cu:35-43.
//
// A group of LPR lanes owns one row.
// Rows of up to LPR*kRegs entries are read from HBM exactly once into
// registers (coalesced, lane-strided), reduced with DPP (max, then sum of
// exp) and written once: 8 bytes of HBM traffic per entry, the algorithmic
// minimum.  Longer rows fall back to three streaming passes whose re-reads
// are served by L2.  exp is the hardware v_exp_f32 path (__expf): relative error
// about 5e-6 at |x - max| ~ 88, far inside the 1e-4 budget; the accurate
// library expf made the kernel VALU-bound.
#include "common.h"
#include "wave_utils.h"

namespace sputnik_hip {
namespace {

constexpr int kBlock = 256;
constexpr int kRegs = 8;

template <int LPR>
__global__ __launch_bounds__(kBlock) void sparse_softmax_kernel(
    int m, const float* __restrict__ values, int64_t values_stride,
    const int* __restrict__ row_indices, const int* __restrict__ row_offsets,
    float* __restrict__ out, int64_t out_stride, float scale) {
  constexpr int kRowsPerBlock = kBlock / LPR;
  const int sub = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const int slot = blockIdx.x * kRowsPerBlock + sub;
  const int replica = blockIdx.y;
  values += replica * values_stride;
  out += replica * out_stride;

  // Rows are taken in storage order, not in row_indices order: neighbouring
  // groups then read neighbouring memory (a row is only a few hundred bytes, so
  // rows dealt by length touch partial cache lines at both ends), and the
  // result never depends on the order.  Lanes of out-of-range slots keep an
  // empty row so that every lane reaches the convergent reductions below.
  (void)row_indices;
  const int row = (slot < m) ? slot : 0;
  const int p0 = (slot < m) ? row_offsets[row] : 0;
  const int p1 = (slot < m) ? row_offsets[row + 1] : 0;
  const int len = p1 - p0;

  // Longest row handled by this wave decides the path (wave-uniform branch).
  int wave_max_len = len;
  if constexpr (LPR < kWave) {
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1)
      wave_max_len = max(wave_max_len, __shfl_xor(wave_max_len, off, kWave));
  }
  wave_max_len = __builtin_amdgcn_readfirstlane(wave_max_len);

  if (wave_max_len <= LPR * kRegs) {
    float x[kRegs];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < kRegs; ++i) {
      const int q = p0 + i * LPR + l;
      x[i] = (q < p1) ? values[q] * scale : -INFINITY;
      mx = fmaxf(mx, x[i]);
    }
    mx = group_max<LPR>(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kRegs; ++i) {
      const int q = p0 + i * LPR + l;
      x[i] = (q < p1) ? __expf(x[i] - mx) : 0.f;
      sum += x[i];
    }
    sum = group_sum<LPR>(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < kRegs; ++i) {
      const int q = p0 + i * LPR + l;
      if (q < p1) out[q] = x[i] * inv;
    }
  } else {
    float mx = -INFINITY;
    for (int q = p0 + l; q < p1; q += LPR) mx = fmaxf(mx, values[q] * scale);
    mx = group_max<LPR>(mx);
    float sum = 0.f;
    for (int q = p0 + l; q < p1; q += LPR) sum += __expf(values[q] * scale - mx);
    sum = group_sum<LPR>(sum);
    const float inv = 1.f / sum;
    for (int q = p0 + l; q < p1; q += LPR) out[q] = __expf(values[q] * scale - mx) * inv;
  }
}

// Gradient of y = softmax(scale * x) over the stored entries of each row:
// dx = scale * y * (dy - sum_row(dy * y)).  Same row-per-group layout; rows
// that fit the registers move 12 bytes per entry (y, dy in; dx out).
template <int LPR>
__global__ __launch_bounds__(kBlock) void sparse_softmax_backward_kernel(
    int m, const float* __restrict__ y, int64_t y_stride, const float* __restrict__ dy,
    int64_t dy_stride, const int* __restrict__ row_offsets, float* __restrict__ dx,
    int64_t dx_stride, float scale) {
  constexpr int kRowsPerBlock = kBlock / LPR;
  const int sub = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const int slot = blockIdx.x * kRowsPerBlock + sub;
  const int replica = blockIdx.y;
  y += replica * y_stride;
  dy += replica * dy_stride;
  dx += replica * dx_stride;
  const int row = (slot < m) ? slot : 0;
  const int p0 = (slot < m) ? row_offsets[row] : 0;
  const int p1 = (slot < m) ? row_offsets[row + 1] : 0;

  int wave_max_len = p1 - p0;
  if constexpr (LPR < kWave) {
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1)
      wave_max_len = max(wave_max_len, __shfl_xor(wave_max_len, off, kWave));
  }
  wave_max_len = __builtin_amdgcn_readfirstlane(wave_max_len);

  if (wave_max_len <= LPR * kRegs) {
    float yv[kRegs], gv[kRegs];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < kRegs; ++i) {
      const int q = p0 + i * LPR + l;
      yv[i] = (q < p1) ? y[q] : 0.f;
      gv[i] = (q < p1) ? dy[q] : 0.f;
      dot = fmaf(yv[i], gv[i], dot);
    }
    dot = group_sum<LPR>(dot);
#pragma unroll
    for (int i = 0; i < kRegs; ++i) {
      const int q = p0 + i * LPR + l;
      if (q < p1) dx[q] = scale * yv[i] * (gv[i] - dot);
    }
  } else {
    float dot = 0.f;
    for (int q = p0 + l; q < p1; q += LPR) dot = fmaf(y[q], dy[q], dot);
    dot = group_sum<LPR>(dot);
    for (int q = p0 + l; q < p1; q += LPR) dx[q] = scale * y[q] * (dy[q] - dot);
  }
}

template <int LPR>
int launch(int m, int replicas, const float* values, int64_t values_stride,
           const int* row_indices, const int* row_offsets, float* out, int64_t out_stride,
           float scale, hipStream_t stream) {
  const int gx = ceil_div(m, kBlock / LPR);
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    hipLaunchKernelGGL((sparse_softmax_kernel<LPR>), dim3(gx, ry), dim3(kBlock), 0, stream, m,
                       values + r0 * values_stride, values_stride, row_indices, row_offsets,
                       out + r0 * out_stride, out_stride, scale);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

template <int LPR>
int launch_backward(int m, int replicas, const float* y, int64_t y_stride, const float* dy,
                    int64_t dy_stride, const int* row_offsets, float* dx, int64_t dx_stride,
                    float scale, hipStream_t stream) {
  const int gx = ceil_div(m, kBlock / LPR);
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    hipLaunchKernelGGL((sparse_softmax_backward_kernel<LPR>), dim3(gx, ry), dim3(kBlock), 0,
                       stream, m, y + r0 * y_stride, y_stride, dy + r0 * dy_stride, dy_stride,
                       row_offsets, dx + r0 * dx_stride, dx_stride, scale);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

// Smallest group whose register capacity (LPR * 8 entries) still holds rows
// 25 % longer than the mean: every unused register slot costs an exp.
inline int lanes_per_row(int m, int nonzeros) {
  const int mean_len = nonzeros / m;
  if (mean_len * 5 <= 16 * 8 * 4) return 16;
  if (mean_len * 5 <= 32 * 8 * 4) return 32;
  return 64;
}

}  // namespace
}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

int sputnik_hip_sparse_softmax_scaled_batched(int m, int n, int nonzeros, int replicas,
                                              const float* values, int64_t values_stride,
                                              const int* row_indices, const int* row_offsets,
                                              const int* column_indices, float scale, float* out,
                                              int64_t out_stride, sputnik_hip_stream_t stream) {
  (void)n;
  (void)column_indices;
  if (m < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || replicas == 0) return 0;
  // Mean row length picks the lanes-per-row split (host-side, no sync: m and
  // nonzeros are arguments).
  switch (lanes_per_row(m, nonzeros)) {
    case 16:
      return launch<16>(m, replicas, values, values_stride, row_indices, row_offsets, out,
                        out_stride, scale, stream);
    case 32:
      return launch<32>(m, replicas, values, values_stride, row_indices, row_offsets, out,
                        out_stride, scale, stream);
    default:
      return launch<64>(m, replicas, values, values_stride, row_indices, row_offsets, out,
                        out_stride, scale, stream);
  }
}

int sputnik_hip_sparse_softmax_batched(int m, int n, int nonzeros, int replicas,
                                       const float* values, int64_t values_stride,
                                       const int* row_indices, const int* row_offsets,
                                       const int* column_indices, float* out,
                                       int64_t out_stride, sputnik_hip_stream_t stream) {
  return sputnik_hip_sparse_softmax_scaled_batched(m, n, nonzeros, replicas, values,
                                                   values_stride, row_indices, row_offsets,
                                                   column_indices, 1.0f, out, out_stride, stream);
}

int sputnik_hip_sparse_softmax_backward_batched(int m, int nonzeros, int replicas,
                                                const float* softmax_out, int64_t out_stride,
                                                const float* grad_out, int64_t grad_out_stride,
                                                const int* row_offsets, float scale,
                                                float* grad_values, int64_t grad_values_stride,
                                                sputnik_hip_stream_t stream) {
  if (m < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || replicas == 0) return 0;
  switch (lanes_per_row(m, nonzeros)) {
    case 16:
      return launch_backward<16>(m, replicas, softmax_out, out_stride, grad_out, grad_out_stride,
                                 row_offsets, grad_values, grad_values_stride, scale, stream);
    case 32:
      return launch_backward<32>(m, replicas, softmax_out, out_stride, grad_out, grad_out_stride,
                                 row_offsets, grad_values, grad_values_stride, scale, stream);
    default:
      return launch_backward<64>(m, replicas, softmax_out, out_stride, grad_out, grad_out_stride,
                                 row_offsets, grad_values, grad_values_stride, scale, stream);
  }
}

int sputnik_hip_sparse_softmax(int m, int n, int nonzeros, const float* values,
                               const int* row_indices, const int* row_offsets,
                               const int* column_indices, float* out,
                               sputnik_hip_stream_t stream) {
  return sputnik_hip_sparse_softmax_batched(m, n, nonzeros, 1, values, 0, row_indices,
                                            row_offsets, column_indices, out, 0, stream);
}

}  // extern "C"
