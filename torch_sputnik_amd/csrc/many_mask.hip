// "many mask" family: one topology per batch element, shared by its heads.
//
// No counterpart in the reference's src/ (the call sites are
// tests/transformer/functions.py:20-177 and
// tests/test_attention_many_masks.py:120-150; layouts from
// tests/transformer/utils.py:17-38).  The per-mask nonzero counts arrive on
// the host, so each mask becomes one batched launch over its heads on the
// caller's stream: `masks` launches instead of `replicas`, with no host
// synchronisation.  The launches share the workspace: they are ordered on
// the stream and each one re-derives its plan.
#include "common.h"

using namespace sputnik_hip;

namespace {

struct MaskWalk {
  int heads;
  bool ok;
};

inline MaskWalk check(int masks, int m, int n, const int* nonzeros, int replicas) {
  MaskWalk w{0, false};
  if (masks < 0 || m < 0 || n < 0 || replicas < 0) return w;
  if (masks == 0) {
    w.ok = replicas == 0;
    return w;
  }
  if (nonzeros == nullptr || replicas % masks != 0) return w;
  for (int i = 0; i < masks; ++i)
    if (nonzeros[i] < 0) return w;
  w.heads = replicas / masks;
  w.ok = true;
  return w;
}

}  // namespace

extern "C" {

int sputnik_hip_spmm_many_mask(int masks, int m, int k, int n, const int* nonzeros, int replicas,
                               const int* row_indices, const float* values,
                               int64_t values_stride, const int* row_offsets,
                               const int* column_indices, const float* dense,
                               int64_t dense_stride, float* out, int64_t out_stride,
                               void* workspace, size_t workspace_bytes,
                               sputnik_hip_stream_t stream) {
  const MaskWalk w = check(masks, m, n, nonzeros, replicas);
  if (!w.ok || k < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  int64_t first = 0;  // first nonzero of mask i in the concatenated arrays
  for (int i = 0; i < masks; ++i) {
    const int64_t r0 = static_cast<int64_t>(i) * w.heads;
    const int st = sputnik_hip_spmm_batched(
        m, k, n, nonzeros[i], w.heads, row_indices + static_cast<int64_t>(i) * m,
        values + r0 * values_stride, values_stride, row_offsets + static_cast<int64_t>(i) * (m + 1),
        column_indices + first, dense + r0 * dense_stride, dense_stride, out + r0 * out_stride,
        out_stride, workspace, workspace_bytes, stream);
    if (st != 0) return st;
    first += nonzeros[i];
  }
  return 0;
}

int sputnik_hip_sddmm_many_mask(int masks, int m, int k, int n, const int* nonzeros,
                                int replicas, const int* row_indices, const int* row_offsets,
                                const int* column_indices, const float* lhs, int64_t lhs_stride,
                                const float* rhs, int64_t rhs_stride, float* out,
                                int64_t out_stride, void* workspace, size_t workspace_bytes,
                                sputnik_hip_stream_t stream) {
  const MaskWalk w = check(masks, m, n, nonzeros, replicas);
  if (!w.ok || k < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  int64_t first = 0;
  for (int i = 0; i < masks; ++i) {
    const int64_t r0 = static_cast<int64_t>(i) * w.heads;
    const int st = sputnik_hip_sddmm_batched(
        m, k, n, nonzeros[i], w.heads, row_indices + static_cast<int64_t>(i) * m,
        row_offsets + static_cast<int64_t>(i) * (m + 1), column_indices + first,
        lhs + r0 * lhs_stride, lhs_stride, rhs + r0 * rhs_stride, rhs_stride,
        out + r0 * out_stride, out_stride, workspace, workspace_bytes, stream);
    if (st != 0) return st;
    first += nonzeros[i];
  }
  return 0;
}

int sputnik_hip_sparse_softmax_many_mask(int masks, int m, const int* nonzeros, int replicas,
                                         const float* values, int64_t values_stride,
                                         const int* row_indices, const int* row_offsets,
                                         const int* column_indices, float scale, float* out,
                                         int64_t out_stride, sputnik_hip_stream_t stream) {
  const MaskWalk w = check(masks, m, 0, nonzeros, replicas);
  if (!w.ok) return SPUTNIK_HIP_INVALID_ARGUMENT;
  int64_t first = 0;
  for (int i = 0; i < masks; ++i) {
    const int64_t r0 = static_cast<int64_t>(i) * w.heads;
    const int st = sputnik_hip_sparse_softmax_scaled_batched(
        m, -1, nonzeros[i], w.heads, values + r0 * values_stride, values_stride,
        row_indices + static_cast<int64_t>(i) * m, row_offsets + static_cast<int64_t>(i) * (m + 1),
        column_indices + first, scale, out + r0 * out_stride, out_stride, stream);
    if (st != 0) return st;
    first += nonzeros[i];
  }
  return 0;
}

int sputnik_hip_sparse_softmax_backward_many_mask(int masks, int m, const int* nonzeros,
                                                  int replicas, const float* softmax_out,
                                                  int64_t out_stride, const float* grad_out,
                                                  int64_t grad_out_stride,
                                                  const int* row_offsets, float scale,
                                                  float* grad_values,
                                                  int64_t grad_values_stride,
                                                  sputnik_hip_stream_t stream) {
  const MaskWalk w = check(masks, m, 0, nonzeros, replicas);
  if (!w.ok) return SPUTNIK_HIP_INVALID_ARGUMENT;
  for (int i = 0; i < masks; ++i) {
    const int64_t r0 = static_cast<int64_t>(i) * w.heads;
    const int st = sputnik_hip_sparse_softmax_backward_batched(
        m, nonzeros[i], w.heads, softmax_out + r0 * out_stride, out_stride,
        grad_out + r0 * grad_out_stride, grad_out_stride,
        row_offsets + static_cast<int64_t>(i) * (m + 1), scale,
        grad_values + r0 * grad_values_stride, grad_values_stride, stream);
    if (st != 0) return st;
  }
  return 0;
}

int sputnik_hip_csr_transpose_many_mask(int masks, int m, int n, const int* nonzeros,
                                        int replicas, const float* values,
                                        int64_t values_stride, const int* row_offsets,
                                        const int* column_indices, float* out_values,
                                        int64_t out_values_stride, int* out_row_offsets,
                                        int* out_column_indices, int* out_permutation,
                                        void* workspace, size_t workspace_bytes,
                                        sputnik_hip_stream_t stream) {
  const MaskWalk w = check(masks, m, n, nonzeros, replicas);
  if (!w.ok) return SPUTNIK_HIP_INVALID_ARGUMENT;
  int64_t first = 0;
  for (int i = 0; i < masks; ++i) {
    const int64_t r0 = static_cast<int64_t>(i) * w.heads;
    const int st = sputnik_hip_csr_transpose(
        m, n, nonzeros[i], values != nullptr ? w.heads : 0,
        values != nullptr ? values + r0 * values_stride : nullptr,
        values_stride, row_offsets + static_cast<int64_t>(i) * (m + 1), column_indices + first,
        out_values != nullptr ? out_values + r0 * out_values_stride : nullptr, out_values_stride,
        out_row_offsets + static_cast<int64_t>(i) * (n + 1), out_column_indices + first,
        out_permutation != nullptr ? out_permutation + first : nullptr, workspace,
        workspace_bytes, stream);
    if (st != 0) return st;
    first += nonzeros[i];
  }
  return 0;
}

}  // extern "C"
