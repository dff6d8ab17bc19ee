// "many mask" family: one topology per batch element, shared by its heads.
//
// No counterpart in the reference's src/ (the call sites are
// tests/transformer/functions.py:20-177 and
// tests/test_attention_many_masks.py:120-150; layouts from
// tests/transformer/utils.py:17-38).  ONE launch per operator serves all masks
// where the shape's kernel takes the concatenated topologies (common.h,
// select_mask: replica r works under topology r / heads, found by the kernel
// from the concatenated row offsets -- no device table to build, no host
// synchronisation): the softmax pair always, SpMM on the panel-resident kernel
// (attention: n = head_dim), SDDMM on both of its kernels (the stationary one
// with one pre-pass for all masks in front), the transpose in its three phases
// (round 4: the grid's last dimension is the mask; a region of tables per mask in
// the workspace).  Other SpMM shapes take one batched launch per mask over its
// heads, ordered on the stream and sharing the workspace.
#include <algorithm>

#include "common.h"

namespace sputnik_hip {
int softmax_many_mask(bool backward, int m, int width, int largest_nonzeros, int replicas, int heads,
                      const float* a, int64_t a_stride, const float* b, int64_t b_stride,
                      const int* row_offsets, float* out, int64_t out_stride, float scale,
                      hipStream_t stream, int masks, const int* mask_nonzeros);
size_t sddmm_tiled_workspace_bytes(int m, int k, int n, int nonzeros, bool summed);
bool spmm_panel_applicable(int m, int k, int n, int nonzeros, const float* dense,
                           int64_t dense_stride, const float* out, int64_t out_stride);
int spmm_panel_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                      const float* values, int64_t values_stride, const int* row_offsets,
                      const int* column_indices, const float* dense, int64_t dense_stride,
                      float* out, int64_t out_stride, hipStream_t stream, Epilogue epi,
                      const int* value_permutation, int block_rows, int mask_heads);
size_t csr_transpose_many_bytes(int masks, int m, int n, int largest_nonzeros);
int csr_transpose_many_launch(int masks, int m, int n, int largest_nonzeros, int64_t total_nonzeros,
                              int heads, const float* values, int64_t values_stride,
                              const int* row_offsets, const int* column_indices, float* out_values,
                              int64_t out_values_stride, int* out_row_offsets,
                              int* out_column_indices, int* out_permutation, void* workspace,
                              size_t workspace_bytes, hipStream_t stream);
}  // namespace sputnik_hip

extern "C" int sputnik_hip_internal_sddmm_many_mask(
    int masks, int m, int k, int n, const int* nonzeros, int largest, int replicas,
    const int* row_indices, const int* row_offsets, const int* column_indices, const float* lhs,
    int64_t lhs_stride, const float* rhs, int64_t rhs_stride, float* out, int64_t out_stride,
    void* workspace, size_t workspace_bytes, hipStream_t stream);

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

inline int largest_of(int masks, const int* nonzeros) {
  return masks > 0 ? *std::max_element(nonzeros, nonzeros + masks) : 0;
}

}  // namespace

extern "C" {

size_t sputnik_hip_sddmm_many_mask_workspace_bytes(int masks, int m, int k, int n,
                                                   int largest_nonzeros) {
  if (masks <= 0 || m <= 0 || k <= 0 || n <= 0 || largest_nonzeros <= 0) return 0;
  // (= sddmm.hip's sddmm_many_mask_plan_bytes: a region ends in the spare word that keeps the
  // masks' start order)
  const size_t tables = sddmm_tiled_workspace_bytes(m, k, n, largest_nonzeros, false);
  if (tables == 0) return 0;
  const size_t one = (tables + sizeof(int) + 255) / 256 * 256;
  return one * static_cast<size_t>(masks);
}

size_t sputnik_hip_csr_transpose_many_mask_workspace_bytes(int masks, int m, int n,
                                                           int largest_nonzeros) {
  if (masks <= 0 || m <= 0 || n <= 0 || largest_nonzeros <= 0) return 0;
  return std::max(csr_transpose_many_bytes(masks, m, n, largest_nonzeros),
                  sputnik_hip_csr_transpose_workspace_bytes(m, n, largest_nonzeros));
}

int sputnik_hip_spmm_many_mask(int masks, int m, int k, int n, const int* nonzeros, int replicas,
                               const int* row_indices, const float* values,
                               int64_t values_stride, const int* row_offsets,
                               const int* column_indices, const float* dense,
                               int64_t dense_stride, float* out, int64_t out_stride,
                               void* workspace, size_t workspace_bytes,
                               sputnik_hip_stream_t stream) {
  const MaskWalk w = check(masks, m, n, nonzeros, replicas);
  if (!w.ok || k < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (masks == 0 || m == 0 || n == 0) return 0;
  // the panel-resident kernel (k up to two panels: attention's P.V and its gradients)
  // serves all masks in one launch
  const int largest = largest_of(masks, nonzeros);
  if (masks > 1 && largest > 0 && k <= 1024 && values_stride >= largest &&
      spmm_panel_applicable(m, k, n, largest, dense, dense_stride, out, out_stride))
    return spmm_panel_launch(m, k, n, largest, replicas, row_indices, values, values_stride,
                             row_offsets, column_indices, dense, dense_stride, out, out_stride,
                             stream, Epilogue{}, nullptr, 0, w.heads);
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
  if (masks == 0 || m == 0) return 0;
  const int largest = largest_of(masks, nonzeros);
  if (largest == 0) return 0;
  if (out_stride < largest) return SPUTNIK_HIP_INVALID_ARGUMENT;
  return sputnik_hip_internal_sddmm_many_mask(masks, m, k, n, nonzeros, largest, replicas,
                                              row_indices, row_offsets, column_indices, lhs, lhs_stride, rhs, rhs_stride, out, out_stride,
                         workspace, workspace_bytes, stream);
}

int sputnik_hip_sparse_softmax_many_mask(int masks, int m, const int* nonzeros, int replicas,
                                         const float* values, int64_t values_stride,
                                         const int* row_indices, const int* row_offsets,
                                         const int* column_indices, float scale, float* out,
                                         int64_t out_stride, sputnik_hip_stream_t stream) {
  const MaskWalk w = check(masks, m, 0, nonzeros, replicas);
  if (!w.ok) return SPUTNIK_HIP_INVALID_ARGUMENT;
  (void)row_indices;
  (void)column_indices;
  if (masks == 0 || m == 0) return 0;
  const int largest = largest_of(masks, nonzeros);
  // value rows are [replicas][width]: the row stride is what may be read of a row
  const int width = static_cast<int>(std::min<int64_t>(values_stride, out_stride));
  if (width < largest) return SPUTNIK_HIP_INVALID_ARGUMENT;
  return softmax_many_mask(false, m, width, largest, replicas, w.heads, values, values_stride,
                           nullptr, 0, row_offsets, out, out_stride, scale, stream, masks, nonzeros);
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
  if (masks == 0 || m == 0) return 0;
  const int largest = largest_of(masks, nonzeros);
  const int width = static_cast<int>(std::min<int64_t>(std::min<int64_t>(out_stride, grad_out_stride),
                                                       grad_values_stride));
  if (width < largest) return SPUTNIK_HIP_INVALID_ARGUMENT;
  return softmax_many_mask(true, m, width, largest, replicas, w.heads, softmax_out, out_stride,
                           grad_out, grad_out_stride, row_offsets, grad_values, grad_values_stride,
                           scale, stream, masks, nonzeros);
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
  if (masks > 1 && m > 0 && n > 0) {
    const bool with_values = values != nullptr && out_values != nullptr;
    const int largest = largest_of(masks, nonzeros);
    if (!with_values || (values_stride >= largest && out_values_stride >= largest)) {
      int64_t total = 0;
      for (int i = 0; i < masks; ++i) total += nonzeros[i];
      const int st = csr_transpose_many_launch(
          masks, m, n, largest, total, with_values ? w.heads : 0, with_values ? values : nullptr,
          values_stride, row_offsets, column_indices, with_values ? out_values : nullptr,
          out_values_stride, out_row_offsets, out_column_indices, out_permutation, workspace,
          workspace_bytes, stream);
      if (st >= 0) return st;
    }
  }
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
