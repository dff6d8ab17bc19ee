// Matrix-core (MFMA) routes of the HALF-storage extension (gfx950).
//
// The reference's five operators are float32 (data_ptr<float>(), src/sddmm_cuda.cu:48-53)
// and stay on the vector kernels.  float16 / bfloat16 operands are this library's
// extension (include/sputnik_hip.h, *_typed): their products are exact in float32, so a
// product with a long reduction over a mask that occupies every tile -- the gradient of
// sparse weights shared by a batch, modules/sparse_linear.py:44-49 -- is a sampled DENSE
// contraction and runs on v_mfma_f32_32x32x16_{f16,bf16}.
#pragma once

#include "common.h"

namespace sputnik_hip {

// out[p] = sum_r < lhs_r[i_p, 0:k], rhs_r[j_p, 0:k] >, operands stored as `in_type`
// (SPUTNIK_HIP_F16 / BF16), float32 sums.  `shape_only`: the answer for a workspace /
// scratch query, which has no pointers to look at.
bool sddmm_mfma_applicable(int m, int k, int n, int nonzeros, int replicas, const void* lhs,
                           int64_t lhs_stride, const void* rhs, int64_t rhs_stride);
bool sddmm_mfma_shape(int m, int k, int n, int nonzeros, int replicas);
// Workgroups that share one output tile (each reduces a contiguous range of the
// (replica, k-step) pairs and writes its own partial vector); 1: straight into `out`.
int sddmm_mfma_splits(int m, int k, int n, int replicas, int planes = 1);
// The plan (topology only, one small launch): per row, where its entries cross the tile
// columns, and whether its columns ascend.  Optional: without it (plan == nullptr), and for
// every tile with a row whose columns do not ascend, the kernel finds a tile's entries by
// walking its rows' entries.
size_t sddmm_mfma_plan_bytes(int m, int n);
int sddmm_mfma_plan(int m, int n, const int* row_offsets, const int* column_indices, void* plan,
                    hipStream_t stream);
// partials: [splits][nonzeros] float32 (== out when splits is 1).
int sddmm_mfma_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_offsets,
                      const int* column_indices, const void* lhs, int64_t lhs_stride,
                      const void* rhs, int64_t rhs_stride, int in_type, float* partials,
                      int splits, const void* plan, hipStream_t stream, int planes = 1,
                      int64_t lhs_plane_stride = 0, int64_t rhs_plane_stride = 0);
// `count` float32 values (a multiple of 4) as sddmm_mfma_planes_of(half_type) planes of
// `half_type`, `count` elements apart, whose (scaled) sum is the value: how an operand that
// arrives as float32 enters the half-operand product without being rounded to the storage
// type (planes = that number above, the operand's plane stride = count).
int sddmm_mfma_planes_of(int half_type);
int sddmm_mfma_split_planes(int64_t count, const float* in, int half_type, void* planes,
                            hipStream_t stream);

// out[i] = partials[0][i] + ... + partials[parts - 1][i], index order (sddmm.hip)
int sum_partial_vectors(int nonzeros, int parts, const float* partials, float* out, hipStream_t stream);

// left_spmm (values shared by the replicas) as a dense contraction: the densified weight
// against the dense operand [replicas][k][n] on half tiles of `tile_type` (spmm_mfma.hip).
// A float32 operand enters as half planes (not rounded).  Workspace: the densified
// weight's planes, then a float32 dense operand's.
bool spmm_mfma_shape(int m, int k, int n, int nonzeros, int replicas, int values_type,
                     int dense_type, int tile_type);
size_t spmm_mfma_workspace_bytes(int m, int k, int n, int replicas, int values_type, int dense_type,
                                 int tile_type);
int spmm_mfma_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_offsets,
                     const int* column_indices, const void* values, int values_type,
                     const void* dense, int dense_type, int64_t dense_stride, int tile_type,
                     const float* bias, int relu, float* out, int64_t out_stride, void* workspace,
                     hipStream_t stream);

}  // namespace sputnik_hip
