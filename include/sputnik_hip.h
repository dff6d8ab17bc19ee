/*
 * sputnik_hip.h -- C ABI of the MI355X (gfx950) sparse kernel library
 * (libsputnik_hip.so).
 *
 * This is the drop-in boundary for the device side of the torch_sputnik
 * operator surface.  Each entry point replaces one library call the
 * reference's host wrappers make (file:line below are in the reference
 * checkout); the host side above it (torch_sputnik_amd/csrc/torch_binding.cpp)
 * mirrors the reference's five operators, src/sputnik.cpp:36-42.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     its name says otherwise; no torch types.
 *   - all launches are asynchronous on `stream`; no entry point synchronises,
 *     allocates or frees device memory, so calls can be captured into a
 *     hipGraph.  Scratch memory is passed in by the caller (query its size
 *     with the matching *_workspace_bytes function), in the way the reference
 *     drives cusparseCsr2cscEx2_bufferSize (src/transpose_cuda.cu:22-31).
 *   - return value: 0 on success, otherwise the hipError_t of the failed
 *     launch, or SPUTNIK_HIP_INVALID_ARGUMENT for a shape the library
 *     rejects.  (The reference returns cudaError_t and aborts on != success,
 *     include/error_check.h:5-10.)
 *   - fp32 values, int32 indices, row-major dense operands, zero-based CSR:
 *     exactly the reference's layout (src/spmm_cuda.cu:49-56).
 *   - every output element is written by the kernels (rows without nonzeros
 *     get zeros), so outputs need no memset (the reference zero-fills them
 *     first, src/spmm_cuda.cu:46).
 *   - `row_indices` is a permutation of [0,m) giving the order in which rows
 *     are dealt to workgroups (load balancing); results never depend on it.
 *   - batched entry points run all replicas in ONE launch; a stride is the
 *     element distance between consecutive replicas (0 = operand shared).
 */
#ifndef SPUTNIK_HIP_H_
#define SPUTNIK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque HIP stream handle (same type as hipStream_t in <hip/hip_runtime_api.h>). */
typedef struct ihipStream_t* sputnik_hip_stream_t;

#define SPUTNIK_HIP_INVALID_ARGUMENT (-1)
/* a fused entry point does not serve this shape / alignment: use the separate operators */
#define SPUTNIK_HIP_UNSUPPORTED (-2)

/* Exported-symbol marker (the library is built with -fvisibility=hidden). */
#define SPUTNIK_HIP_API __attribute__((visibility("default")))

/* Storage types of the *_typed entry points (round 3).  The reference is float32
 * only (data_ptr<float>(), src/spmm_cuda.cu:51); float16 / bfloat16 operands are an
 * extension (BASELINE.json config 5) that the kernels read and write directly --
 * all arithmetic stays float32. */
#define SPUTNIK_HIP_F32 0
#define SPUTNIK_HIP_F16 1
#define SPUTNIK_HIP_BF16 2

/* Library / build identification: "sputnik_hip <version> gfx950". */
SPUTNIK_HIP_API const char* sputnik_hip_version(void);

/* Hash of the kernel sources this library was built from (12 hex digits).  The
 * profiles under profiles/ record it, and bench.py quotes a PMC traffic figure
 * only when it was collected on the build that is running. */
SPUTNIK_HIP_API const char* sputnik_hip_build_id(void);

/* Name of the device kernel that serves an SpMM call of this shape (the
 * dispatcher's choice: spmm.hip, spmm_tiled.hip), e.g. "spmm_flat_kernel<0>"; for
 * benchmarks and profiles, which report the dominant kernel by name.  The
 * pointer is to a static string. */
SPUTNIK_HIP_API const char* sputnik_hip_spmm_kernel_name(int m, int k, int n, int nonzeros,
                                                         int replicas);
/* The same for an SDDMM call (sddmm.hip, sddmm_tiled.hip, sddmm_flat.hip): operands of
 * `elem_bytes` (4: float32, 2: float16 / bfloat16), `planned` != 0 for a product on a
 * workspace that sputnik_hip_sddmm_plan filled ("sddmm_flat_kernel", "sddmm_quad_kernel",
 * "sddmm_stationary_kernel", "sddmm_rowwave_kernel"). */
SPUTNIK_HIP_API const char* sputnik_hip_sddmm_kernel_name(int m, int k, int n, int nonzeros,
                                                          int replicas, int elem_bytes,
                                                          int planned);

/* ------------------------------------------------------------------------
 * SpMM   C[m,n] = A_csr[m,k] * B[k,n]
 * replaces sputnik::CudaSpmm(m,k,n,nnz,row_indices,values,row_offsets,
 *          column_indices,dense,out,stream)          src/spmm_cuda.cu:49-56
 * ---------------------------------------------------------------------- */
SPUTNIK_HIP_API int sputnik_hip_spmm(int m, int k, int n, int nonzeros,
                     const int* row_indices, const float* values,
                     const int* row_offsets, const int* column_indices,
                     const float* dense, float* out,
                     sputnik_hip_stream_t stream);

/*
 * Batched SpMM, one launch for all replicas.  Replaces the host loops
 *   src/spmm_cuda.cu:48-57               (values_stride = nonzeros)
 *   src/left_replicated_spmm.cu:32-41    (values_stride = 0: shared weights)
 * dense_stride is normally k*n and out_stride m*n.
 * `workspace` may be NULL (then only the workspace-free kernels are used);
 * otherwise it must hold sputnik_hip_spmm_workspace_bytes(m,k,n,nonzeros).
 */
SPUTNIK_HIP_API size_t sputnik_hip_spmm_workspace_bytes(int m, int k, int n, int nonzeros);

SPUTNIK_HIP_API int sputnik_hip_spmm_batched(int m, int k, int n, int nonzeros, int replicas,
                             const int* row_indices, const float* values,
                             int64_t values_stride, const int* row_offsets,
                             const int* column_indices, const float* dense,
                             int64_t dense_stride, float* out,
                             int64_t out_stride, void* workspace,
                             size_t workspace_bytes,
                             sputnik_hip_stream_t stream);

/*
 * The same in two steps, for callers whose sparsity pattern is static (layer
 * weights, attention masks): `plan` runs the topology-only pre-pass into the
 * workspace once, `batched_planned` may then be called any number of times
 * with that workspace (values, dense and out may change between calls; the
 * three index arrays, m, k and n may not).  No counterpart in the reference,
 * which re-derives everything per call (src/spmm_cuda.cu:48-57).
 * A planned workspace is also the call's SCRATCH (a single product against a narrow dense
 * operand keeps the partial tiles of its K split there): calls that share one workspace
 * must be ordered on one stream; concurrent streams take a workspace each (the plan is
 * topology only: `plan` may be run into any number of them).
 */
/* (Round 4) A single product against a narrow dense operand (n < 512, k >= 2048, one
 * replica) deals its K chunks to several workgroups per tile; their partial tiles live in
 * the workspace (sputnik_hip_spmm_workspace_bytes counts them) and are scratch of the
 * call: one workspace, planned or not, serves one stream at a time for such a shape. */
SPUTNIK_HIP_API int sputnik_hip_spmm_plan(int m, int k, int n, int nonzeros,
                          const int* row_indices, const int* row_offsets,
                          const int* column_indices, void* workspace,
                          size_t workspace_bytes, sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_spmm_batched_planned(int m, int k, int n, int nonzeros,
                             int replicas, const int* row_indices,
                             const float* values, int64_t values_stride,
                             const int* row_offsets, const int* column_indices,
                             const float* dense, int64_t dense_stride, float* out,
                             int64_t out_stride, const void* workspace,
                             size_t workspace_bytes, sputnik_hip_stream_t stream);

/* ------------------------------------------------------------------------
 * SDDMM  out[p] = < lhs[i_p, 0:k], rhs[j_p, 0:k] >  for each stored (i_p,j_p)
 * lhs is [m,k], rhs is [n,k] (row-major, i.e. already "transposed").
 * replaces sputnik::CudaSddmm(m,k,n,nnz,row_indices,row_offsets,
 *          column_indices,lhs,rhs,out,stream)       src/sddmm_cuda.cu:46-53
 * and the host loop src/sddmm_cuda.cu:45-54 (batched form).
 * ---------------------------------------------------------------------- */
SPUTNIK_HIP_API int sputnik_hip_sddmm(int m, int k, int n, int nonzeros,
                      const int* row_indices, const int* row_offsets,
                      const int* column_indices, const float* lhs,
                      const float* rhs, float* out,
                      sputnik_hip_stream_t stream);

/* `workspace` may be NULL (workspace-free kernel only); otherwise it must hold
 * sputnik_hip_sddmm_workspace_bytes(m,k,n,nonzeros) (0 = none needed). */
SPUTNIK_HIP_API size_t sputnik_hip_sddmm_workspace_bytes(int m, int k, int n, int nonzeros);

SPUTNIK_HIP_API int sputnik_hip_sddmm_batched(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_indices, const int* row_offsets,
                              const int* column_indices, const float* lhs,
                              int64_t lhs_stride, const float* rhs,
                              int64_t rhs_stride, float* out,
                              int64_t out_stride, void* workspace,
                              size_t workspace_bytes, sputnik_hip_stream_t stream);

/* Two-step form for static masks (as sputnik_hip_spmm_plan): `plan` runs the
 * topology-only pre-pass into the workspace once, `batched_planned` reuses it.
 * No counterpart in the reference (src/sddmm_cuda.cu:45-54 re-derives per call). */
SPUTNIK_HIP_API int sputnik_hip_sddmm_plan(int m, int k, int n, int nonzeros,
                              const int* row_indices, const int* row_offsets,
                              const int* column_indices, void* workspace,
                              size_t workspace_bytes, sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sddmm_batched_planned(int m, int k, int n, int nonzeros,
                              int replicas, const int* row_indices, const int* row_offsets,
                              const int* column_indices, const float* lhs,
                              int64_t lhs_stride, const float* rhs, int64_t rhs_stride,
                              float* out, int64_t out_stride, const void* workspace,
                              size_t workspace_bytes, sputnik_hip_stream_t stream);

/*
 * SDDMM on operands stored as `in_type` (SPUTNIK_HIP_F32 / F16 / BF16; lhs and rhs
 * alike; strides in elements), output stored as `out_type` (F32, or in_type).  Half
 * operands are read as they are -- the LDS slab of rhs rows holds half values (half
 * the staging and half the LDS traffic of the float form), products are exact
 * (v_dot2_f32_f16 / _bf16), sums float32 -- where src/sddmm_cuda.cu:48-53 knows float
 * only.  workspace / planned as sputnik_hip_sddmm_batched{,_planned} (same size, same
 * plan: it does not depend on the types).  Returns SPUTNIK_HIP_UNSUPPORTED for a half
 * OUTPUT whose product needs more than one pass over the output (k wider than one
 * panel): take a float output then.
 */
/*
 * SpMM / left_spmm (values_stride = 0) with operands stored as float32, float16 or
 * bfloat16 -- values and dense independently, a half pair of ONE kind -- product
 * float32, optional bias / ReLU epilogue as sputnik_hip_spmm_bias_batched.  Half
 * operands are read as they are (src/spmm_cuda.cu:42,51 knows float only): the
 * panel-resident kernel widens the rows of B on their way into LDS (half the bytes
 * from memory, same compute loop) and serves up to two panels (k <= 1024) with
 * n >= 64; larger products, where the chunked float kernels win by a factor of two,
 * are widened into the workspace by one pass inside this call when the workspace has
 * sputnik_hip_spmm_typed_workspace_bytes; everything else (and every call without
 * that room) takes the row-gather kernel, which reads half rows of B from L2.
 */
SPUTNIK_HIP_API size_t sputnik_hip_spmm_typed_workspace_bytes(int m, int k, int n, int nonzeros,
                              int replicas, int values_type, int64_t values_stride,
                              int dense_type);
SPUTNIK_HIP_API int sputnik_hip_spmm_typed(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_indices, const void* values, int values_type,
                              int64_t values_stride, const int* row_offsets,
                              const int* column_indices, const void* dense, int dense_type,
                              int64_t dense_stride, const float* bias, int relu, float* out,
                              int64_t out_stride, void* workspace, size_t workspace_bytes,
                              sputnik_hip_stream_t stream);

/*
 * left_spmm (values shared by the replicas: src/left_replicated_spmm.cu:32-41) as a DENSE
 * contraction on the matrix cores, for the half-storage extension (round 5,
 * csrc/spmm_mfma.hip): the CSR values are scattered into a zeroed [m, k] image of
 * `tile_type` (SPUTNIK_HIP_F16 / BF16) and multiplied against the dense operand
 * [replicas][k][n] on v_mfma_f32_32x32x16 tiles, float32 sums, float32 product with the
 * bias / ReLU epilogue of sputnik_hip_spmm_bias_batched.  At layer densities every tile of
 * the weight is occupied, and the tiles cost 1 / density times the sparse flops on a unit
 * sixteen times faster than the vector pipe.  An operand given as float32 (values_type /
 * dense_type = SPUTNIK_HIP_F32; the other, if half, has type tile_type) is NOT rounded to the
 * tile type: it enters as half planes whose sum is the value (float16: 22 bits, bfloat16:
 * 24), at one more tile product per plane pair.  sputnik_hip_spmm_typed takes this route by
 * itself for a half dense operand; this entry adds the float32 dense operand (the incoming
 * gradient of modules/sparse_linear.py:60-65 in a half-storage layer).
 * Returns SPUTNIK_HIP_UNSUPPORTED where the route does not serve the call (k not a
 * multiple of 64, n not of 8, a grid under 192 tiles, density under 0.06 per tile
 * product, too little workspace, ...): take sputnik_hip_spmm_typed / _batched then.
 * The float32 operators never come here.
 */
SPUTNIK_HIP_API size_t sputnik_hip_left_spmm_half_tiles_workspace_bytes(int m, int k, int n,
                              int nonzeros, int replicas, int values_type, int dense_type,
                              int tile_type);
SPUTNIK_HIP_API int sputnik_hip_left_spmm_half_tiles(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_offsets, const int* column_indices,
                              const void* values, int values_type, const void* dense,
                              int dense_type, int64_t dense_stride, int tile_type,
                              const float* bias, int relu, float* out, int64_t out_stride,
                              void* workspace, size_t workspace_bytes,
                              sputnik_hip_stream_t stream);

/*
 * A sparse layer on half-stored activations, all three products on the matrix cores with NO
 * layout pass (round 5, csrc/sparse_linear_half.hip, csrc/mfma_gemm.h).  The reference's
 * SparseLinear runs x.transpose(1, 2).contiguous() in front of left_spmm and the
 * transposes back behind it (modules/sparse_linear.py:28,44-65,89); the tile kernel reads
 * every operand in the layout the caller has it:
 *   forward          y[b][o][s]  = sum_i W[o][i] x[b][s][i]            y float32 [batch, out, seq]
 *   weight gradient  dW[p]       = sum_b sum_s dy[b][o_p][s] x[b][s][i_p]   float32 [nonzeros]
 *   input gradient   dx[b][s][i] = sum_o dy[b][o][s] W[o][i]           [batch, seq, in], float32 or tile type
 * x [batch, seq, in] in `tile_type` (SPUTNIK_HIP_F16 / BF16).  W is given as its IMAGE: the
 * CSR values (float32 or tile_type) scattered into a zeroed [planes][out][in] array of the
 * tile type -- sputnik_hip_sparse_linear_half_image, one launch per step, shared
 * by the forward pass and the input gradient.  dy [batch, out, seq] is given either in the
 * tile type (grad_type = tile_type) or as the PLANES of the float32 tensor
 * (sputnik_hip_half_planes, grad_type = SPUTNIK_HIP_F32): float32 operands are never rounded
 * to the storage type -- they enter as half planes whose sum is the value (float16: two
 * planes, 22 bits; bfloat16: three, 24 bits).  The weight gradient takes a plan
 * (sputnik_hip_sparse_linear_half_plan, topology only, may be NULL) and scratch for the
 * partial vectors of the workgroups that share a tile.
 * sputnik_hip_sparse_linear_half_supported: 1 where the three products take this route
 * (out, in, seq multiples of 64, a grid of at least 192 tiles, density from 0.06 per plane of
 * the values); an entry point returns SPUTNIK_HIP_UNSUPPORTED for a call it does not serve
 * (bfloat16 tiles with BOTH float32 values and a float32 dy in the input gradient: nine
 * plane pairs) -- use the typed operators then.  The float32 operators never come here.
 */
SPUTNIK_HIP_API int sputnik_hip_sparse_linear_half_supported(int out_features, int in_features, int seq,
                              int batch, int nonzeros, int values_type, int tile_type);
SPUTNIK_HIP_API size_t sputnik_hip_sparse_linear_half_image_bytes(int out_features, int in_features,
                              int values_type, int tile_type);
SPUTNIK_HIP_API int sputnik_hip_sparse_linear_half_image(int out_features, int in_features, int nonzeros,
                              const int* row_offsets, const int* column_indices, const void* values,
                              int values_type, int tile_type, void* image, size_t image_bytes,
                              sputnik_hip_stream_t stream);
SPUTNIK_HIP_API size_t sputnik_hip_half_planes_bytes(int64_t count, int tile_type);
SPUTNIK_HIP_API int sputnik_hip_half_planes(int64_t count, const float* in, int tile_type, void* planes,
                              sputnik_hip_stream_t stream);
SPUTNIK_HIP_API int sputnik_hip_sparse_linear_half_forward(int out_features, int in_features, int seq,
                              int batch, const void* image, int values_type, const void* x,
                              int tile_type, const float* bias, int relu, float* y,
                              sputnik_hip_stream_t stream);
SPUTNIK_HIP_API size_t sputnik_hip_sparse_linear_half_plan_bytes(int out_features, int in_features);
SPUTNIK_HIP_API int sputnik_hip_sparse_linear_half_plan(int out_features, int in_features,
                              const int* row_offsets, const int* column_indices, void* plan,
                              sputnik_hip_stream_t stream);
SPUTNIK_HIP_API size_t sputnik_hip_sparse_linear_half_scratch_bytes(int out_features, int in_features,
                              int seq, int batch, int nonzeros, int grad_type, int tile_type);
SPUTNIK_HIP_API int sputnik_hip_sparse_linear_half_weight_gradient(int out_features, int in_features,
                              int seq, int batch, int nonzeros, const int* row_offsets,
                              const int* column_indices, const void* grad_output, int grad_type,
                              const void* x, int tile_type, float* grad_values, const void* plan,
                              void* scratch, size_t scratch_bytes, sputnik_hip_stream_t stream);
SPUTNIK_HIP_API int sputnik_hip_sparse_linear_half_input_gradient(int out_features, int in_features,
                              int seq, int batch, const void* grad_output, int grad_type,
                              const void* image, int values_type, int tile_type, void* grad_input,
                              int grad_input_type, sputnik_hip_stream_t stream);

/* sum over the replicas (sputnik_hip_sddmm_sum_batched{,_planned}) on operands stored as
 * `in_type`; the partial vectors and the result are float32.  Workspace / scratch sizes as
 * the float form's (sputnik_hip_sddmm_sum_workspace_bytes / _scratch_bytes).
 * Round 5: half operands with a long reduction (replicas * k >= 1024) over a mask of
 * density >= 0.05 (m, n >= 128, k a multiple of 64) take the MATRIX CORES
 * (csrc/sddmm_mfma.hip): the product is then a sampled dense contraction -- every 128 x 128
 * tile of the mask is occupied -- and the half products are exact in float32 either way.
 * The float32 form stays on the vector kernels. */
SPUTNIK_HIP_API int sputnik_hip_sddmm_sum_typed(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_indices, const int* row_offsets,
                              const int* column_indices, const void* lhs, int64_t lhs_stride,
                              const void* rhs, int64_t rhs_stride, int in_type, float* out,
                              void* workspace, size_t workspace_bytes, int planned,
                              void* scratch, size_t scratch_bytes, sputnik_hip_stream_t stream);

/* The same for a (float32, half) pair of operands -- the weight gradient of a layer whose
 * activations are stored in half precision while the incoming gradient is float32
 * (modules/sparse_linear.py:44-49 with the extension's storage types).  Where the
 * matrix-core route serves the shape (long reduction, mask density from 0.05: the dense
 * 128 x 128 tiles on v_mfma_f32_32x32x16_{f16,bf16}, sampled at the mask) the float32
 * operand is NOT rounded to the storage type: one pass splits it into two half planes,
 * v = hi + lo, and the tiles accumulate hi * x + lo * x (exact products, float32 sums).
 * `scratch` holds sputnik_hip_sddmm_sum_mixed_scratch_bytes(...) bytes (the planes, then the
 * partial vectors); workspace / planned as sputnik_hip_sddmm_sum_typed.  Returns
 * SPUTNIK_HIP_UNSUPPORTED for every other shape and for a float32 operand whose replicas do
 * not lie back to back: widen the half operand and take sputnik_hip_sddmm_sum_typed then.
 * lhs_type == rhs_type is sputnik_hip_sddmm_sum_typed itself. */
SPUTNIK_HIP_API size_t sputnik_hip_sddmm_sum_mixed_scratch_bytes(int m, int k, int n, int nonzeros,
                              int replicas, int lhs_type, int rhs_type);
SPUTNIK_HIP_API int sputnik_hip_sddmm_sum_mixed(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_indices, const int* row_offsets,
                              const int* column_indices, const void* lhs, int lhs_type,
                              int64_t lhs_stride, const void* rhs, int rhs_type,
                              int64_t rhs_stride, float* out, void* workspace,
                              size_t workspace_bytes, int planned, void* scratch,
                              size_t scratch_bytes, sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sddmm_typed(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_indices, const int* row_offsets,
                              const int* column_indices, const void* lhs, int64_t lhs_stride,
                              const void* rhs, int64_t rhs_stride, int in_type, void* out,
                              int64_t out_stride, int out_type, void* workspace,
                              size_t workspace_bytes, int planned, sputnik_hip_stream_t stream);

/* Sum of the replicas' products, out[nonzeros] = sum_r sddmm(lhs_r, rhs_r): the
 * gradient of sparse values shared by a batch.  The reference returns the
 * [replicas, nonzeros] products (tests/test_linear_3d.py:64-69,
 * tests/test_left_spmm.py:57-64) and leaves the sum over the batch to autograd;
 * here the (replica, k-panel) pairs of the tiled kernel run in ONE launch into
 * `scratch` and a second kernel adds them up in index order (deterministic).
 * `scratch` holds sputnik_hip_sddmm_sum_scratch_bytes(...) bytes (0: not
 * needed), 16-byte aligned.  `workspace` holds
 * sputnik_hip_sddmm_sum_workspace_bytes(...) bytes; the planned form takes one
 * that sputnik_hip_sddmm_sum_plan filled -- NOT a plan of sputnik_hip_sddmm_plan:
 * with its panels side by side the summed product may cut k (and with it the
 * mask's column slabs) differently from the plain one. */
SPUTNIK_HIP_API size_t sputnik_hip_sddmm_sum_workspace_bytes(int m, int k, int n, int nonzeros);

SPUTNIK_HIP_API int sputnik_hip_sddmm_sum_plan(int m, int k, int n, int nonzeros,
                              const int* row_indices, const int* row_offsets,
                              const int* column_indices, void* workspace,
                              size_t workspace_bytes, sputnik_hip_stream_t stream);

SPUTNIK_HIP_API size_t sputnik_hip_sddmm_sum_scratch_bytes(int m, int k, int n, int nonzeros,
                                                           int replicas);

SPUTNIK_HIP_API int sputnik_hip_sddmm_sum_batched(int m, int k, int n, int nonzeros,
                              int replicas, const int* row_indices, const int* row_offsets,
                              const int* column_indices, const float* lhs,
                              int64_t lhs_stride, const float* rhs, int64_t rhs_stride,
                              float* out, void* workspace, size_t workspace_bytes,
                              void* scratch, size_t scratch_bytes,
                              sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sddmm_sum_batched_planned(int m, int k, int n, int nonzeros,
                              int replicas, const int* row_indices, const int* row_offsets,
                              const int* column_indices, const float* lhs,
                              int64_t lhs_stride, const float* rhs, int64_t rhs_stride,
                              float* out, const void* workspace, size_t workspace_bytes,
                              void* scratch, size_t scratch_bytes,
                              sputnik_hip_stream_t stream);

/* Up to four summed SDDMMs of ONE shape (m, k, n), replica count and operand strides in
 * one call -- the weight gradients of a group of projections that share their input
 * (modules/sparse_attention.py:108-110: dW_q, dW_k, dW_v = dY_q, dY_k, dY_v against the same
 * x): each problem runs as sputnik_hip_sddmm_sum_batched_planned does, its own planned
 * workspace and scratch, and ONE launch adds the partial vectors of all of them (a sum is
 * four microseconds of launch and little else).  Results are those of the single calls,
 * bit for bit. */
typedef struct sputnik_hip_sddmm_sum_problem {
  const int* row_indices;
  const int* row_offsets;
  const int* column_indices;
  const float* lhs;        /* [replicas][m][k] */
  const float* rhs;        /* [replicas][n][k] */
  float* out;              /* [nonzeros] */
  const void* workspace;   /* planned: sputnik_hip_sddmm_sum_plan */
  size_t workspace_bytes;
  void* scratch;           /* sputnik_hip_sddmm_sum_scratch_bytes */
  size_t scratch_bytes;
  int nonzeros;
} sputnik_hip_sddmm_sum_problem;

SPUTNIK_HIP_API int sputnik_hip_sddmm_sum_group_planned(int m, int k, int n, int replicas,
                              int count, const sputnik_hip_sddmm_sum_problem* problems,
                              int64_t lhs_stride, int64_t rhs_stride,
                              sputnik_hip_stream_t stream);

/* ------------------------------------------------------------------------
 * Sparse softmax: per CSR row, exp(x - max) / sum(exp(x - max)) over the
 * stored entries.  `n` is unused (the reference passes -1,
 * src/softmax_cuda.cu:22).
 * replaces sputnik::SparseSoftmax(m,n,nnz,values,row_indices,row_offsets,
 *          column_indices,out,stream)               src/softmax_cuda.cu:36-42
 * and the host loop src/softmax_cuda.cu:35-43 (batched form).
 * ---------------------------------------------------------------------- */
SPUTNIK_HIP_API int sputnik_hip_sparse_softmax(int m, int n, int nonzeros, const float* values,
                               const int* row_indices, const int* row_offsets,
                               const int* column_indices, float* out,
                               sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sparse_softmax_batched(int m, int n, int nonzeros, int replicas,
                                       const float* values, int64_t values_stride,
                                       const int* row_indices,
                                       const int* row_offsets,
                                       const int* column_indices, float* out,
                                       int64_t out_stride,
                                       sputnik_hip_stream_t stream);

/*
 * csr_transpose of values stored as `values_type` (SPUTNIK_HIP_F32 / F16 / BF16); the
 * transposed values are float32 (what the products of the backward passes take,
 * modules/sparse_linear.py:52-63).  Half values are read through the permutation by
 * the one gather that moves them: no widened copy of the input.  The half forms need
 * `out_permutation` (nonzeros ints); workspace as sputnik_hip_csr_transpose; `checked`
 * != 0 = sputnik_hip_csr_transpose_checked's behaviour (waits for the stream, reports a
 * pattern the transpose is not defined for).
 */
SPUTNIK_HIP_API int sputnik_hip_csr_transpose_typed(int m, int n, int nonzeros, int replicas,
                              const void* values, int values_type, int64_t values_stride,
                              const int* row_offsets, const int* column_indices,
                              float* out_values, int64_t out_values_stride,
                              int* out_row_offsets, int* out_column_indices,
                              int* out_permutation, void* workspace, size_t workspace_bytes,
                              int checked, sputnik_hip_stream_t stream);

/* ------------------------------------------------------------------------
 * CSR transpose  CSR(m x n) -> CSR(n x m), stable (source rows ascend within
 * each output row: the ordering of cuSPARSE CSR2CSC_ALG1).
 * replaces cusparseCsr2cscEx2_bufferSize            src/transpose_cuda.cu:22-31
 *      and cusparseCsr2cscEx2                       src/transpose_cuda.cu:90-99
 * `replicas` value arrays (stride values_stride / out_values_stride elements)
 * share one permutation; the reference only ever has replicas = 1.
 * `out_permutation` (may be NULL) receives, for every output slot, the index
 * of the source nonzero, so a caller can reuse a static topology's transpose.
 * Preconditions (as for cusparseCsr2cscEx2 on valid CSR): column indices in
 * [0, n), a row stores a column at most once (order inside a row is free).
 * A violation is DETECTED on the device: the status word (the last 16 bytes of
 * the workspace) is non-zero afterwards, and the outputs are unspecified.
 * sputnik_hip_csr_transpose is asynchronous and leaves the word to the caller;
 * sputnik_hip_csr_transpose_checked (same arguments) waits for the stream and
 * returns SPUTNIK_HIP_INVALID_ARGUMENT.
 * Workspace: the smaller of about 4 * (2 * ceil(m / 32) * n + n) bytes (mask / count
 * tables: 2 MiB at 2048 x 2048) and, where those would exceed eight table entries
 * per nonzero, 4 * (2 n + 3 nnz) bytes (histogram path, O(n + nnz) like the
 * buffer cusparseCsr2cscEx2_bufferSize asks for: 5.4 MiB at 65536^2, density
 * 1e-4, where the tables would be 1 GiB); + 16.
 * ---------------------------------------------------------------------- */
SPUTNIK_HIP_API size_t sputnik_hip_csr_transpose_workspace_bytes(int m, int n, int nonzeros);

SPUTNIK_HIP_API int sputnik_hip_csr_transpose_checked(int m, int n, int nonzeros, int replicas,
                              const float* values, int64_t values_stride,
                              const int* row_offsets, const int* column_indices,
                              float* out_values, int64_t out_values_stride,
                              int* out_row_offsets, int* out_column_indices,
                              int* out_permutation, void* workspace, size_t workspace_bytes,
                              sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_csr_transpose(int m, int n, int nonzeros, int replicas,
                              const float* values, int64_t values_stride,
                              const int* row_offsets, const int* column_indices,
                              float* out_values, int64_t out_values_stride,
                              int* out_row_offsets, int* out_column_indices,
                              int* out_permutation, void* workspace,
                              size_t workspace_bytes,
                              sputnik_hip_stream_t stream);

/* ========================================================================
 * Extensions around the five operators.  The reference binding
 * as committed (src/sputnik.cpp:36-42) has none of these; the call sites that
 * fix their meaning are cited per entry.
 * ====================================================================== */

/*
 * SpMM with a fused epilogue: out[i, :] = act(sum + bias[i]); `bias` ([m],
 * indexed by output row, may be NULL) and relu (0/1) are both optional.
 * Call site: torch_sputnik.spmm_bias(m, k, values, row_indices, row_offsets,
 * column_indices, bias, dense), tests/test_spmm_bias_relu.py:35-37; the
 * SparseLinear callers add the bias in a separate pass
 * (tests/test_linear_3d.py:47).
 */
SPUTNIK_HIP_API int sputnik_hip_spmm_bias_batched(int m, int k, int n, int nonzeros,
                             int replicas, const int* row_indices,
                             const float* values, int64_t values_stride,
                             const int* row_offsets, const int* column_indices,
                             const float* dense, int64_t dense_stride,
                             const float* bias, int relu, float* out,
                             int64_t out_stride, void* workspace,
                             size_t workspace_bytes, sputnik_hip_stream_t stream);

/*
 * SpMM over a topology whose values are stored in ANOTHER order of the same
 * entries: entry p of (row_offsets, column_indices) takes
 * values[value_permutation[p]].  This is the product with the TRANSPOSE of a
 * matrix whose transposed topology and permutation are known (cached), as the
 * backward passes need it (modules/spmm.py:59-66 transposes the values per
 * call): the gather happens inside the kernel, no permuted copy is made.
 * Served by the panel kernel only (k <= 4096, n a multiple of 4 and >= 64,
 * m >= 16, 16-byte aligned operands); anything else returns
 * SPUTNIK_HIP_UNSUPPORTED and the caller permutes first
 * (sputnik_hip_permute_last_batched / _banded_batched) and calls
 * sputnik_hip_spmm_batched.  `_supported` answers 1 where the gather inside the
 * kernel is also the FASTER form: one panel, k <= 512 (with two panels every
 * value is gathered twice: 162 us against 41 + 50 us at the attention shapes).
 */
SPUTNIK_HIP_API int sputnik_hip_spmm_permuted_supported(int m, int k, int n, int nonzeros);

SPUTNIK_HIP_API int sputnik_hip_spmm_permuted_batched(int m, int k, int n, int nonzeros,
                             int replicas, const int* row_indices, const float* values,
                             int64_t values_stride, const int* value_permutation,
                             const int* row_offsets, const int* column_indices,
                             const float* dense, int64_t dense_stride, float* out,
                             int64_t out_stride, sputnik_hip_stream_t stream);

/*
 * SpMM whose product C[m, n] is STORED TRANSPOSED in blocks of `block_rows` rows:
 *   out[(row / block_rows) * n * block_rows + col * block_rows + row % block_rows]
 * per replica, i.e. every block of block_rows rows of C as its transpose
 * [n][block_rows].  With block_rows = head_dim this is the head split that
 * follows every projection (`four_d_to_three_d` on a transposed view,
 * modules/sparse_attention.py:38-45,108-126); with block_rows = m it is C^T, the
 * head merge in front of the output projection -- layout passes that the
 * reference runs as separate strided copies.  The panel kernel writes its
 * 256 x 64 tile through LDS in that order.  `value_permutation` may be NULL or
 * as in sputnik_hip_spmm_permuted_batched; bias / relu as in
 * sputnik_hip_spmm_bias_batched.  Rows are processed in natural order (no
 * row_indices).  Served: what the panel kernel serves, with block_rows a
 * multiple of 64 that divides m and divides or is a multiple of 256; otherwise
 * SPUTNIK_HIP_UNSUPPORTED (the caller runs the product and
 * sputnik_hip_transpose_batched).  `_supported` also requires k <= 1024.
 */
SPUTNIK_HIP_API int sputnik_hip_spmm_transposed_out_supported(int m, int k, int n, int nonzeros,
                                                              int block_rows);

SPUTNIK_HIP_API int sputnik_hip_spmm_transposed_out_batched(int m, int k, int n, int nonzeros,
                             int replicas, const float* values, int64_t values_stride,
                             const int* value_permutation, const int* row_offsets,
                             const int* column_indices, const float* dense,
                             int64_t dense_stride, const float* bias, int relu,
                             int block_rows, float* out, int64_t out_stride,
                             sputnik_hip_stream_t stream);

/*
 * A GROUP of up to four SpMMs of one shape (m, k, n; values shared by the
 * replicas) in one launch: the projections of an attention block
 * (modules/sparse_attention.py:108-126 runs them one after the other).
 *   accumulate = 0: out_p[r] = A_p * dense_p[r] for every problem p.  Problems
 *     that name the same `dense` share one copy of its panel in LDS (the q, k
 *     and v projections of one input); with block_rows > 0 every product is
 *     stored head split as in sputnik_hip_spmm_transposed_out_batched.
 *   accumulate = 1: all problems name the SAME `out`, which receives
 *     sum_p A_p * dense_p[r] (overwritten, not added to): the input gradient
 *     sum_w W_w^T dY_w of those projections, accumulated in registers.
 * value_permutation: NULL in all problems or set in all (as in
 * sputnik_hip_spmm_permuted_batched).  row_indices: the processing order for
 * accumulate = 0 and block_rows = 0; ignored (may be NULL) otherwise.
 * Served: k <= 512, n a multiple of 4 and >= 64, m >= 16, 16-byte aligned
 * operands, at most 4 problems, and one of the combinations
 * (block_rows > 0, no permutation, accumulate = 0), (block_rows = 0, accumulate
 * = 1), (block_rows = 0, no permutation, accumulate = 0); anything else returns
 * SPUTNIK_HIP_UNSUPPORTED and the caller runs the products one by one.
 */
typedef struct sputnik_hip_spmm_problem {
  const int* row_indices;
  const int* row_offsets;
  const int* column_indices;
  const float* values;            /* [nonzeros] */
  const int* value_permutation;   /* NULL, or [nonzeros] */
  const float* dense;             /* [replicas][k][n], dense_stride apart */
  float* out;                     /* [replicas][m][n] (or its blocked transpose) */
  int nonzeros;
} sputnik_hip_spmm_problem;

SPUTNIK_HIP_API int sputnik_hip_spmm_group_supported(int m, int k, int n, int count,
                                                     int block_rows, int accumulate);

SPUTNIK_HIP_API int sputnik_hip_spmm_group_batched(int m, int k, int n, int replicas, int count,
                             const sputnik_hip_spmm_problem* problems,
                             int64_t dense_stride, int64_t out_stride, int block_rows,
                             int accumulate, sputnik_hip_stream_t stream);

/*
 * softmax(scale * x) per CSR row: folds the 1/sqrt(d) of
 * modules/sparse_attention.py:72 into the softmax pass.
 */
SPUTNIK_HIP_API int sputnik_hip_sparse_softmax_scaled_batched(int m, int n, int nonzeros,
                             int replicas, const float* values, int64_t values_stride,
                             const int* row_indices, const int* row_offsets,
                             const int* column_indices, float scale, float* out,
                             int64_t out_stride, sputnik_hip_stream_t stream);

/*
 * Gradient of y = softmax(scale * x):
 *   grad_values = scale * y * (grad_out - rowsum(grad_out * y)),
 * row sums over the stored entries.  The reference calls the raw op inside
 * attention (modules/sparse_attention.py:76), which cuts the gradient; the
 * intended autograd wrapper is tests/transformer/functions.py:70-120.
 */
SPUTNIK_HIP_API int sputnik_hip_sparse_softmax_backward_batched(int m, int nonzeros,
                             int replicas, const float* softmax_out, int64_t out_stride,
                             const float* grad_out, int64_t grad_out_stride,
                             const int* row_offsets, float scale, float* grad_values,
                             int64_t grad_values_stride, sputnik_hip_stream_t stream);

/*
 * The softmax pair on values stored as `dtype` (SPUTNIK_HIP_F32 / F16 / BF16; inputs
 * and output alike; strides in elements).  Half types move 2 + 2 bytes per entry
 * forward, 2 + 2 + 2 backward -- half the HBM traffic of src/softmax_cuda.cu:38-42 --
 * and are widened / rounded to nearest even in registers; max, exp, sum and the
 * gradient's row dot product are float32.  The float entry points above are
 * dtype = SPUTNIK_HIP_F32 of these.
 */
SPUTNIK_HIP_API int sputnik_hip_sparse_softmax_typed(int m, int n, int nonzeros, int replicas,
                             const void* values, int64_t values_stride, const int* row_indices,
                             const int* row_offsets, const int* column_indices, float scale,
                             void* out, int64_t out_stride, int dtype,
                             sputnik_hip_stream_t stream);
SPUTNIK_HIP_API int sputnik_hip_sparse_softmax_backward_typed(int m, int nonzeros, int replicas,
                             const void* softmax_out, int64_t out_stride, const void* grad_out,
                             int64_t grad_out_stride, const int* row_offsets, float scale,
                             void* grad_values, int64_t grad_values_stride, int dtype,
                             sputnik_hip_stream_t stream);

/*
 * Fused sparse attention forward, one launch for
 *   out = spmm(softmax(scale * sddmm(q, k)), v)
 * i.e. the chain of modules/sparse_attention.py:66-82 (sddmm :68-71, the
 * division by sqrt(d) :72, sparse_softmax :76, spmm :79-82) without the
 * [replicas, nonzeros] intermediates.  q [m,d], k and v [n,d] per replica,
 * out [m,d]; `lse` (may be NULL) receives log(sum(exp(scale*score))) per row
 * for a later backward.  Served shapes: d = 64 with 16-byte aligned operands
 * (sputnik_hip_sparse_attention_supported); anything else returns
 * SPUTNIK_HIP_UNSUPPORTED and the caller composes the three operators.
 * Rows without entries produce zeros (and lse = -inf).
 */
SPUTNIK_HIP_API int sputnik_hip_sparse_attention_supported(int m, int n, int d, int nonzeros);

SPUTNIK_HIP_API size_t sputnik_hip_sparse_attention_workspace_bytes(int m, int n, int d,
                                                                    int nonzeros);

SPUTNIK_HIP_API int sputnik_hip_sparse_attention_forward(int m, int n, int d, int nonzeros,
                             int replicas, const int* row_indices, const int* row_offsets,
                             const int* column_indices, const float* q, int64_t q_stride,
                             const float* k, int64_t k_stride, const float* v,
                             int64_t v_stride, float scale, float* out, int64_t out_stride,
                             float* lse, int64_t lse_stride, void* workspace,
                             size_t workspace_bytes, sputnik_hip_stream_t stream);

/* The same in two steps for a static mask: plan once, run any number of times. */
SPUTNIK_HIP_API int sputnik_hip_sparse_attention_plan(int m, int n, int d, int nonzeros,
                             const int* row_indices, const int* row_offsets,
                             const int* column_indices, void* workspace,
                             size_t workspace_bytes, sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sparse_attention_forward_planned(int m, int n, int d,
                             int nonzeros, int replicas, const int* row_indices,
                             const int* row_offsets, const int* column_indices, const float* q,
                             int64_t q_stride, const float* k, int64_t k_stride, const float* v,
                             int64_t v_stride, float scale, float* out, int64_t out_stride,
                             float* lse, int64_t lse_stride, const void* workspace,
                             size_t workspace_bytes, sputnik_hip_stream_t stream);

/* ------------------------------------------------------------------------
 * "many mask" family: `masks` topologies of the same m x n shape, laid out
 * as tests/transformer/utils.py:17-38 builds them:
 *   row_indices    [masks * m]        local row ids of mask i at i*m
 *   row_offsets    [masks * (m + 1)]  each mask's offsets start at 0
 *   column_indices [sum nonzeros[i]]  concatenated
 *   nonzeros       [masks]            HOST array
 * `replicas` (a multiple of `masks`) value / dense slices; replica r uses
 * mask r / (replicas / masks), i.e. the heads of one batch element share its
 * mask (tests/test_attention_many_masks.py:107-150).  Replica r keeps its
 * nonzeros[mask] values at values + r * values_stride; the stride is at least
 * max(nonzeros), and anything past a replica's own count is never written (it
 * may be read: the rows are moved in aligned 16-byte pieces).
 * Call sites: torch_sputnik.{sddmm,sparse_softmax,spmm,csr_transpose}_many_mask,
 * tests/transformer/functions.py:20,41,50,59,81,135,156,165,177.
 * One launch per operator serves all masks (softmax pair; SDDMM; SpMM where the
 * panel-resident kernel applies, n = head_dim), see csrc/many_mask.hip.
 * Workspaces: the single-mask query at max(nonzeros); for SDDMM
 * sputnik_hip_sddmm_many_mask_workspace_bytes (one plan per mask; with less the
 * call still works, on the workspace-free kernel).
 * ---------------------------------------------------------------------- */
SPUTNIK_HIP_API size_t sputnik_hip_sddmm_many_mask_workspace_bytes(int masks, int m, int k, int n,
                                                                   int largest_nonzeros);

SPUTNIK_HIP_API int sputnik_hip_spmm_many_mask(int masks, int m, int k, int n,
                             const int* nonzeros, int replicas, const int* row_indices,
                             const float* values, int64_t values_stride,
                             const int* row_offsets, const int* column_indices,
                             const float* dense, int64_t dense_stride, float* out,
                             int64_t out_stride, void* workspace,
                             size_t workspace_bytes, sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sddmm_many_mask(int masks, int m, int k, int n,
                             const int* nonzeros, int replicas, const int* row_indices,
                             const int* row_offsets, const int* column_indices,
                             const float* lhs, int64_t lhs_stride, const float* rhs,
                             int64_t rhs_stride, float* out, int64_t out_stride,
                             void* workspace, size_t workspace_bytes,
                             sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sparse_softmax_many_mask(int masks, int m,
                             const int* nonzeros, int replicas, const float* values,
                             int64_t values_stride, const int* row_indices,
                             const int* row_offsets, const int* column_indices,
                             float scale, float* out, int64_t out_stride,
                             sputnik_hip_stream_t stream);

SPUTNIK_HIP_API int sputnik_hip_sparse_softmax_backward_many_mask(int masks, int m,
                             const int* nonzeros, int replicas, const float* softmax_out,
                             int64_t out_stride, const float* grad_out,
                             int64_t grad_out_stride, const int* row_offsets, float scale,
                             float* grad_values, int64_t grad_values_stride,
                             sputnik_hip_stream_t stream);

/* out_row_offsets [masks][n + 1], out_column_indices / out_permutation
 * (may be NULL) laid out like column_indices.  With a workspace of
 * sputnik_hip_csr_transpose_many_mask_workspace_bytes (a region of tables and
 * room for the permutation per mask) all masks are transposed by the same three
 * launches, plus one gather for the values of all replicas when a mask has
 * several heads; with less, but at least
 * sputnik_hip_csr_transpose_workspace_bytes(m, n, largest), mask after mask. */
SPUTNIK_HIP_API size_t sputnik_hip_csr_transpose_many_mask_workspace_bytes(int masks, int m, int n,
                             int largest_nonzeros);
SPUTNIK_HIP_API int sputnik_hip_csr_transpose_many_mask(int masks, int m, int n,
                             const int* nonzeros, int replicas, const float* values,
                             int64_t values_stride, const int* row_offsets,
                             const int* column_indices, float* out_values,
                             int64_t out_values_stride, int* out_row_offsets,
                             int* out_column_indices, int* out_permutation,
                             void* workspace, size_t workspace_bytes,
                             sputnik_hip_stream_t stream);

/*
 * out[r][i] = in[r][permutation[i]], i < n, for `rows` value arrays (strides in
 * elements) that share one permutation: the values of a static pattern in the
 * order of its transpose, given the permutation sputnik_hip_csr_transpose
 * returned once (out_permutation).  Replaces the per-backward csr_transpose of
 * modules/spmm.py:59-62, modules/sddmm.py:60-63 and
 * modules/sparse_linear.py:52-55 for topologies that do not change.
 */
SPUTNIK_HIP_API int sputnik_hip_permute_last_batched(int n, int rows, const float* in, int64_t in_stride,
                                     const int* permutation, float* out,
                                     int64_t out_stride, sputnik_hip_stream_t stream);

/*
 * The same permutation for MANY rows (attention weights: one row per batch x
 * head), through LDS.  The caller regroups the permutation once per topology by
 * the band of B = sputnik_hip_permute_band_size() consecutive SOURCE entries a
 * value comes from: `dest_list` holds the output positions i, band after band
 * (band b = list positions [b*B, min((b+1)*B, n)): exactly the positions whose
 * source permutation[i] lies in [b*B, (b+1)*B)), ascending inside a band, and
 * `source_in_band[t] = permutation[dest_list[t]] - (t / B) * B`.  (A stable sort
 * of the positions by permutation[i] / B, or one counting pass on the host.)
 *   out[r][dest_list[t]] = in[r][(t / B) * B + source_in_band[t]]
 */
SPUTNIK_HIP_API int sputnik_hip_permute_band_size(void);

SPUTNIK_HIP_API int sputnik_hip_permute_banded_batched(int n, int rows, const float* in,
                                     int64_t in_stride, const int* dest_list,
                                     const int* source_in_band, float* out,
                                     int64_t out_stride, sputnik_hip_stream_t stream);

/*
 * The library's developer / test knobs (SPUTNIK_HIP_* environment variables:
 * kernel choice for small inputs, timing experiments) are read once, at first
 * use, never on the launch path; this re-reads them.  Not to be called while
 * other threads issue launches.  (tests/conftest.py steers the parity tests
 * onto every kernel with it.)
 */
SPUTNIK_HIP_API void sputnik_hip_reload_options(void);

/*
 * Batched 2-D transpose  out[b][c][r] = in[b][r][c]  (row-major, fp32; batch
 * strides in elements).  The layout pass of the reference's modules:
 * `x.transpose(1, 2).contiguous()` in front of left_spmm
 * (modules/sparse_linear.py:89), the same behind every projection and the head
 * split / merge copies of modules/sparse_attention.py:108-126 are all this
 * operation (the head split [B, H*D, S] -> [B*H, S, D] is a transpose of B*H
 * matrices of D x S).  No workspace, no synchronisation.
 */
SPUTNIK_HIP_API int sputnik_hip_transpose_batched(int batches, int rows, int cols, const float* in,
                                  int64_t in_batch_stride, float* out,
                                  int64_t out_batch_stride,
                                  sputnik_hip_stream_t stream);

/*
 * The same with a change of storage type on the way (in_type / out_type:
 * SPUTNIK_HIP_F32 / F16 / BF16; supported: equal types, half -> float,
 * float -> half).  BASELINE.json's config 5 stores activations in fp16 while the
 * operators compute and return fp32 (src/spmm_cuda.cu:42): the widening happens
 * inside the layout pass in front of left_spmm, the narrowing of the gradient
 * inside the pass behind it -- no pass of its own.
 */
SPUTNIK_HIP_API int sputnik_hip_transpose_cast_batched(int batches, int rows, int cols, const void* in,
                                       int in_type, int64_t in_batch_stride, void* out,
                                       int out_type, int64_t out_batch_stride,
                                       sputnik_hip_stream_t stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* SPUTNIK_HIP_H_ */
