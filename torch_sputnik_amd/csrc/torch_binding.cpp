// Host side of the drop-in: the reference's five operators
// (src/sputnik.cpp:36-42) registered with TORCH_LIBRARY as
// torch.ops.torch_sputnik.{spmm,left_spmm,sddmm,sparse_softmax,csr_transpose}
// for HIP tensors only, on top of the C ABI in include/sputnik_hip.h.
//
// Each op mirrors one host wrapper of the reference:
//   spmm            src/spmm_cuda.cu:9-60
//   left_spmm       src/left_replicated_spmm.cu:8-44
//   sddmm           src/sddmm_cuda.cu:7-57
//   sparse_softmax  src/softmax_cuda.cu:7-46
//   csr_transpose   src/transpose_cuda.cu:45-102
// Same argument order and meaning; differences, all on purpose:
//   - the reference's shape `assert`s (no-ops under NDEBUG) are TORCH_CHECKs,
//     so a bad call raises RuntimeError instead of being undefined;
//   - the replica loop is one launch, outputs are at::empty (every element is
//     written by the kernels) instead of torch::zeros;
//   - integer tensors may be int64 (the reference's own SparseLinear backward
//     hands an int64 row_indices to left_spmm, modules/sparse_linear.py:57-65)
//     and are cast to int32 here; non-contiguous inputs are made contiguous;
//   - a device guard on the inputs' device (the reference has none).
// There is no CPU kernel: CPU tensors raise from the dispatcher.  Autograd is
// not registered here; the Python autograd.Functions stay the callers.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <algorithm>
#include <vector>

#include "../../include/sputnik_hip.h"

namespace {

using at::Tensor;

void check_status(int status, const char* what) {
  TORCH_CHECK(status == 0, "torch_sputnik::", what, ": kernel library returned error ", status,
              status == SPUTNIK_HIP_INVALID_ARGUMENT ? " (invalid argument)" : " (hipError_t)");
}

sputnik_hip_stream_t current_stream(const Tensor& t) {
  return reinterpret_cast<sputnik_hip_stream_t>(
      c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

Tensor as_index(const Tensor& t, const char* name, const Tensor& like) {
  TORCH_CHECK(t.scalar_type() == at::kInt || t.scalar_type() == at::kLong, name,
              " must be an int32 (or int64) tensor, got ", t.scalar_type());
  TORCH_CHECK(t.device() == like.device(), name, " must be on ", like.device(), ", got ",
              t.device());
  TORCH_CHECK(t.dim() == 1, name, " should have 1 dimension, got ", t.dim());
  return t.to(at::kInt).contiguous();
}

// float32 is the reference's only dtype (data_ptr<float>(), src/spmm_cuda.cu:51);
// float16 / bfloat16 STORAGE is accepted as an extension (BASELINE.json config 5
// names fp16), all arithmetic accumulates in float32.  THE RESULT-TYPE RULE, per op
// (one place that says it):
//   spmm, left_spmm, spmm_bias*, the planned / permuted / transposed-store / group
//   forms, sddmm, sddmm_sum, csr_transpose (values_t), sparse_attention*, every
//   many-mask op                         -> float32 results whatever the operand type
//                                           (the reference's rule, src/spmm_cuda.cu:42);
//   sparse_softmax, sparse_softmax_scaled, sparse_softmax_backward
//                                        -> the type of `values` (an elementwise map of
//                                           the stored values: half in, half out; float32
//                                           in, float32 out exactly as the reference);
//   sddmm_narrow                         -> the operands' (half) type, on request only;
//   transpose_last2_as                   -> the type its out_type argument names.
// Ops that take float32 only in their kernels widen half operands once with as_float
// below; the ops whose kernels read half storage natively take as_storage.
Tensor as_float(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU (HIP) tensor, got ", t.device());
  const auto st = t.scalar_type();
  TORCH_CHECK(st == at::kFloat || st == at::kHalf || st == at::kBFloat16, name,
              " must be float32 (or float16 / bfloat16 storage), got ", st);
  return t.to(at::kFloat).contiguous();
}

// Operands of the natively typed kernels (softmax pair, SDDMM): float32, float16 or
// bfloat16 storage is handed to the kernel as it is -- no widening copy.
Tensor as_storage(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU (HIP) tensor, got ", t.device());
  const auto st = t.scalar_type();
  TORCH_CHECK(st == at::kFloat || st == at::kHalf || st == at::kBFloat16, name,
              " must be float32, float16 or bfloat16, got ", st);
  return t.contiguous();
}
int type_code(at::ScalarType t) {   // -1: not a storage type of the library
  return t == at::kFloat ? SPUTNIK_HIP_F32 : t == at::kHalf ? SPUTNIK_HIP_F16
         : t == at::kBFloat16 ? SPUTNIK_HIP_BF16 : -1;
}

struct Topology {
  Tensor row_indices, row_offsets, column_indices;
  int nonzeros;
};

Topology check_topology(int64_t m, const Tensor& row_indices, const Tensor& row_offsets,
                        const Tensor& column_indices, const Tensor& like) {
  Topology t;
  t.row_indices = as_index(row_indices, "row_indices", like);
  t.row_offsets = as_index(row_offsets, "row_offsets", like);
  t.column_indices = as_index(column_indices, "column_indices", like);
  TORCH_CHECK(t.row_indices.size(0) + 1 == t.row_offsets.size(0),
              "row_offsets should have one more entry than row_indices, got ",
              t.row_offsets.size(0), " and ", t.row_indices.size(0));
  TORCH_CHECK(t.row_indices.size(0) == m, "number of row_indices (", t.row_indices.size(0),
              ") and m (", m, ") must match");
  TORCH_CHECK(t.column_indices.size(0) < (int64_t{1} << 31), "too many nonzeros");
  t.nonzeros = static_cast<int>(t.column_indices.size(0));
  return t;
}

int to_int(int64_t v, const char* name) {
  TORCH_CHECK(v >= 0 && v < (int64_t{1} << 31), name, " out of range: ", v);
  return static_cast<int>(v);
}

// A plan is the workspace of the matching *_workspace_bytes query after the
// topology-only pre-pass ran in it (spmm_plan / sddmm_plan / sparse_attention_plan).
void check_plan(const Tensor& plan, size_t bytes, const Tensor& like) {
  TORCH_CHECK(plan.scalar_type() == at::kByte && plan.is_contiguous() && plan.dim() == 1,
              "plan must be the uint8 tensor a *_plan op returned");
  TORCH_CHECK(plan.device() == like.device(), "plan must be on ", like.device());
  TORCH_CHECK(static_cast<size_t>(plan.numel()) >= bytes,
              "plan is too small for this call (", plan.numel(), " < ", bytes,
              " bytes): it was made for other sizes");
}

Tensor permute_last(const Tensor& values_in, const Tensor& permutation);
Tensor transpose_last2(const Tensor& x);

// Shared by spmm (values [nnz] / [R,nnz]) and left_spmm (values [nnz], shared).
Tensor spmm_impl(int64_t m64, int64_t k64, const Tensor& values_in, const Tensor& row_indices,
                 const Tensor& row_offsets, const Tensor& column_indices, const Tensor& dense_in,
                 bool left, const char* what, const c10::optional<Tensor>& bias_in = c10::nullopt,
                 bool relu = false, const c10::optional<Tensor>& plan = c10::nullopt,
                 const c10::optional<Tensor>& permutation = c10::nullopt, int64_t block_rows = 0) {
  const int m = to_int(m64, "m"), k = to_int(k64, "k");
  // float16 / bfloat16 operands of the plain product go to the kernels as they are
  // (sputnik_hip_spmm_typed); the permuted / transposed-store forms, which the modules
  // reach with float32 operands, and a float16 / bfloat16 mix take the widened path.
  const auto is_half = [](const Tensor& t) {
    return t.scalar_type() == at::kHalf || t.scalar_type() == at::kBFloat16;
  };
  const bool native_half =
      (is_half(values_in) || is_half(dense_in)) && !permutation.has_value() && block_rows == 0 &&
      !(is_half(values_in) && is_half(dense_in) && values_in.scalar_type() != dense_in.scalar_type());
  Tensor values = native_half ? as_storage(values_in, "values") : as_float(values_in, "values");
  const Tensor dense = native_half ? as_storage(dense_in, "dense") : as_float(dense_in, "dense");
  TORCH_CHECK(dense.device() == values.device(), "values and dense must be on one device");
  TORCH_CHECK(dense.dim() == 2 || dense.dim() == 3, "dense should have 2 or 3 dimensions, got ",
              dense.dim());
  if (left) {
    TORCH_CHECK(values.dim() == 1, "left_spmm: values should have 1 dimension, got ",
                values.dim());
  } else {
    TORCH_CHECK(values.dim() == 1 || values.dim() == 2,
                "values should have 1 or 2 dimensions, got ", values.dim());
    TORCH_CHECK(values.dim() == dense.dim() - 1,
                "values and dense must be replicated the same: values.dim()=", values.dim(),
                ", dense.dim()=", dense.dim());
  }
  const c10::DeviceGuard guard(values.device());
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, values);

  const int dim_offset = static_cast<int>(dense.dim()) - 2;
  const int replicas = dim_offset == 1 ? to_int(dense.size(0), "replicas") : 1;
  const int n = to_int(dense.size(dim_offset + 1), "n");
  TORCH_CHECK(values.size(-1) == topo.nonzeros, "number of values (", values.size(-1),
              ") must equal the number of column_indices (", topo.nonzeros, ")");
  TORCH_CHECK(dense.size(dim_offset) == k, "inner matrix dimensions must match: dense has ",
              dense.size(dim_offset), " rows, k = ", k);
  if (!left && values.dim() == 2) {
    TORCH_CHECK(values.size(0) == replicas, "first dim of values (", values.size(0),
                ") and dense (", replicas, ") must match");
  }

  const auto options = values.options().dtype(at::kFloat);   // the product is float32
  const int64_t values_stride = (left || values.dim() == 1) ? 0 : topo.nonzeros;
  if (block_rows > 0) {
    // The product stored as the transposes of its blocks of `block_rows` rows,
    // [replicas * m / block_rows, n, block_rows]: the head split behind a projection
    // (modules/sparse_attention.py:38-45) or the whole C^T (block_rows = m), written
    // by the panel kernel's store phase where it serves the shape; elsewhere the
    // usual product followed by the tiled transpose kernel.
    TORCH_CHECK(m % block_rows == 0, "block_rows (", block_rows, ") must divide m = ", m);
    if (permutation.has_value())
      TORCH_CHECK(permutation->scalar_type() == at::kInt && permutation->dim() == 1 &&
                      permutation->is_contiguous() && permutation->device() == values.device() &&
                      permutation->size(0) == topo.nonzeros,
                  "permutation must be a contiguous int32 vector of ", topo.nonzeros,
                  " entries on ", values.device());
    const bool fused_perm = !permutation.has_value() ||
                            sputnik_hip_spmm_permuted_supported(m, k, n, topo.nonzeros);
    if (fused_perm && sputnik_hip_spmm_transposed_out_supported(m, k, n, topo.nonzeros,
                                                                static_cast<int>(block_rows))) {
      Tensor bias;
      if (bias_in.has_value()) {
        bias = as_float(*bias_in, "bias");
        TORCH_CHECK(bias.device() == values.device() && bias.dim() == 1 && bias.size(0) == m,
                    "bias should have m = ", m, " elements on ", values.device());
      }
      Tensor tout = at::empty({replicas * (m / block_rows), n, block_rows}, options);
      const int st = sputnik_hip_spmm_transposed_out_batched(
          m, k, n, topo.nonzeros, replicas, values.data_ptr<float>(), values_stride,
          permutation.has_value() ? permutation->data_ptr<int>() : nullptr,
          topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(),
          dense.data_ptr<float>(), static_cast<int64_t>(k) * n,
          bias.defined() ? bias.data_ptr<float>() : nullptr, relu ? 1 : 0,
          static_cast<int>(block_rows), tout.data_ptr<float>(), static_cast<int64_t>(m) * n,
          current_stream(values));
      if (st != SPUTNIK_HIP_UNSUPPORTED) {
        check_status(st, what);
        return tout;
      }
    }
    const Tensor c = spmm_impl(m64, k64, values_in, row_indices, row_offsets, column_indices,
                               dense_in, left, what, bias_in, relu, plan, permutation, 0);
    return transpose_last2(c.reshape({replicas * (m / block_rows), block_rows, n}));
  }
  Tensor out = (replicas == 1 && !left) ? at::empty({m, n}, options)
                                        : at::empty({replicas, m, n}, options);
  if (permutation.has_value()) {
    // values are those of ANOTHER ordering of the same entries (the topology here
    // is its transpose): entry p takes values[permutation[p]].  One kernel where
    // the panel kernel serves the shape in one pass; otherwise the values are
    // permuted first and the call goes on as usual.
    TORCH_CHECK(permutation->scalar_type() == at::kInt && permutation->dim() == 1 &&
                    permutation->is_contiguous() && permutation->device() == values.device() &&
                    permutation->size(0) == topo.nonzeros,
                "permutation must be a contiguous int32 vector of ", topo.nonzeros,
                " entries on ", values.device());
    int st = SPUTNIK_HIP_UNSUPPORTED;
    if (sputnik_hip_spmm_permuted_supported(m, k, n, topo.nonzeros))
      st = sputnik_hip_spmm_permuted_batched(
          m, k, n, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
          values.data_ptr<float>(), values_stride, permutation->data_ptr<int>(),
          topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(),
          dense.data_ptr<float>(), static_cast<int64_t>(k) * n, out.data_ptr<float>(),
          static_cast<int64_t>(m) * n, current_stream(values));
    if (st != SPUTNIK_HIP_UNSUPPORTED) {
      check_status(st, what);
      return out;
    }
    values = permute_last(values, *permutation);
  }
  if (native_half) {   // (a plan has nothing to add: the half-reading kernels have no pre-pass)
    Tensor bias;
    if (bias_in.has_value()) {
      bias = as_float(*bias_in, "bias");
      TORCH_CHECK(bias.device() == values.device() && bias.dim() == 1 && bias.size(0) == m,
                  "bias should have m = ", m, " elements on ", values.device());
    }
    const size_t typed_ws = sputnik_hip_spmm_typed_workspace_bytes(
        m, k, n, topo.nonzeros, replicas, type_code(values.scalar_type()), values_stride,
        type_code(dense.scalar_type()));
    Tensor workspace;
    char* ws = nullptr;
    if (typed_ws > 0) {
      workspace = at::empty({static_cast<int64_t>(typed_ws + 256)}, options.dtype(at::kByte));
      ws = static_cast<char*>(workspace.data_ptr());
      ws += (256 - reinterpret_cast<uintptr_t>(ws) % 256) % 256;
    }
    check_status(sputnik_hip_spmm_typed(
                     m, k, n, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
                     values.data_ptr(), type_code(values.scalar_type()), values_stride,
                     topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(),
                     dense.data_ptr(), type_code(dense.scalar_type()), static_cast<int64_t>(k) * n,
                     bias.defined() ? bias.data_ptr<float>() : nullptr, relu ? 1 : 0,
                     out.data_ptr<float>(), static_cast<int64_t>(m) * n, ws, typed_ws,
                     current_stream(values)),
                 what);
    return out;
  }
  const size_t ws_bytes = sputnik_hip_spmm_workspace_bytes(m, k, n, topo.nonzeros);
  if (plan.has_value()) {
    // static topology: the pre-pass ran once (spmm_plan); kernels only
    check_plan(*plan, ws_bytes, values);
    check_status(sputnik_hip_spmm_batched_planned(
                     m, k, n, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
                     values.data_ptr<float>(), values_stride, topo.row_offsets.data_ptr<int>(),
                     topo.column_indices.data_ptr<int>(), dense.data_ptr<float>(),
                     static_cast<int64_t>(k) * n, out.data_ptr<float>(),
                     static_cast<int64_t>(m) * n, ws_bytes ? plan->data_ptr() : nullptr, ws_bytes,
                     current_stream(values)),
                 what);
    return out;
  }
  Tensor workspace;
  if (ws_bytes > 0)
    workspace = at::empty({static_cast<int64_t>(ws_bytes)}, options.dtype(at::kByte));

  Tensor bias;
  if (bias_in.has_value()) {
    bias = as_float(*bias_in, "bias");
    TORCH_CHECK(bias.device() == values.device(), "bias must be on ", values.device());
    TORCH_CHECK(bias.dim() == 1 && bias.size(0) == m, "bias should have m = ", m,
                " elements (one per output row), got ", bias.sizes());
  }
  check_status(sputnik_hip_spmm_bias_batched(
                   m, k, n, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
                   values.data_ptr<float>(), values_stride, topo.row_offsets.data_ptr<int>(),
                   topo.column_indices.data_ptr<int>(), dense.data_ptr<float>(),
                   static_cast<int64_t>(k) * n, bias.defined() ? bias.data_ptr<float>() : nullptr,
                   relu ? 1 : 0, out.data_ptr<float>(), static_cast<int64_t>(m) * n,
                   ws_bytes ? workspace.data_ptr() : nullptr, ws_bytes, current_stream(values)),
               what);
  return out;
}

Tensor spmm(int64_t m, int64_t k, const Tensor& values, const Tensor& row_indices,
            const Tensor& row_offsets, const Tensor& column_indices, const Tensor& dense) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, false, "spmm");
}

// spmm / left_spmm over a topology whose values are given in another order
// (values_here[p] = values[permutation[p]]): the transposed products of the
// backward passes, modules/spmm.py:59-66, without the permuted copy.
Tensor spmm_permuted(int64_t m, int64_t k, const Tensor& values, const Tensor& permutation,
                     const Tensor& row_indices, const Tensor& row_offsets,
                     const Tensor& column_indices, const Tensor& dense,
                     const c10::optional<Tensor>& plan) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, false,
                   "spmm_permuted", c10::nullopt, false, plan, permutation);
}

// spmm / left_spmm with the product stored as the transposes of its blocks of
// `block_rows` rows -> [replicas * m / block_rows, n, block_rows]; `permutation`
// as in spmm_permuted.
Tensor spmm_transposed_out(int64_t m, int64_t k, const Tensor& values,
                           const c10::optional<Tensor>& permutation, const Tensor& row_indices,
                           const Tensor& row_offsets, const Tensor& column_indices,
                           const Tensor& dense, int64_t block_rows, bool left,
                           const c10::optional<Tensor>& plan) {
  TORCH_CHECK(block_rows > 0, "block_rows must be positive, got ", block_rows);
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, left,
                   "spmm_transposed_out", c10::nullopt, false, plan, permutation, block_rows);
}

// ---------------------------------------------------------------------------
// Groups of projections in one launch (sputnik_hip.h: sputnik_hip_spmm_group_batched).
// ---------------------------------------------------------------------------
struct GroupArgs {
  std::vector<Tensor> values, permutations, dense;
  std::vector<Topology> topo;
  int m, k, n, replicas;
};

GroupArgs check_group(int64_t m64, int64_t k64, at::TensorList values, at::TensorList permutations,
                      at::TensorList row_indices, at::TensorList row_offsets,
                      at::TensorList column_indices, at::TensorList dense, const char* what) {
  const size_t count = values.size();
  TORCH_CHECK(count >= 1, what, ": expected at least one matrix");
  TORCH_CHECK(row_indices.size() == count && row_offsets.size() == count &&
                  column_indices.size() == count && (dense.size() == count || dense.size() == 1) &&
                  (permutations.empty() || permutations.size() == count),
              what, ": the lists must have one entry per matrix");
  GroupArgs a;
  a.m = to_int(m64, "m");
  a.k = to_int(k64, "k");
  for (size_t p = 0; p < dense.size(); ++p) {
    Tensor d = as_float(dense[p], "dense");
    TORCH_CHECK(d.dim() == 3 && d.size(1) == a.k, what, ": dense should be [B, k = ", a.k,
                ", n], got ", d.sizes());
    if (p == 0) {
      a.replicas = to_int(d.size(0), "replicas");
      a.n = to_int(d.size(2), "n");
    }
    TORCH_CHECK(d.size(0) == a.replicas && d.size(2) == a.n && d.device() == dense[0].device(),
                what, ": all dense operands must have one shape and device");
    a.dense.push_back(d);
  }
  for (size_t p = 0; p < count; ++p) {
    Tensor v = as_float(values[p], "values");
    TORCH_CHECK(v.dim() == 1 && v.device() == a.dense[0].device(), what,
                ": values should have 1 dimension (shared by the batch) on ", a.dense[0].device());
    Topology t = check_topology(a.m, row_indices[p], row_offsets[p], column_indices[p], v);
    TORCH_CHECK(v.size(0) == t.nonzeros, "number of values (", v.size(0),
                ") must equal the number of column_indices (", t.nonzeros, ")");
    if (!permutations.empty()) {
      const Tensor& perm = permutations[p];
      TORCH_CHECK(perm.scalar_type() == at::kInt && perm.dim() == 1 && perm.is_contiguous() &&
                      perm.device() == v.device() && perm.size(0) == t.nonzeros,
                  "permutation must be a contiguous int32 vector of ", t.nonzeros, " entries");
      a.permutations.push_back(perm);
    }
    a.values.push_back(v);
    a.topo.push_back(t);
  }
  return a;
}

// One input, several sparse weights, every product stored head split
// ([B * m / block_rows, n, block_rows] each; block_rows = 0: plain [B, m, n]).
std::vector<Tensor> left_spmm_group(int64_t m64, int64_t k64, at::TensorList values,
                                    at::TensorList row_indices, at::TensorList row_offsets,
                                    at::TensorList column_indices, const Tensor& dense,
                                    int64_t block_rows) {
  const GroupArgs a = check_group(m64, k64, values, {}, row_indices, row_offsets, column_indices,
                                  {dense}, "left_spmm_group");
  const c10::DeviceGuard guard(a.dense[0].device());
  const size_t count = a.values.size();
  std::vector<Tensor> outs;
  if (sputnik_hip_spmm_group_supported(a.m, a.k, a.n, static_cast<int>(count),
                                       static_cast<int>(block_rows), 0)) {
    std::vector<sputnik_hip_spmm_problem> problems(count);
    bool all_nonempty = true;
    for (size_t p = 0; p < count; ++p) {
      outs.push_back(block_rows > 0
                         ? at::empty({a.replicas * (a.m / block_rows), a.n, block_rows},
                                     a.values[p].options())
                         : at::empty({a.replicas, a.m, a.n}, a.values[p].options()));
      all_nonempty = all_nonempty && a.topo[p].nonzeros > 0;
      problems[p] = {a.topo[p].row_indices.data_ptr<int>(), a.topo[p].row_offsets.data_ptr<int>(),
                     a.topo[p].column_indices.data_ptr<int>(), a.values[p].data_ptr<float>(),
                     nullptr, a.dense[0].data_ptr<float>(), outs[p].data_ptr<float>(),
                     a.topo[p].nonzeros};
    }
    const int st = !all_nonempty ? SPUTNIK_HIP_UNSUPPORTED
                                 : sputnik_hip_spmm_group_batched(
                                       a.m, a.k, a.n, a.replicas, static_cast<int>(count),
                                       problems.data(), static_cast<int64_t>(a.k) * a.n,
                                       static_cast<int64_t>(a.m) * a.n,
                                       static_cast<int>(block_rows), 0,
                                       current_stream(a.dense[0]));
    if (st != SPUTNIK_HIP_UNSUPPORTED) {
      check_status(st, "left_spmm_group");
      return outs;
    }
    outs.clear();
  }
  for (size_t p = 0; p < count; ++p)   // one by one
    outs.push_back(spmm_impl(m64, k64, values[p], row_indices[p], row_offsets[p],
                             column_indices[p], dense, true, "left_spmm_group", c10::nullopt,
                             false, c10::nullopt, c10::nullopt, block_rows));
  return outs;
}

// sum over the matrices of  A_p @ dense_p  (values gathered through
// `permutations` when given): the input gradient of a group of projections.
Tensor left_spmm_group_sum(int64_t m64, int64_t k64, at::TensorList values,
                           at::TensorList permutations, at::TensorList row_indices,
                           at::TensorList row_offsets, at::TensorList column_indices,
                           at::TensorList dense) {
  const GroupArgs a = check_group(m64, k64, values, permutations, row_indices, row_offsets,
                                  column_indices, dense, "left_spmm_group_sum");
  TORCH_CHECK(a.dense.size() == a.values.size(),
              "left_spmm_group_sum: one dense operand per matrix");
  const c10::DeviceGuard guard(a.dense[0].device());
  const size_t count = a.values.size();
  if (sputnik_hip_spmm_group_supported(a.m, a.k, a.n, static_cast<int>(count), 0, 1)) {
    Tensor out = at::empty({a.replicas, a.m, a.n}, a.values[0].options());
    std::vector<sputnik_hip_spmm_problem> problems(count);
    bool all_nonempty = true;
    for (size_t p = 0; p < count; ++p) {
      all_nonempty = all_nonempty && a.topo[p].nonzeros > 0;
      problems[p] = {a.topo[p].row_indices.data_ptr<int>(), a.topo[p].row_offsets.data_ptr<int>(),
                     a.topo[p].column_indices.data_ptr<int>(), a.values[p].data_ptr<float>(),
                     a.permutations.empty() ? nullptr : a.permutations[p].data_ptr<int>(),
                     a.dense[p].data_ptr<float>(), out.data_ptr<float>(), a.topo[p].nonzeros};
    }
    const int st = !all_nonempty ? SPUTNIK_HIP_UNSUPPORTED
                                 : sputnik_hip_spmm_group_batched(
                                       a.m, a.k, a.n, a.replicas, static_cast<int>(count),
                                       problems.data(), static_cast<int64_t>(a.k) * a.n,
                                       static_cast<int64_t>(a.m) * a.n, 0, 1,
                                       current_stream(a.dense[0]));
    if (st != SPUTNIK_HIP_UNSUPPORTED) {
      check_status(st, "left_spmm_group_sum");
      return out;
    }
  }
  Tensor total;
  for (size_t p = 0; p < count; ++p) {   // one by one
    const Tensor part = spmm_impl(
        m64, k64, values[p], row_indices[p], row_offsets[p], column_indices[p], dense[p], true,
        "left_spmm_group_sum", c10::nullopt, false, c10::nullopt,
        permutations.empty() ? c10::optional<Tensor>() : c10::optional<Tensor>(permutations[p]));
    total = p == 0 ? part : total + part;
  }
  return total;
}

Tensor left_spmm_permuted(int64_t m, int64_t k, const Tensor& values, const Tensor& permutation,
                          const Tensor& row_indices, const Tensor& row_offsets,
                          const Tensor& column_indices, const Tensor& dense,
                          const c10::optional<Tensor>& plan) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, true,
                   "left_spmm_permuted", c10::nullopt, false, plan, permutation);
}

Tensor left_spmm(int64_t m, int64_t k, const Tensor& values, const Tensor& row_indices,
                 const Tensor& row_offsets, const Tensor& column_indices, const Tensor& dense) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, true,
                   "left_spmm");
}

// left_spmm as a dense contraction on half tiles (sputnik_hip.h: left_spmm_half_tiles):
// values [nnz] and dense [R, k, n] float32 or of the tile type (1 float16 / 2 bfloat16).
// An EMPTY tensor says the route does not serve the call: take left_spmm then.
Tensor left_spmm_half_tiles(int64_t m64, int64_t k64, const Tensor& values_in,
                            const Tensor& row_offsets, const Tensor& column_indices,
                            const Tensor& dense_in, int64_t tile_type) {
  const int m = to_int(m64, "m"), k = to_int(k64, "k");
  const Tensor values = as_storage(values_in, "values");
  const Tensor dense = as_storage(dense_in, "dense");
  TORCH_CHECK(values.dim() == 1, "left_spmm_half_tiles: values should have 1 dimension");
  TORCH_CHECK(dense.dim() == 3 && dense.size(1) == k, "left_spmm_half_tiles: dense must be [R, k, n]");
  TORCH_CHECK(dense.device() == values.device(), "values and dense must be on one device");
  const c10::DeviceGuard guard(values.device());
  const Tensor ro = as_index(row_offsets, "row_offsets", values);
  const Tensor ci = as_index(column_indices, "column_indices", values);
  TORCH_CHECK(ro.size(0) == m + 1, "row_offsets should have m + 1 entries");
  TORCH_CHECK(values.size(0) == ci.size(0), "number of values and column_indices must match");
  const int nonzeros = to_int(ci.size(0), "nonzeros");
  const int replicas = to_int(dense.size(0), "replicas"), n = to_int(dense.size(2), "n");
  const int vt = type_code(values.scalar_type()), dt = type_code(dense.scalar_type());
  const size_t ws_bytes = sputnik_hip_left_spmm_half_tiles_workspace_bytes(
      m, k, n, nonzeros, replicas, vt, dt, static_cast<int>(tile_type));
  const auto options = values.options().dtype(at::kFloat);
  if (ws_bytes == 0) return at::empty({0}, options);
  Tensor workspace = at::empty({static_cast<int64_t>(ws_bytes)}, values.options().dtype(at::kByte));
  Tensor out = at::empty({replicas, m, n}, options);
  const int st = sputnik_hip_left_spmm_half_tiles(
      m, k, n, nonzeros, replicas, ro.data_ptr<int>(), ci.data_ptr<int>(), values.data_ptr(), vt,
      dense.data_ptr(), dt, static_cast<int64_t>(k) * n, static_cast<int>(tile_type), nullptr, 0,
      out.data_ptr<float>(), static_cast<int64_t>(m) * n, workspace.data_ptr(), ws_bytes,
      current_stream(values));
  if (st == SPUTNIK_HIP_UNSUPPORTED) return at::empty({0}, options);
  check_status(st, "left_spmm_half_tiles");
  return out;
}

// ---- a sparse layer on half-stored activations: the three products on the matrix cores,
// no layout pass (sputnik_hip.h: sparse_linear_half_*) ----
int tile_code_of(const Tensor& x) {
  const int code = type_code(x.scalar_type());
  TORCH_CHECK(code == SPUTNIK_HIP_F16 || code == SPUTNIK_HIP_BF16,
              "expected float16 / bfloat16 activations, got ", x.scalar_type());
  return code;
}

// The weight's image [planes][out][in] in the tile type (uint8 tensor).
Tensor half_linear_image(int64_t out64, int64_t in64, const Tensor& values_in, const Tensor& row_offsets,
                         const Tensor& column_indices, int64_t tile_type) {
  const int out_f = to_int(out64, "out"), in_f = to_int(in64, "in");
  const Tensor values = as_storage(values_in, "values");
  const c10::DeviceGuard guard(values.device());
  const Tensor ro = as_index(row_offsets, "row_offsets", values);
  const Tensor ci = as_index(column_indices, "column_indices", values);
  TORCH_CHECK(ro.size(0) == out_f + 1 && values.dim() == 1 && values.size(0) == ci.size(0),
              "half_linear_image: CSR arrays do not match");
  const int vt = type_code(values.scalar_type());
  const size_t bytes = sputnik_hip_sparse_linear_half_image_bytes(out_f, in_f, vt, static_cast<int>(tile_type));
  Tensor image = at::empty({static_cast<int64_t>(bytes)}, values.options().dtype(at::kByte));
  check_status(sputnik_hip_sparse_linear_half_image(out_f, in_f, to_int(ci.size(0), "nonzeros"),
                                                    ro.data_ptr<int>(), ci.data_ptr<int>(),
                                                    values.data_ptr(), vt, static_cast<int>(tile_type),
                                                    image.data_ptr(), bytes, current_stream(values)),
               "half_linear_image");
  return image;
}

// float32 tensor -> its half planes (uint8 tensor, planes x numel x 2 bytes).
Tensor half_planes(const Tensor& t_in, int64_t tile_type) {
  TORCH_CHECK(t_in.is_cuda() && t_in.scalar_type() == at::kFloat, "half_planes: expected a float32 GPU tensor");
  const Tensor t = t_in.contiguous();
  TORCH_CHECK(t.numel() % 4 == 0, "half_planes: element count must be a multiple of 4");
  const c10::DeviceGuard guard(t.device());
  const size_t bytes = sputnik_hip_half_planes_bytes(t.numel(), static_cast<int>(tile_type));
  Tensor planes = at::empty({static_cast<int64_t>(bytes)}, t.options().dtype(at::kByte));
  check_status(sputnik_hip_half_planes(t.numel(), t.data_ptr<float>(), static_cast<int>(tile_type),
                                       planes.data_ptr(), current_stream(t)),
               "half_planes");
  return planes;
}

// y [batch, out, seq] float32 = W x^T per batch element, x [batch, seq, in] half.
Tensor half_linear_forward(int64_t out64, const Tensor& image, int64_t values_type, const Tensor& x_in) {
  const Tensor x = x_in.contiguous();
  TORCH_CHECK(x.is_cuda() && x.dim() == 3, "half_linear_forward: x must be a GPU tensor [batch, seq, in]");
  const int tile = tile_code_of(x);
  const c10::DeviceGuard guard(x.device());
  const int out_f = to_int(out64, "out"), batch = to_int(x.size(0), "batch"), seq = to_int(x.size(1), "seq"),
            in_f = to_int(x.size(2), "in");
  TORCH_CHECK(static_cast<size_t>(image.numel()) >=
                  sputnik_hip_sparse_linear_half_image_bytes(out_f, in_f, static_cast<int>(values_type), tile),
              "half_linear_forward: the image was made for another shape");
  Tensor y = at::empty({batch, out_f, seq}, x.options().dtype(at::kFloat));
  check_status(sputnik_hip_sparse_linear_half_forward(out_f, in_f, seq, batch, image.data_ptr(),
                                                      static_cast<int>(values_type), x.data_ptr(), tile,
                                                      nullptr, 0, y.data_ptr<float>(), current_stream(x)),
               "half_linear_forward");
  return y;
}

Tensor half_linear_plan(int64_t out64, int64_t in64, const Tensor& row_offsets, const Tensor& column_indices) {
  const int out_f = to_int(out64, "out"), in_f = to_int(in64, "in");
  TORCH_CHECK(row_offsets.is_cuda(), "half_linear_plan: expected GPU index tensors");
  const c10::DeviceGuard guard(row_offsets.device());
  const Tensor ro = as_index(row_offsets, "row_offsets", row_offsets);
  const Tensor ci = as_index(column_indices, "column_indices", row_offsets);
  const size_t bytes = sputnik_hip_sparse_linear_half_plan_bytes(out_f, in_f);
  Tensor plan = at::empty({static_cast<int64_t>(bytes)}, row_offsets.options().dtype(at::kByte));
  check_status(sputnik_hip_sparse_linear_half_plan(out_f, in_f, ro.data_ptr<int>(), ci.data_ptr<int>(),
                                                   plan.data_ptr(), current_stream(row_offsets)),
               "half_linear_plan");
  return plan;
}

// grad: the planes of the float32 dy (half_planes; grad_is_planes) or dy in x's type.
Tensor half_linear_weight_gradient(int64_t out64, const Tensor& row_offsets, const Tensor& column_indices,
                                   const Tensor& grad, bool grad_is_planes, const Tensor& x_in,
                                   const c10::optional<Tensor>& plan) {
  const Tensor x = x_in.contiguous();
  const int tile = tile_code_of(x);
  const c10::DeviceGuard guard(x.device());
  const int out_f = to_int(out64, "out"), batch = to_int(x.size(0), "batch"), seq = to_int(x.size(1), "seq"),
            in_f = to_int(x.size(2), "in");
  const Tensor ro = as_index(row_offsets, "row_offsets", x);
  const Tensor ci = as_index(column_indices, "column_indices", x);
  const int nonzeros = to_int(ci.size(0), "nonzeros");
  const int grad_type = grad_is_planes ? SPUTNIK_HIP_F32 : tile;
  const Tensor g = grad.contiguous();
  const size_t want = grad_is_planes
                          ? sputnik_hip_half_planes_bytes(static_cast<int64_t>(batch) * out_f * seq, tile)
                          : static_cast<size_t>(batch) * out_f * seq * 2;
  TORCH_CHECK(static_cast<size_t>(g.numel()) * g.element_size() >= want,
              "half_linear_weight_gradient: grad does not hold [batch, out, seq]");
  const size_t scratch_bytes = sputnik_hip_sparse_linear_half_scratch_bytes(out_f, in_f, seq, batch,
                                                                            nonzeros, grad_type, tile);
  Tensor scratch;
  if (scratch_bytes) scratch = at::empty({static_cast<int64_t>(scratch_bytes)}, x.options().dtype(at::kByte));
  Tensor out = at::empty({nonzeros}, x.options().dtype(at::kFloat));
  check_status(sputnik_hip_sparse_linear_half_weight_gradient(
                   out_f, in_f, seq, batch, nonzeros, ro.data_ptr<int>(), ci.data_ptr<int>(), g.data_ptr(),
                   grad_type, x.data_ptr(), tile, out.data_ptr<float>(),
                   plan.has_value() ? plan->data_ptr() : nullptr,
                   scratch_bytes ? scratch.data_ptr() : nullptr, scratch_bytes, current_stream(x)),
               "half_linear_weight_gradient");
  return out;
}

// dx [batch, seq, in] in `like`'s type (the activations'), or an EMPTY tensor where the
// route does not serve the call.
Tensor half_linear_input_gradient(int64_t out64, int64_t in64, const Tensor& grad, bool grad_is_planes,
                                  const Tensor& image, int64_t values_type, const Tensor& like,
                                  int64_t batch64, int64_t seq64) {
  const int tile = tile_code_of(like);
  const c10::DeviceGuard guard(like.device());
  const int out_f = to_int(out64, "out"), in_f = to_int(in64, "in"), batch = to_int(batch64, "batch"),
            seq = to_int(seq64, "seq");
  const Tensor g = grad.contiguous();
  Tensor dx = at::empty({batch, seq, in_f}, like.options());
  const int st = sputnik_hip_sparse_linear_half_input_gradient(
      out_f, in_f, seq, batch, g.data_ptr(), grad_is_planes ? SPUTNIK_HIP_F32 : tile, image.data_ptr(),
      static_cast<int>(values_type), tile, dx.data_ptr(), tile, current_stream(like));
  if (st == SPUTNIK_HIP_UNSUPPORTED) return at::empty({0}, like.options());
  check_status(st, "half_linear_input_gradient");
  return dx;
}

// sddmm_sum on one float32 and one half operand (3-D, contiguous).  False: the library
// does not serve the pair on this shape -- the caller widens the half operand.
bool sddmm_sum_mixed(int m, int n, const Tensor& row_indices, const Tensor& row_offsets,
                     const Tensor& column_indices, const Tensor& lhs_in, const Tensor& rhs_in,
                     const c10::optional<Tensor>& plan, Tensor* result) {
  const auto is_half = [](const Tensor& t) {
    return t.scalar_type() == at::kHalf || t.scalar_type() == at::kBFloat16;
  };
  if (!((lhs_in.scalar_type() == at::kFloat && is_half(rhs_in)) ||
        (rhs_in.scalar_type() == at::kFloat && is_half(lhs_in))))
    return false;
  if (!lhs_in.is_cuda() || lhs_in.dim() != 3 || rhs_in.dim() != 3 ||
      lhs_in.device() != rhs_in.device() || lhs_in.size(0) != rhs_in.size(0) ||
      lhs_in.size(-1) != rhs_in.size(-1) || lhs_in.size(1) != m || rhs_in.size(1) != n)
    return false;   // (the plain path reports what is wrong with the shapes)
  const Tensor lhs = lhs_in.contiguous(), rhs = rhs_in.contiguous();
  const c10::DeviceGuard guard(lhs.device());
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, lhs);
  const int replicas = to_int(lhs.size(0), "replicas"), k = to_int(lhs.size(2), "k");
  const int lhs_code = type_code(lhs.scalar_type()), rhs_code = type_code(rhs.scalar_type());
  const size_t scratch_bytes = sputnik_hip_sddmm_sum_mixed_scratch_bytes(
      m, k, n, topo.nonzeros, replicas, lhs_code, rhs_code);
  if (scratch_bytes == 0) return false;
  const size_t ws_bytes = sputnik_hip_sddmm_sum_workspace_bytes(m, k, n, topo.nonzeros);
  Tensor workspace;
  void* ws = nullptr;
  if (plan.has_value()) {
    check_plan(*plan, ws_bytes, lhs);
    ws = ws_bytes ? plan->data_ptr() : nullptr;
  } else if (ws_bytes > 0) {
    workspace = at::empty({static_cast<int64_t>(ws_bytes)}, lhs.options().dtype(at::kByte));
    ws = workspace.data_ptr();
  }
  Tensor scratch = at::empty({static_cast<int64_t>(scratch_bytes)}, lhs.options().dtype(at::kByte));
  Tensor out = at::empty({topo.nonzeros}, lhs.options().dtype(at::kFloat));
  const int status = sputnik_hip_sddmm_sum_mixed(
      m, k, n, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
      topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(), lhs.data_ptr(), lhs_code,
      static_cast<int64_t>(m) * k, rhs.data_ptr(), rhs_code, static_cast<int64_t>(n) * k,
      out.data_ptr<float>(), ws, ws_bytes, plan.has_value() ? 1 : 0, scratch.data_ptr(),
      scratch_bytes, current_stream(lhs));
  if (status == SPUTNIK_HIP_UNSUPPORTED) return false;
  check_status(status, "sddmm_sum (float32 x half)");
  *result = out;
  return true;
}

// lhs / rhs float32, float16 or bfloat16 (handed to the kernels as they are; operands
// of different types take the wider one).  out_type: -1 / SPUTNIK_HIP_F32 = float32 --
// the reference's output type (src/sddmm_cuda.cu:43) --, or the operands' half type.
Tensor sddmm_impl(int64_t m64, int64_t n64, const Tensor& row_indices, const Tensor& row_offsets,
                  const Tensor& column_indices, const Tensor& lhs_in, const Tensor& rhs_in,
                  const c10::optional<Tensor>& plan, bool sum_replicas = false,
                  int64_t out_type = -1) {
  const int m = to_int(m64, "m"), n = to_int(n64, "n");
  if (sum_replicas && lhs_in.scalar_type() != rhs_in.scalar_type()) {
    // a (float32, half) pair: the matrix-core route takes the float32 operand as two half
    // planes instead of widening the half one (sputnik_hip.h: sddmm_sum_mixed)
    Tensor mixed;
    if (sddmm_sum_mixed(m, n, row_indices, row_offsets, column_indices, lhs_in, rhs_in, plan, &mixed))
      return mixed;
  }
  const auto st = at::promote_types(lhs_in.scalar_type(), rhs_in.scalar_type());
  const Tensor lhs = as_storage(lhs_in, "lhs_matrix").to(st);
  const Tensor rhs = as_storage(rhs_in, "rhs_matrix").to(st);
  const int in_code = type_code(st);
  TORCH_CHECK(lhs.device() == rhs.device(), "lhs_matrix and rhs_matrix must be on one device");
  TORCH_CHECK(lhs.dim() == 2 || lhs.dim() == 3, "expected 2-dim or 3-dim lhs_matrix, got ",
              lhs.dim());
  TORCH_CHECK(rhs.dim() == lhs.dim(), "rhs_matrix and lhs_matrix must match number of dims");
  TORCH_CHECK(lhs.size(-1) == rhs.size(-1), "last dim of input matrices must match: ",
              lhs.size(-1), " vs ", rhs.size(-1));
  const c10::DeviceGuard guard(lhs.device());
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, lhs);

  const int dim_offset = static_cast<int>(lhs.dim()) - 2;
  const int replicas = dim_offset == 1 ? to_int(lhs.size(0), "replicas") : 1;
  const int k = to_int(lhs.size(dim_offset + 1), "k");
  TORCH_CHECK(lhs.size(dim_offset) == m, "first dim of lhs_matrix (", lhs.size(dim_offset),
              ") must match output rows m = ", m);
  TORCH_CHECK(rhs.size(dim_offset) == n, "first dim of rhs_matrix (", rhs.size(dim_offset),
              ") must match output cols n = ", n);
  TORCH_CHECK(replicas == 1 || rhs.size(0) == replicas,
              "first dim of lhs_matrix and rhs_matrix must match");
  const int out_code = out_type < 0 ? SPUTNIK_HIP_F32 : static_cast<int>(out_type);
  TORCH_CHECK(out_code == SPUTNIK_HIP_F32 || (out_code == in_code && !sum_replicas),
              "sddmm: the output is float32 or has the operands' (half) type");
  const auto out_options = lhs.options().dtype(out_code == SPUTNIK_HIP_F32 ? at::kFloat : st);

  // 1-D whenever there is a single replica, as src/sddmm_cuda.cu:43 does.
  Tensor out = (replicas == 1 || sum_replicas)
                   ? at::empty({topo.nonzeros}, out_options)
                   : at::empty({replicas, topo.nonzeros}, out_options);
  const size_t ws_bytes = sum_replicas
                              ? sputnik_hip_sddmm_sum_workspace_bytes(m, k, n, topo.nonzeros)
                              : sputnik_hip_sddmm_workspace_bytes(m, k, n, topo.nonzeros);
  Tensor workspace;
  void* ws = nullptr;
  if (plan.has_value()) {
    check_plan(*plan, ws_bytes, lhs);
    ws = ws_bytes ? plan->data_ptr() : nullptr;
  } else if (ws_bytes > 0) {
    workspace = at::empty({static_cast<int64_t>(ws_bytes)}, lhs.options().dtype(at::kByte));
    ws = workspace.data_ptr();
  }
  const int planned = plan.has_value() ? 1 : 0;
  if (sum_replicas) {
    // sum over the batch inside the call (sputnik_hip.h: sddmm_sum_batched)
    const size_t scratch_bytes =
        sputnik_hip_sddmm_sum_scratch_bytes(m, k, n, topo.nonzeros, replicas);
    Tensor scratch;
    if (scratch_bytes > 0)
      scratch = at::empty({static_cast<int64_t>(scratch_bytes)}, lhs.options().dtype(at::kByte));
    check_status(sputnik_hip_sddmm_sum_typed(
                     m, k, n, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
                     topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(),
                     lhs.data_ptr(), static_cast<int64_t>(m) * k, rhs.data_ptr(),
                     static_cast<int64_t>(n) * k, in_code, out.data_ptr<float>(), ws, ws_bytes,
                     planned, scratch_bytes ? scratch.data_ptr() : nullptr, scratch_bytes,
                     current_stream(lhs)),
                 planned ? "sddmm_sum_planned" : "sddmm_sum");
    return out;
  }
  const int status = sputnik_hip_sddmm_typed(
      m, k, n, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
      topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(), lhs.data_ptr(),
      static_cast<int64_t>(m) * k, rhs.data_ptr(), static_cast<int64_t>(n) * k, in_code,
      out.data_ptr(), topo.nonzeros, out_code, ws, ws_bytes, planned, current_stream(lhs));
  if (status == SPUTNIK_HIP_UNSUPPORTED && out_code != SPUTNIK_HIP_F32) {
    // a half output of a product that takes several passes: float32 result, rounded once
    return sddmm_impl(m64, n64, row_indices, row_offsets, column_indices, lhs, rhs, plan, false,
                      SPUTNIK_HIP_F32).to(st);
  }
  check_status(status, planned ? "sddmm_planned" : "sddmm");
  return out;
}

// sddmm with the result stored in the operands' half type (float32 operands: as sddmm)
Tensor sddmm_narrow(int64_t m, int64_t n, const Tensor& row_indices, const Tensor& row_offsets,
                    const Tensor& column_indices, const Tensor& lhs, const Tensor& rhs) {
  const int code = type_code(at::promote_types(lhs.scalar_type(), rhs.scalar_type()));
  return sddmm_impl(m, n, row_indices, row_offsets, column_indices, lhs, rhs, c10::nullopt, false,
                    code);
}

Tensor sddmm(int64_t m, int64_t n, const Tensor& row_indices, const Tensor& row_offsets,
             const Tensor& column_indices, const Tensor& lhs, const Tensor& rhs) {
  return sddmm_impl(m, n, row_indices, row_offsets, column_indices, lhs, rhs, c10::nullopt);
}

// ---------------------------------------------------------------------------
// Static topologies: run the topology-only pre-pass once, keep the result (a
// uint8 "plan" tensor) and hand it to the *_planned ops.  The reference
// re-derives everything per call (src/spmm_cuda.cu:48-57).
// ---------------------------------------------------------------------------
Tensor make_plan_tensor(size_t bytes, const Tensor& like) {
  return at::empty({static_cast<int64_t>(std::max<size_t>(bytes, 16))},
                   like.options().dtype(at::kByte));
}

Tensor spmm_plan(int64_t m64, int64_t k64, int64_t n64, const Tensor& row_indices,
                 const Tensor& row_offsets, const Tensor& column_indices) {
  const int m = to_int(m64, "m"), k = to_int(k64, "k"), n = to_int(n64, "n");
  TORCH_CHECK(row_offsets.is_cuda(), "row_offsets must be a GPU (HIP) tensor");
  const c10::DeviceGuard guard(row_offsets.device());
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, row_offsets);
  const size_t bytes = sputnik_hip_spmm_workspace_bytes(m, k, n, topo.nonzeros);
  Tensor plan = make_plan_tensor(bytes, row_offsets);
  check_status(sputnik_hip_spmm_plan(m, k, n, topo.nonzeros, topo.row_indices.data_ptr<int>(),
                                     topo.row_offsets.data_ptr<int>(),
                                     topo.column_indices.data_ptr<int>(),
                                     bytes ? plan.data_ptr() : nullptr, bytes,
                                     current_stream(row_offsets)),
               "spmm_plan");
  return plan;
}

Tensor spmm_planned(int64_t m, int64_t k, const Tensor& values, const Tensor& row_indices,
                    const Tensor& row_offsets, const Tensor& column_indices, const Tensor& dense,
                    const Tensor& plan) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, false,
                   "spmm_planned", c10::nullopt, false, plan);
}

Tensor left_spmm_planned(int64_t m, int64_t k, const Tensor& values, const Tensor& row_indices,
                         const Tensor& row_offsets, const Tensor& column_indices,
                         const Tensor& dense, const Tensor& plan) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, true,
                   "left_spmm_planned", c10::nullopt, false, plan);
}

Tensor sddmm_plan(int64_t m64, int64_t n64, int64_t k64, const Tensor& row_indices,
                  const Tensor& row_offsets, const Tensor& column_indices) {
  const int m = to_int(m64, "m"), n = to_int(n64, "n"), k = to_int(k64, "k");
  TORCH_CHECK(row_offsets.is_cuda(), "row_offsets must be a GPU (HIP) tensor");
  const c10::DeviceGuard guard(row_offsets.device());
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, row_offsets);
  const size_t bytes = sputnik_hip_sddmm_workspace_bytes(m, k, n, topo.nonzeros);
  Tensor plan = make_plan_tensor(bytes, row_offsets);
  check_status(sputnik_hip_sddmm_plan(m, k, n, topo.nonzeros, topo.row_indices.data_ptr<int>(),
                                      topo.row_offsets.data_ptr<int>(),
                                      topo.column_indices.data_ptr<int>(),
                                      bytes ? plan.data_ptr() : nullptr, bytes,
                                      current_stream(row_offsets)),
               "sddmm_plan");
  return plan;
}

// Several summed products of one shape in one call: the weight gradients of a group of
// projections (sputnik_hip.h: sddmm_sum_group_planned).  float32, 3-D contiguous operands,
// one planned workspace per weight; rhs is shared.
std::vector<Tensor> sddmm_sum_group_planned(int64_t m64, int64_t n64,
                                            const std::vector<Tensor>& row_indices,
                                            const std::vector<Tensor>& row_offsets,
                                            const std::vector<Tensor>& column_indices,
                                            const std::vector<Tensor>& lhs_in, const Tensor& rhs_in,
                                            const std::vector<Tensor>& plans) {
  const int m = to_int(m64, "m"), n = to_int(n64, "n");
  const size_t count = lhs_in.size();
  TORCH_CHECK(count >= 1 && count <= 4, "sddmm_sum_group: one to four products, got ", count);
  TORCH_CHECK(row_indices.size() == count && row_offsets.size() == count &&
                  column_indices.size() == count && plans.size() == count,
              "sddmm_sum_group: one topology, lhs and plan per product");
  const Tensor rhs = as_float(rhs_in, "rhs_matrix");
  TORCH_CHECK(rhs.dim() == 3 && rhs.size(1) == n, "sddmm_sum_group: rhs_matrix should be [replicas, n = ",
              n, ", k]");
  const int replicas = to_int(rhs.size(0), "replicas"), k = to_int(rhs.size(2), "k");
  const c10::DeviceGuard guard(rhs.device());
  std::vector<Tensor> lhs(count), outs(count), scratch(count);
  std::vector<Topology> topo;
  std::vector<sputnik_hip_sddmm_sum_problem> problems(count);
  for (size_t p = 0; p < count; ++p) {
    lhs[p] = as_float(lhs_in[p], "lhs_matrix");
    TORCH_CHECK(lhs[p].device() == rhs.device() && lhs[p].dim() == 3 && lhs[p].size(0) == replicas &&
                    lhs[p].size(1) == m && lhs[p].size(2) == k,
                "sddmm_sum_group: lhs_matrix ", p, " should be [", replicas, ", ", m, ", ", k, "] on ",
                rhs.device());
    topo.push_back(check_topology(m, row_indices[p], row_offsets[p], column_indices[p], lhs[p]));
    const int nnz = topo[p].nonzeros;
    outs[p] = at::empty({nnz}, rhs.options());
    const size_t ws_bytes = sputnik_hip_sddmm_sum_workspace_bytes(m, k, n, nnz);
    check_plan(plans[p], ws_bytes, rhs);
    const size_t scratch_bytes = sputnik_hip_sddmm_sum_scratch_bytes(m, k, n, nnz, replicas);
    if (scratch_bytes > 0)
      scratch[p] = at::empty({static_cast<int64_t>(scratch_bytes)}, rhs.options().dtype(at::kByte));
    sputnik_hip_sddmm_sum_problem& q = problems[p];
    q.row_indices = topo[p].row_indices.data_ptr<int>();
    q.row_offsets = topo[p].row_offsets.data_ptr<int>();
    q.column_indices = topo[p].column_indices.data_ptr<int>();
    q.lhs = lhs[p].data_ptr<float>();
    q.rhs = rhs.data_ptr<float>();
    q.out = outs[p].data_ptr<float>();
    q.workspace = ws_bytes ? plans[p].data_ptr() : nullptr;
    q.workspace_bytes = ws_bytes;
    q.scratch = scratch_bytes ? scratch[p].data_ptr() : nullptr;
    q.scratch_bytes = scratch_bytes;
    q.nonzeros = nnz;
  }
  check_status(sputnik_hip_sddmm_sum_group_planned(m, k, n, replicas, static_cast<int>(count),
                                                   problems.data(), static_cast<int64_t>(m) * k,
                                                   static_cast<int64_t>(n) * k, current_stream(rhs)),
               "sddmm_sum_group_planned");
  return outs;
}

// the summed product has a plan of its own (sputnik_hip.h: sddmm_sum_plan)
Tensor sddmm_sum_plan(int64_t m64, int64_t n64, int64_t k64, const Tensor& row_indices,
                      const Tensor& row_offsets, const Tensor& column_indices) {
  const int m = to_int(m64, "m"), n = to_int(n64, "n"), k = to_int(k64, "k");
  TORCH_CHECK(row_offsets.is_cuda(), "row_offsets must be a GPU (HIP) tensor");
  const c10::DeviceGuard guard(row_offsets.device());
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, row_offsets);
  const size_t bytes = sputnik_hip_sddmm_sum_workspace_bytes(m, k, n, topo.nonzeros);
  Tensor plan = make_plan_tensor(bytes, row_offsets);
  check_status(sputnik_hip_sddmm_sum_plan(m, k, n, topo.nonzeros,
                                          topo.row_indices.data_ptr<int>(),
                                          topo.row_offsets.data_ptr<int>(),
                                          topo.column_indices.data_ptr<int>(),
                                          bytes ? plan.data_ptr() : nullptr, bytes,
                                          current_stream(row_offsets)),
               "sddmm_sum_plan");
  return plan;
}

Tensor sddmm_planned(int64_t m, int64_t n, const Tensor& row_indices, const Tensor& row_offsets,
                     const Tensor& column_indices, const Tensor& lhs, const Tensor& rhs,
                     const Tensor& plan) {
  return sddmm_impl(m, n, row_indices, row_offsets, column_indices, lhs, rhs, plan);
}

// sum over the replicas of sddmm: the gradient of values shared by a batch
Tensor sddmm_sum(int64_t m, int64_t n, const Tensor& row_indices, const Tensor& row_offsets,
                 const Tensor& column_indices, const Tensor& lhs, const Tensor& rhs) {
  return sddmm_impl(m, n, row_indices, row_offsets, column_indices, lhs, rhs, c10::nullopt, true);
}

Tensor sddmm_sum_planned(int64_t m, int64_t n, const Tensor& row_indices,
                         const Tensor& row_offsets, const Tensor& column_indices,
                         const Tensor& lhs, const Tensor& rhs, const Tensor& plan) {
  return sddmm_impl(m, n, row_indices, row_offsets, column_indices, lhs, rhs, plan, true);
}

Tensor sparse_softmax_scaled(const Tensor& values_in, const Tensor& row_indices,
                             const Tensor& row_offsets, const Tensor& column_indices,
                             double scale) {
  // float16 / bfloat16 values are read and written as such (half the traffic of
  // src/softmax_cuda.cu:38-42), the result has the values' type
  const Tensor values = as_storage(values_in, "values");
  TORCH_CHECK(values.dim() == 1 || values.dim() == 2,
              "values should have 1 or 2 dimensions, got ", values.dim());
  const c10::DeviceGuard guard(values.device());
  const int m = to_int(row_indices.size(0), "m");
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, values);
  TORCH_CHECK(values.size(-1) == topo.nonzeros, "number of values (", values.size(-1),
              ") must equal the number of column_indices (", topo.nonzeros, ")");
  const int replicas = values.dim() == 2 ? to_int(values.size(0), "replicas") : 1;

  Tensor out = at::empty_like(values);
  check_status(sputnik_hip_sparse_softmax_typed(
                   m, /*n=*/-1, topo.nonzeros, replicas, values.data_ptr(), topo.nonzeros,
                   topo.row_indices.data_ptr<int>(), topo.row_offsets.data_ptr<int>(),
                   topo.column_indices.data_ptr<int>(), static_cast<float>(scale),
                   out.data_ptr(), topo.nonzeros, type_code(values.scalar_type()),
                   current_stream(values)),
               "sparse_softmax");
  return out;
}

Tensor sparse_softmax(const Tensor& values, const Tensor& row_indices, const Tensor& row_offsets,
                      const Tensor& column_indices) {
  return sparse_softmax_scaled(values, row_indices, row_offsets, column_indices, 1.0);
}

// grad of softmax(scale * x) w.r.t. x; softmax_out / grad_out [nnz] or [R,nnz].
Tensor sparse_softmax_backward(const Tensor& softmax_out_in, const Tensor& grad_out_in,
                               const Tensor& row_offsets_in, double scale) {
  // one storage type for both operands and the result (the wider one if they differ)
  const auto st = at::promote_types(softmax_out_in.scalar_type(), grad_out_in.scalar_type());
  const Tensor y = as_storage(softmax_out_in, "softmax_out").to(st);
  const Tensor g = as_storage(grad_out_in, "grad_out").to(st);
  TORCH_CHECK(y.sizes() == g.sizes(), "softmax_out and grad_out must have one shape, got ",
              y.sizes(), " and ", g.sizes());
  TORCH_CHECK(y.dim() == 1 || y.dim() == 2, "softmax_out should have 1 or 2 dimensions, got ",
              y.dim());
  TORCH_CHECK(y.device() == g.device(), "softmax_out and grad_out must be on one device");
  const c10::DeviceGuard guard(y.device());
  const Tensor row_offsets = as_index(row_offsets_in, "row_offsets", y);
  TORCH_CHECK(row_offsets.size(0) >= 1, "row_offsets must not be empty");
  const int m = to_int(row_offsets.size(0) - 1, "m");
  const int nonzeros = to_int(y.size(-1), "nonzeros");
  const int replicas = y.dim() == 2 ? to_int(y.size(0), "replicas") : 1;
  Tensor out = at::empty_like(y);
  check_status(sputnik_hip_sparse_softmax_backward_typed(
                   m, nonzeros, replicas, y.data_ptr(), nonzeros, g.data_ptr(), nonzeros,
                   row_offsets.data_ptr<int>(), static_cast<float>(scale), out.data_ptr(),
                   nonzeros, type_code(st), current_stream(y)),
               "sparse_softmax_backward");
  return out;
}

// Returns {values_t, row_offsets_t, column_indices_t} (+ {permutation} when
// asked).  values may be [nnz] (reference contract) or [R,nnz] (extension).
std::vector<Tensor> csr_transpose_impl(int64_t m64, int64_t n64, const Tensor& values_in,
                                       const Tensor& row_offsets_in,
                                       const Tensor& column_indices_in, bool want_permutation,
                                       bool checked) {
  const int m = to_int(m64, "m"), n = to_int(n64, "n");
  // float16 / bfloat16 values are read as they are by the gather that moves them
  // (sputnik_hip_csr_transpose_typed); the transposed values are float32 either way
  const Tensor values = as_storage(values_in, "values");
  const bool half_values = values.scalar_type() != at::kFloat;
  TORCH_CHECK(values.dim() == 1 || values.dim() == 2,
              "values should have 1 (or, as an extension, 2) dimensions, got ", values.dim());
  const c10::DeviceGuard guard(values.device());
  const Tensor row_offsets = as_index(row_offsets_in, "row_offsets", values);
  const Tensor column_indices = as_index(column_indices_in, "column_indices", values);
  TORCH_CHECK(values.size(-1) == column_indices.size(0),
              "expected same number of values and indices, got ", values.size(-1), " and ",
              column_indices.size(0));
  TORCH_CHECK(row_offsets.size(0) == m + 1, "expected m+1 row offsets, got ",
              row_offsets.size(0), " for m = ", m);
  const int nonzeros = to_int(column_indices.size(0), "nonzeros");
  const int replicas = values.dim() == 2 ? to_int(values.size(0), "replicas") : 1;

  const auto index_options = values.options().dtype(at::kInt);
  Tensor out_values = at::empty(values.sizes(), values.options().dtype(at::kFloat));
  Tensor out_row_offsets = at::empty({n + 1}, index_options);
  Tensor out_column_indices = at::empty({nonzeros}, index_options);
  Tensor permutation;
  if (want_permutation || half_values) permutation = at::empty({nonzeros}, index_options);

  const size_t ws_bytes = sputnik_hip_csr_transpose_workspace_bytes(m, n, nonzeros);
  Tensor workspace =
      at::empty({static_cast<int64_t>(ws_bytes)}, values.options().dtype(at::kByte));
  // csr_transpose itself is asynchronous like the reference's (src/transpose_cuda.cu:90-99:
  // no host round trip).  The form that also returns the permutation takes a `checked` flag:
  // a cache that is going to KEEP the result (once per static topology) sets it and gets the
  // checked entry, which waits for the stream and reports a pattern the transpose is not
  // defined for (a row storing a column twice, a column out of range) instead of handing out
  // a silently wrong permutation; a per-call user leaves it off and stays asynchronous and
  // capturable (the wait is illegal inside a stream capture).
  checked = checked && want_permutation;
  int status;
  if (half_values) {
    status = sputnik_hip_csr_transpose_typed(
        m, n, nonzeros, replicas, values.data_ptr(), type_code(values.scalar_type()), nonzeros,
        row_offsets.data_ptr<int>(), column_indices.data_ptr<int>(), out_values.data_ptr<float>(),
        nonzeros, out_row_offsets.data_ptr<int>(), out_column_indices.data_ptr<int>(),
        permutation.data_ptr<int>(), workspace.data_ptr(), ws_bytes, checked ? 1 : 0,
        current_stream(values));
  } else {
    const auto entry =
        checked ? sputnik_hip_csr_transpose_checked : sputnik_hip_csr_transpose;
    status = entry(m, n, nonzeros, replicas, values.data_ptr<float>(), nonzeros,
                   row_offsets.data_ptr<int>(), column_indices.data_ptr<int>(),
                   out_values.data_ptr<float>(), nonzeros, out_row_offsets.data_ptr<int>(),
                   out_column_indices.data_ptr<int>(),
                   want_permutation ? permutation.data_ptr<int>() : nullptr, workspace.data_ptr(),
                   ws_bytes, current_stream(values));
  }
  TORCH_CHECK(!(checked && status == SPUTNIK_HIP_INVALID_ARGUMENT),
              "torch_sputnik::csr_transpose_with_permutation: the pattern is not a valid CSR "
              "matrix for a transpose (a row stores a column twice, or a column index is out of "
              "range)");
  check_status(status, "csr_transpose");
  std::vector<Tensor> out{out_values, out_row_offsets, out_column_indices};
  if (want_permutation) out.push_back(permutation);
  return out;
}

std::vector<Tensor> csr_transpose(int64_t m, int64_t n, const Tensor& values,
                                  const Tensor& row_offsets, const Tensor& column_indices) {
  return csr_transpose_impl(m, n, values, row_offsets, column_indices, false, false);
}

std::vector<Tensor> csr_transpose_with_permutation(int64_t m, int64_t n, const Tensor& values,
                                                   const Tensor& row_offsets,
                                                   const Tensor& column_indices, bool checked) {
  return csr_transpose_impl(m, n, values, row_offsets, column_indices, true, checked);
}

// Fused softmax(scale * sddmm(q, k)) @ v over a fixed mask
// (modules/sparse_attention.py:66-82).  q [R,m,d] / [m,d]; k, v [R,n,d] / [n,d].
// Shapes the fused kernel does not serve are composed from the three operators.
std::vector<Tensor> sparse_attention_impl(const Tensor& q_in, const Tensor& k_in,
                                          const Tensor& v_in, const Tensor& row_indices,
                                          const Tensor& row_offsets,
                                          const Tensor& column_indices, double scale,
                                          bool want_lse,
                                          const c10::optional<Tensor>& plan = c10::nullopt) {
  const Tensor q = as_float(q_in, "query");
  const Tensor k = as_float(k_in, "key");
  const Tensor v = as_float(v_in, "value");
  TORCH_CHECK(q.dim() == 2 || q.dim() == 3, "expected 2-dim or 3-dim query, got ", q.dim());
  TORCH_CHECK(k.dim() == q.dim() && v.dim() == q.dim(), "query, key, value must match in dims");
  TORCH_CHECK(k.sizes() == v.sizes(), "key and value must have one shape");
  TORCH_CHECK(q.size(-1) == k.size(-1), "query and key must have one head dimension");
  TORCH_CHECK(q.device() == k.device() && q.device() == v.device(),
              "query, key, value must be on one device");
  const c10::DeviceGuard guard(q.device());
  const int m = to_int(q.size(-2), "m"), n = to_int(k.size(-2), "n");
  const int d = to_int(q.size(-1), "d");
  const int replicas = q.dim() == 3 ? to_int(q.size(0), "replicas") : 1;
  TORCH_CHECK(q.dim() == 2 || k.size(0) == replicas, "first dim of query and key must match");
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, q);

  if (!sputnik_hip_sparse_attention_supported(m, n, d, topo.nonzeros)) {
    TORCH_CHECK(!want_lse, "sparse_attention_with_lse: head dimension ", d,
                " is not served by the fused kernel (64 is)");
    Tensor weights = sparse_softmax_scaled(
        sddmm(m, n, topo.row_indices, topo.row_offsets, topo.column_indices, q, k),
        topo.row_indices, topo.row_offsets, topo.column_indices, scale);
    return {spmm(m, n, weights, topo.row_indices, topo.row_offsets, topo.column_indices, v)};
  }
  Tensor out = at::empty_like(q);
  Tensor lse;
  if (want_lse)
    lse = q.dim() == 3 ? at::empty({replicas, m}, q.options()) : at::empty({m}, q.options());
  const size_t ws_bytes = sputnik_hip_sparse_attention_workspace_bytes(m, n, d, topo.nonzeros);
  if (plan.has_value()) {
    check_plan(*plan, ws_bytes, q);
    check_status(sputnik_hip_sparse_attention_forward_planned(
                     m, n, d, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
                     topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(),
                     q.data_ptr<float>(), static_cast<int64_t>(m) * d, k.data_ptr<float>(),
                     static_cast<int64_t>(n) * d, v.data_ptr<float>(),
                     static_cast<int64_t>(n) * d, static_cast<float>(scale),
                     out.data_ptr<float>(), static_cast<int64_t>(m) * d,
                     want_lse ? lse.data_ptr<float>() : nullptr, m, plan->data_ptr(), ws_bytes,
                     current_stream(q)),
                 "sparse_attention_planned");
    if (want_lse) return {out, lse};
    return {out};
  }
  Tensor workspace = at::empty({static_cast<int64_t>(ws_bytes)}, q.options().dtype(at::kByte));
  check_status(sputnik_hip_sparse_attention_forward(
                   m, n, d, topo.nonzeros, replicas, topo.row_indices.data_ptr<int>(),
                   topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(),
                   q.data_ptr<float>(), static_cast<int64_t>(m) * d, k.data_ptr<float>(),
                   static_cast<int64_t>(n) * d, v.data_ptr<float>(), static_cast<int64_t>(n) * d,
                   static_cast<float>(scale), out.data_ptr<float>(),
                   static_cast<int64_t>(m) * d, want_lse ? lse.data_ptr<float>() : nullptr, m,
                   workspace.data_ptr(), ws_bytes, current_stream(q)),
               "sparse_attention");
  if (want_lse) return {out, lse};
  return {out};
}

Tensor sparse_attention(const Tensor& q, const Tensor& k, const Tensor& v,
                        const Tensor& row_indices, const Tensor& row_offsets,
                        const Tensor& column_indices, double scale) {
  return sparse_attention_impl(q, k, v, row_indices, row_offsets, column_indices, scale, false)[0];
}

// {out, lse}: lse[r, i] = log sum_j exp(scale * <q_i, k_j>) over the stored j
// (-inf for rows without entries), what a backward needs to rebuild the weights.
std::vector<Tensor> sparse_attention_with_lse(const Tensor& q, const Tensor& k, const Tensor& v,
                                              const Tensor& row_indices,
                                              const Tensor& row_offsets,
                                              const Tensor& column_indices, double scale) {
  return sparse_attention_impl(q, k, v, row_indices, row_offsets, column_indices, scale, true);
}

// Empty plan (numel 0 is never returned; 16 bytes) when the fused kernel does not
// serve the shape: sparse_attention_planned then composes the three operators.
Tensor sparse_attention_plan(int64_t m64, int64_t n64, int64_t d64, const Tensor& row_indices,
                             const Tensor& row_offsets, const Tensor& column_indices) {
  const int m = to_int(m64, "m"), n = to_int(n64, "n"), d = to_int(d64, "d");
  TORCH_CHECK(row_offsets.is_cuda(), "row_offsets must be a GPU (HIP) tensor");
  const c10::DeviceGuard guard(row_offsets.device());
  const Topology topo = check_topology(m, row_indices, row_offsets, column_indices, row_offsets);
  const size_t bytes = sputnik_hip_sparse_attention_workspace_bytes(m, n, d, topo.nonzeros);
  Tensor plan = make_plan_tensor(bytes, row_offsets);
  if (sputnik_hip_sparse_attention_supported(m, n, d, topo.nonzeros))
    check_status(sputnik_hip_sparse_attention_plan(
                     m, n, d, topo.nonzeros, topo.row_indices.data_ptr<int>(),
                     topo.row_offsets.data_ptr<int>(), topo.column_indices.data_ptr<int>(),
                     plan.data_ptr(), bytes, current_stream(row_offsets)),
                 "sparse_attention_plan");
  return plan;
}

Tensor sparse_attention_planned(const Tensor& q, const Tensor& k, const Tensor& v,
                                const Tensor& row_indices, const Tensor& row_offsets,
                                const Tensor& column_indices, double scale, const Tensor& plan) {
  return sparse_attention_impl(q, k, v, row_indices, row_offsets, column_indices, scale, false,
                               plan)[0];
}

Tensor spmm_bias(int64_t m, int64_t k, const Tensor& values, const Tensor& row_indices,
                 const Tensor& row_offsets, const Tensor& column_indices, const Tensor& bias,
                 const Tensor& dense) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, false,
                   "spmm_bias", bias, false);
}

Tensor spmm_bias_relu(int64_t m, int64_t k, const Tensor& values, const Tensor& row_indices,
                      const Tensor& row_offsets, const Tensor& column_indices,
                      const Tensor& bias, const Tensor& dense) {
  return spmm_impl(m, k, values, row_indices, row_offsets, column_indices, dense, false,
                   "spmm_bias_relu", bias, true);
}

// ---------------------------------------------------------------------------
// many-mask family (tests/transformer/functions.py, tests/transformer/utils.py:17-38):
// b topologies, concatenated; replica r uses mask r / (R / b).
// ---------------------------------------------------------------------------
struct ManyMask {
  int masks, m, replicas, width;  // width = max nonzeros
  std::vector<int> nonzeros;
  Tensor row_indices, row_offsets, column_indices;  // flat int32
  bool uniform;                                     // every mask has `width` nonzeros
};

Tensor as_flat_index(const Tensor& t, const char* name, const Tensor& like) {
  TORCH_CHECK(t.scalar_type() == at::kInt || t.scalar_type() == at::kLong, name,
              " must be an int32 (or int64) tensor, got ", t.scalar_type());
  TORCH_CHECK(t.device() == like.device(), name, " must be on ", like.device(), ", got ",
              t.device());
  return t.to(at::kInt).contiguous().view({-1});
}

// `nonzeros` is the tensor tests/transformer/utils.py:36 builds with
// torch.tensor(nnzs): it lives on the host.  A device tensor is accepted but
// costs a synchronising copy.
ManyMask check_many_mask(int64_t b, int64_t m, const Tensor& nonzeros,
                         const c10::optional<Tensor>& row_indices, const Tensor& row_offsets,
                         const Tensor& column_indices, const Tensor& like, int64_t replicas) {
  ManyMask mm;
  mm.masks = to_int(b, "b");
  mm.m = to_int(m, "m");
  mm.replicas = to_int(replicas, "replicas");
  TORCH_CHECK(mm.masks > 0, "b must be positive");
  TORCH_CHECK(mm.replicas % mm.masks == 0, "the number of replicas (", replicas,
              ") must be a multiple of the number of masks (", b, ")");
  TORCH_CHECK(nonzeros.numel() == mm.masks, "nonzeros should have b = ", b, " entries, got ",
              nonzeros.numel());
  TORCH_CHECK(at::isIntegralType(nonzeros.scalar_type(), false), "nonzeros must be integral");
  const Tensor host = nonzeros.to(at::kCPU, at::kLong).contiguous();
  int64_t total = 0;
  mm.width = 0;
  mm.uniform = true;
  for (int i = 0; i < mm.masks; ++i) {
    const int64_t v = host.data_ptr<int64_t>()[i];
    mm.nonzeros.push_back(to_int(v, "nonzeros[i]"));
    total += v;
    mm.width = std::max(mm.width, mm.nonzeros.back());
  }
  for (int v : mm.nonzeros) mm.uniform = mm.uniform && v == mm.width;
  mm.row_offsets = as_flat_index(row_offsets, "row_offsets", like);
  mm.column_indices = as_flat_index(column_indices, "column_indices", like);
  TORCH_CHECK(mm.row_offsets.numel() == static_cast<int64_t>(mm.masks) * (mm.m + 1),
              "row_offsets should have b * (m + 1) = ", static_cast<int64_t>(mm.masks) * (mm.m + 1),
              " entries, got ", mm.row_offsets.numel());
  TORCH_CHECK(mm.column_indices.numel() == total, "column_indices should have sum(nonzeros) = ",
              total, " entries, got ", mm.column_indices.numel());
  if (row_indices.has_value()) {
    mm.row_indices = as_flat_index(*row_indices, "row_indices", like);
    TORCH_CHECK(mm.row_indices.numel() == static_cast<int64_t>(mm.masks) * mm.m,
                "row_indices should have b * m = ", static_cast<int64_t>(mm.masks) * mm.m,
                " entries, got ", mm.row_indices.numel());
  }
  return mm;
}

Tensor many_mask_values(const Tensor& t, const ManyMask& mm, const char* name) {
  TORCH_CHECK(t.dim() == 2, name, " should be [replicas, max(nonzeros)], got ", t.sizes());
  TORCH_CHECK(t.size(0) == mm.replicas, name, ": expected ", mm.replicas, " replicas, got ",
              t.size(0));
  TORCH_CHECK(t.size(1) >= mm.width, name, ": rows must hold max(nonzeros) = ", mm.width,
              " entries, got ", t.size(1));
  return t;
}

Tensor spmm_many_mask(int64_t b, int64_t m64, int64_t k64, const Tensor& nonzeros,
                      const Tensor& values_in, const Tensor& row_indices,
                      const Tensor& row_offsets, const Tensor& column_indices,
                      const Tensor& dense_in) {
  const Tensor values = as_float(values_in, "values");
  const Tensor dense = as_float(dense_in, "dense");
  TORCH_CHECK(dense.dim() == 3, "dense should be [replicas, k, n], got ", dense.sizes());
  TORCH_CHECK(dense.device() == values.device(), "values and dense must be on one device");
  const c10::DeviceGuard guard(values.device());
  const ManyMask mm = check_many_mask(b, m64, nonzeros, row_indices, row_offsets, column_indices,
                                      values, dense.size(0));
  many_mask_values(values, mm, "values");
  const int k = to_int(k64, "k"), n = to_int(dense.size(2), "n");
  TORCH_CHECK(dense.size(1) == k, "inner matrix dimensions must match: dense has ",
              dense.size(1), " rows, k = ", k);
  Tensor out = at::empty({mm.replicas, mm.m, n}, values.options());
  const size_t ws_bytes = sputnik_hip_spmm_workspace_bytes(mm.m, k, n, mm.width);
  Tensor workspace;
  if (ws_bytes > 0)
    workspace = at::empty({static_cast<int64_t>(ws_bytes)}, values.options().dtype(at::kByte));
  check_status(sputnik_hip_spmm_many_mask(
                   mm.masks, mm.m, k, n, mm.nonzeros.data(), mm.replicas,
                   mm.row_indices.data_ptr<int>(), values.data_ptr<float>(), values.size(1),
                   mm.row_offsets.data_ptr<int>(), mm.column_indices.data_ptr<int>(),
                   dense.data_ptr<float>(), static_cast<int64_t>(k) * n, out.data_ptr<float>(),
                   static_cast<int64_t>(mm.m) * n, ws_bytes ? workspace.data_ptr() : nullptr,
                   ws_bytes, current_stream(values)),
               "spmm_many_mask");
  return out;
}

Tensor sddmm_many_mask(int64_t b, int64_t m64, int64_t n64, const Tensor& nonzeros,
                       const Tensor& row_indices, const Tensor& row_offsets,
                       const Tensor& column_indices, const Tensor& lhs_in, const Tensor& rhs_in) {
  const Tensor lhs = as_float(lhs_in, "lhs_matrix");
  const Tensor rhs = as_float(rhs_in, "rhs_matrix");
  TORCH_CHECK(lhs.dim() == 3 && rhs.dim() == 3, "expected 3-dim lhs_matrix and rhs_matrix");
  TORCH_CHECK(lhs.device() == rhs.device(), "lhs_matrix and rhs_matrix must be on one device");
  TORCH_CHECK(lhs.size(0) == rhs.size(0), "first dim of lhs_matrix and rhs_matrix must match");
  TORCH_CHECK(lhs.size(2) == rhs.size(2), "last dim of input matrices must match");
  const c10::DeviceGuard guard(lhs.device());
  const ManyMask mm = check_many_mask(b, m64, nonzeros, row_indices, row_offsets, column_indices,
                                      lhs, lhs.size(0));
  const int n = to_int(n64, "n"), k = to_int(lhs.size(2), "k");
  TORCH_CHECK(lhs.size(1) == mm.m, "lhs_matrix should have m = ", mm.m, " rows");
  TORCH_CHECK(rhs.size(1) == n, "rhs_matrix should have n = ", n, " rows");
  // Entries past a replica's own count stay zero.
  Tensor out = mm.uniform ? at::empty({mm.replicas, mm.width}, lhs.options())
                          : at::zeros({mm.replicas, mm.width}, lhs.options());
  // (one plan per mask: all masks run in one launch, csrc/many_mask.hip)
  const size_t ws_bytes = sputnik_hip_sddmm_many_mask_workspace_bytes(
      mm.masks, mm.m, k, n, mm.width);
  Tensor workspace;
  if (ws_bytes > 0)
    workspace = at::empty({static_cast<int64_t>(ws_bytes)}, lhs.options().dtype(at::kByte));
  check_status(sputnik_hip_sddmm_many_mask(
                   mm.masks, mm.m, k, n, mm.nonzeros.data(), mm.replicas,
                   mm.row_indices.data_ptr<int>(), mm.row_offsets.data_ptr<int>(),
                   mm.column_indices.data_ptr<int>(), lhs.data_ptr<float>(),
                   static_cast<int64_t>(mm.m) * k, rhs.data_ptr<float>(),
                   static_cast<int64_t>(n) * k, out.data_ptr<float>(), mm.width,
                   ws_bytes ? workspace.data_ptr() : nullptr, ws_bytes, current_stream(lhs)),
               "sddmm_many_mask");
  return out;
}

Tensor sparse_softmax_many_mask_scaled(int64_t b, int64_t m64, const Tensor& nonzeros,
                                       const Tensor& values_in, const Tensor& row_indices,
                                       const Tensor& row_offsets, const Tensor& column_indices,
                                       double scale) {
  const Tensor values = as_float(values_in, "values");
  const c10::DeviceGuard guard(values.device());
  TORCH_CHECK(values.dim() == 2, "values should be [replicas, max(nonzeros)]");
  const ManyMask mm = check_many_mask(b, m64, nonzeros, row_indices, row_offsets, column_indices,
                                      values, values.size(0));
  many_mask_values(values, mm, "values");
  Tensor out = mm.uniform && values.size(1) == mm.width ? at::empty_like(values)
                                                        : at::zeros_like(values);
  check_status(sputnik_hip_sparse_softmax_many_mask(
                   mm.masks, mm.m, mm.nonzeros.data(), mm.replicas, values.data_ptr<float>(),
                   values.size(1), mm.row_indices.data_ptr<int>(),
                   mm.row_offsets.data_ptr<int>(), mm.column_indices.data_ptr<int>(),
                   static_cast<float>(scale), out.data_ptr<float>(), out.size(1),
                   current_stream(values)),
               "sparse_softmax_many_mask");
  return out;
}

Tensor sparse_softmax_many_mask(int64_t b, int64_t m, const Tensor& nonzeros,
                                const Tensor& values, const Tensor& row_indices,
                                const Tensor& row_offsets, const Tensor& column_indices) {
  return sparse_softmax_many_mask_scaled(b, m, nonzeros, values, row_indices, row_offsets,
                                         column_indices, 1.0);
}

Tensor sparse_softmax_backward_many_mask(int64_t b, int64_t m64, const Tensor& nonzeros,
                                         const Tensor& softmax_out_in, const Tensor& grad_out_in,
                                         const Tensor& row_offsets, double scale) {
  const Tensor y = as_float(softmax_out_in, "softmax_out");
  const Tensor g = as_float(grad_out_in, "grad_out");
  TORCH_CHECK(y.dim() == 2 && y.sizes() == g.sizes(),
              "softmax_out and grad_out should be [replicas, max(nonzeros)] and match");
  const c10::DeviceGuard guard(y.device());
  const int masks = to_int(b, "b"), m = to_int(m64, "m");
  TORCH_CHECK(masks > 0 && y.size(0) % masks == 0, "replicas must be a multiple of b");
  TORCH_CHECK(nonzeros.numel() == masks, "nonzeros should have b entries");
  const Tensor host = nonzeros.to(at::kCPU, at::kLong).contiguous();
  std::vector<int> counts;
  bool full = true;
  for (int i = 0; i < masks; ++i) {
    counts.push_back(to_int(host.data_ptr<int64_t>()[i], "nonzeros[i]"));
    TORCH_CHECK(counts.back() <= y.size(1), "nonzeros[i] exceeds the row length of softmax_out");
    full = full && counts.back() == y.size(1);
  }
  const Tensor offsets = as_flat_index(row_offsets, "row_offsets", y);
  TORCH_CHECK(offsets.numel() == static_cast<int64_t>(masks) * (m + 1),
              "row_offsets should have b * (m + 1) entries");
  Tensor out = full ? at::empty_like(y) : at::zeros_like(y);
  check_status(sputnik_hip_sparse_softmax_backward_many_mask(
                   masks, m, counts.data(), to_int(y.size(0), "replicas"), y.data_ptr<float>(),
                   y.size(1), g.data_ptr<float>(), g.size(1), offsets.data_ptr<int>(),
                   static_cast<float>(scale), out.data_ptr<float>(), out.size(1),
                   current_stream(y)),
               "sparse_softmax_backward_many_mask");
  return out;
}

std::vector<Tensor> csr_transpose_many_mask(int64_t b, int64_t m64, int64_t n64,
                                            const Tensor& nonzeros, const Tensor& values_in,
                                            const Tensor& row_offsets,
                                            const Tensor& column_indices) {
  const Tensor values = as_float(values_in, "values");
  const c10::DeviceGuard guard(values.device());
  TORCH_CHECK(values.dim() == 2, "values should be [replicas, max(nonzeros)]");
  const ManyMask mm = check_many_mask(b, m64, nonzeros, c10::nullopt, row_offsets,
                                      column_indices, values, values.size(0));
  many_mask_values(values, mm, "values");
  const int n = to_int(n64, "n");
  const auto index_options = values.options().dtype(at::kInt);
  Tensor out_values = mm.uniform && values.size(1) == mm.width ? at::empty_like(values)
                                                               : at::zeros_like(values);
  // [b, n + 1]: tests/transformer/utils.py:51-62 (diffsort_many_mask) indexes it per mask
  Tensor out_row_offsets = at::empty({mm.masks, n + 1}, index_options);
  Tensor out_column_indices = at::empty({mm.column_indices.numel()}, index_options);
  const size_t ws_bytes = sputnik_hip_csr_transpose_many_mask_workspace_bytes(mm.masks, mm.m, n, mm.width);
  Tensor workspace =
      at::empty({static_cast<int64_t>(ws_bytes)}, values.options().dtype(at::kByte));
  check_status(sputnik_hip_csr_transpose_many_mask(
                   mm.masks, mm.m, n, mm.nonzeros.data(), mm.replicas, values.data_ptr<float>(),
                   values.size(1), mm.row_offsets.data_ptr<int>(),
                   mm.column_indices.data_ptr<int>(), out_values.data_ptr<float>(),
                   out_values.size(1), out_row_offsets.data_ptr<int>(),
                   out_column_indices.data_ptr<int>(), nullptr, workspace.data_ptr(), ws_bytes,
                   current_stream(values)),
               "csr_transpose_many_mask");
  return {out_values, out_row_offsets, out_column_indices};
}

// Layout pass of the reference's modules (modules/sparse_linear.py:89,
// modules/sparse_attention.py:108-126): x[..., R, C] -> contiguous [..., C, R],
// i.e. `x.transpose(-1, -2).contiguous()` as ONE tiled kernel, optionally
// changing the storage type on the way (out_type: -1 keep, 0 float32, 1 float16,
// 2 bfloat16; half <-> float and equal types).
Tensor transpose_last2_as(const Tensor& x_in, int64_t out_type) {
  TORCH_CHECK(x_in.is_cuda(), "transpose_last2: expected a GPU (HIP) tensor, got ", x_in.device());
  TORCH_CHECK(x_in.dim() >= 2, "transpose_last2: expected at least 2 dimensions, got ", x_in.dim());
  const int in_code = type_code(x_in.scalar_type());
  const int out_code = out_type < 0 ? in_code : static_cast<int>(out_type);
  const at::ScalarType out_dtype = out_code == SPUTNIK_HIP_F32 ? at::kFloat
                                   : out_code == SPUTNIK_HIP_F16 ? at::kHalf : at::kBFloat16;
  const bool native = in_code >= 0 && out_code >= 0 && out_code <= 2 &&
                      (in_code == out_code || in_code == SPUTNIK_HIP_F32 || out_code == SPUTNIK_HIP_F32);
  if (!native) {   // (other element types: ATen's strided copy)
    Tensor t = x_in.transpose(-1, -2).contiguous();
    return out_type < 0 ? t : t.to(out_dtype);
  }
  const Tensor x = x_in.contiguous();
  const c10::DeviceGuard guard(x.device());
  std::vector<int64_t> sizes = x.sizes().vec();
  const int rows = to_int(sizes[sizes.size() - 2], "rows"), cols = to_int(sizes.back(), "cols");
  std::swap(sizes[sizes.size() - 2], sizes[sizes.size() - 1]);
  Tensor out = at::empty(sizes, x.options().dtype(out_dtype));
  const int64_t per = static_cast<int64_t>(rows) * cols;
  const int batches = per == 0 ? 0 : to_int(x.numel() / per, "batches");
  check_status(sputnik_hip_transpose_cast_batched(batches, rows, cols, x.data_ptr(), in_code, per,
                                                  out.data_ptr(), out_code, per,
                                                  current_stream(x)),
               "transpose_last2");
  return out;
}

Tensor transpose_last2(const Tensor& x) { return transpose_last2_as(x, -1); }

// values[..., permutation] for value arrays [nnz] / [R, nnz] that share one int32
// permutation (the transposed order of a static pattern): `index_select(-1, perm)`
// as one kernel that reads the permutation once for all rows.
Tensor permute_last(const Tensor& values_in, const Tensor& permutation) {
  const Tensor values = as_float(values_in, "values");
  TORCH_CHECK(values.dim() == 1 || values.dim() == 2, "values should have 1 or 2 dimensions, got ",
              values.dim());
  TORCH_CHECK(permutation.scalar_type() == at::kInt && permutation.dim() == 1 &&
                  permutation.is_contiguous() && permutation.device() == values.device(),
              "permutation must be a contiguous int32 vector on ", values.device());
  TORCH_CHECK(permutation.size(0) == values.size(-1), "permutation has ", permutation.size(0),
              " entries, values ", values.size(-1));
  const c10::DeviceGuard guard(values.device());
  const int n = to_int(values.size(-1), "n");
  const int rows = values.dim() == 2 ? to_int(values.size(0), "rows") : 1;
  Tensor out = at::empty_like(values);
  check_status(sputnik_hip_permute_last_batched(n, rows, values.data_ptr<float>(), n,
                                                permutation.data_ptr<int>(),
                                                out.data_ptr<float>(), n, current_stream(values)),
               "permute_last");
  return out;
}

// permute_last through LDS for many rows; the lists come from
// torch_sputnik_amd.functional (one stable sort per cached permutation)
Tensor permute_last_banded(const Tensor& values_in, const Tensor& dest_list,
                           const Tensor& source_in_band) {
  const Tensor values = as_float(values_in, "values");
  TORCH_CHECK(values.dim() == 1 || values.dim() == 2, "values should have 1 or 2 dimensions, got ",
              values.dim());
  for (const Tensor* t : {&dest_list, &source_in_band})
    TORCH_CHECK(t->scalar_type() == at::kInt && t->dim() == 1 && t->is_contiguous() &&
                    t->device() == values.device() && t->size(0) == values.size(-1),
                "dest_list / source_in_band must be contiguous int32 vectors of ",
                values.size(-1), " entries on ", values.device());
  const c10::DeviceGuard guard(values.device());
  const int n = to_int(values.size(-1), "n");
  const int rows = values.dim() == 2 ? to_int(values.size(0), "rows") : 1;
  Tensor out = at::empty_like(values);
  check_status(sputnik_hip_permute_banded_batched(
                   n, rows, values.data_ptr<float>(), n, dest_list.data_ptr<int>(),
                   source_in_band.data_ptr<int>(), out.data_ptr<float>(), n,
                   current_stream(values)),
               "permute_last_banded");
  return out;
}

int64_t permute_band_size() { return sputnik_hip_permute_band_size(); }

}  // namespace

TORCH_LIBRARY(torch_sputnik, m) {
  m.def(
      "spmm(int m, int k, Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, Tensor dense_matrix) -> Tensor");
  m.def(
      "left_spmm(int m, int k, Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, Tensor dense_matrix) -> Tensor");
  m.def(
      "left_spmm_half_tiles(int m, int k, Tensor values, Tensor row_offsets, Tensor column_indices, "
      "Tensor dense_matrix, int tile_type) -> Tensor");
  m.def("half_linear_image(int out_features, int in_features, Tensor values, Tensor row_offsets, "
        "Tensor column_indices, int tile_type) -> Tensor");
  m.def("half_planes(Tensor t, int tile_type) -> Tensor");
  m.def("half_linear_forward(int out_features, Tensor image, int values_type, Tensor x) -> Tensor");
  m.def("half_linear_plan(int out_features, int in_features, Tensor row_offsets, Tensor column_indices) "
        "-> Tensor");
  m.def("half_linear_weight_gradient(int out_features, Tensor row_offsets, Tensor column_indices, "
        "Tensor grad, bool grad_is_planes, Tensor x, Tensor? plan) -> Tensor");
  m.def("half_linear_input_gradient(int out_features, int in_features, Tensor grad, bool grad_is_planes, "
        "Tensor image, int values_type, Tensor like, int batch, int seq) -> Tensor");
  m.def(
      "sddmm(int m, int n, Tensor row_indices, Tensor row_offsets, Tensor column_indices, "
      "Tensor lhs_matrix, Tensor rhs_matrix) -> Tensor");
  m.def(
      "sddmm_narrow(int m, int n, Tensor row_indices, Tensor row_offsets, Tensor column_indices, "
      "Tensor lhs_matrix, Tensor rhs_matrix) -> Tensor");
  m.def(
      "sparse_softmax(Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices) -> Tensor");
  m.def(
      "csr_transpose(int m, int n, Tensor values, Tensor row_offsets, Tensor column_indices) "
      "-> Tensor[]");
  m.def(
      "csr_transpose_with_permutation(int m, int n, Tensor values, Tensor row_offsets, "
      "Tensor column_indices, bool checked=True) -> Tensor[]");
  // extensions
  m.def(
      "spmm_bias(int m, int k, Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, Tensor bias, Tensor dense_matrix) -> Tensor");
  m.def(
      "spmm_bias_relu(int m, int k, Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, Tensor bias, Tensor dense_matrix) -> Tensor");
  m.def(
      "sparse_softmax_scaled(Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, float scale) -> Tensor");
  m.def(
      "sparse_softmax_backward(Tensor softmax_out, Tensor grad_out, Tensor row_offsets, "
      "float scale) -> Tensor");
  m.def(
      "sparse_attention(Tensor query, Tensor key, Tensor value, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, float scale) -> Tensor");
  m.def(
      "sparse_attention_with_lse(Tensor query, Tensor key, Tensor value, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, float scale) -> Tensor[]");
  m.def(
      "spmm_plan(int m, int k, int n, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices) -> Tensor");
  m.def(
      "spmm_planned(int m, int k, Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, Tensor dense_matrix, Tensor plan) -> Tensor");
  m.def(
      "left_spmm_planned(int m, int k, Tensor values, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, Tensor dense_matrix, Tensor plan) -> Tensor");
  m.def(
      "sddmm_plan(int m, int n, int k, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices) -> Tensor");
  m.def(
      "sddmm_planned(int m, int n, Tensor row_indices, Tensor row_offsets, Tensor column_indices, "
      "Tensor lhs_matrix, Tensor rhs_matrix, Tensor plan) -> Tensor");
  m.def(
      "sddmm_sum_plan(int m, int n, int k, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices) -> Tensor");
  m.def(
      "sddmm_sum(int m, int n, Tensor row_indices, Tensor row_offsets, Tensor column_indices, "
      "Tensor lhs_matrix, Tensor rhs_matrix) -> Tensor");
  m.def(
      "sddmm_sum_planned(int m, int n, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices, Tensor lhs_matrix, Tensor rhs_matrix, Tensor plan) -> Tensor");
  m.def(
      "sddmm_sum_group_planned(int m, int n, Tensor[] row_indices, Tensor[] row_offsets, "
      "Tensor[] column_indices, Tensor[] lhs_matrices, Tensor rhs_matrix, Tensor[] plans) -> Tensor[]");
  m.def(
      "sparse_attention_plan(int m, int n, int d, Tensor row_indices, Tensor row_offsets, "
      "Tensor column_indices) -> Tensor");
  m.def(
      "sparse_attention_planned(Tensor query, Tensor key, Tensor value, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, float scale, Tensor plan) -> Tensor");
  m.def(
      "spmm_many_mask(int b, int m, int k, Tensor nonzeros, Tensor values, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, Tensor dense_matrix) -> Tensor");
  m.def(
      "sddmm_many_mask(int b, int m, int n, Tensor nonzeros, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, Tensor lhs_matrix, Tensor rhs_matrix) -> Tensor");
  m.def(
      "sparse_softmax_many_mask(int b, int m, Tensor nonzeros, Tensor values, "
      "Tensor row_indices, Tensor row_offsets, Tensor column_indices) -> Tensor");
  m.def(
      "sparse_softmax_many_mask_scaled(int b, int m, Tensor nonzeros, Tensor values, "
      "Tensor row_indices, Tensor row_offsets, Tensor column_indices, float scale) -> Tensor");
  m.def(
      "sparse_softmax_backward_many_mask(int b, int m, Tensor nonzeros, Tensor softmax_out, "
      "Tensor grad_out, Tensor row_offsets, float scale) -> Tensor");
  m.def(
      "csr_transpose_many_mask(int b, int m, int n, Tensor nonzeros, Tensor values, "
      "Tensor row_offsets, Tensor column_indices) -> Tensor[]");
  m.def("permute_last(Tensor values, Tensor permutation) -> Tensor");
  m.def("permute_last_banded(Tensor values, Tensor dest_list, Tensor source_in_band) -> Tensor");
  m.def("permute_band_size() -> int", &permute_band_size);
  m.def(
      "spmm_permuted(int m, int k, Tensor values, Tensor permutation, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, Tensor dense_matrix, Tensor? plan) -> Tensor");
  m.def(
      "spmm_transposed_out(int m, int k, Tensor values, Tensor? permutation, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, Tensor dense_matrix, int block_rows, bool left, "
      "Tensor? plan) -> Tensor");
  m.def(
      "left_spmm_group(int m, int k, Tensor[] values, Tensor[] row_indices, Tensor[] row_offsets, "
      "Tensor[] column_indices, Tensor dense_matrix, int block_rows) -> Tensor[]");
  m.def(
      "left_spmm_group_sum(int m, int k, Tensor[] values, Tensor[] permutations, "
      "Tensor[] row_indices, Tensor[] row_offsets, Tensor[] column_indices, "
      "Tensor[] dense_matrices) -> Tensor");
  m.def(
      "left_spmm_permuted(int m, int k, Tensor values, Tensor permutation, Tensor row_indices, "
      "Tensor row_offsets, Tensor column_indices, Tensor dense_matrix, Tensor? plan) -> Tensor");
  m.def("transpose_last2(Tensor x) -> Tensor");
  m.def("transpose_last2_as(Tensor x, int out_type) -> Tensor");
}

// "CUDA" is the dispatch key of HIP tensors in a ROCm build of PyTorch.
TORCH_LIBRARY_IMPL(torch_sputnik, CUDA, m) {
  m.impl("spmm", &spmm);
  m.impl("left_spmm", &left_spmm);
  m.impl("sddmm", &sddmm);
  m.impl("sddmm_narrow", &sddmm_narrow);
  m.impl("sparse_softmax", &sparse_softmax);
  m.impl("csr_transpose", &csr_transpose);
  m.impl("csr_transpose_with_permutation", &csr_transpose_with_permutation);
  m.impl("spmm_bias", &spmm_bias);
  m.impl("spmm_bias_relu", &spmm_bias_relu);
  m.impl("sparse_softmax_scaled", &sparse_softmax_scaled);
  m.impl("sparse_softmax_backward", &sparse_softmax_backward);
  m.impl("sparse_attention", &sparse_attention);
  m.impl("sparse_attention_with_lse", &sparse_attention_with_lse);
  m.impl("spmm_plan", &spmm_plan);
  m.impl("spmm_planned", &spmm_planned);
  m.impl("left_spmm_planned", &left_spmm_planned);
  m.impl("sddmm_plan", &sddmm_plan);
  m.impl("sddmm_planned", &sddmm_planned);
  m.impl("left_spmm_half_tiles", &left_spmm_half_tiles);
  m.impl("half_linear_image", &half_linear_image);
  m.impl("half_planes", &half_planes);
  m.impl("half_linear_forward", &half_linear_forward);
  m.impl("half_linear_plan", &half_linear_plan);
  m.impl("half_linear_weight_gradient", &half_linear_weight_gradient);
  m.impl("half_linear_input_gradient", &half_linear_input_gradient);
  m.impl("sddmm_sum", &sddmm_sum);
  m.impl("sddmm_sum_plan", &sddmm_sum_plan);
  m.impl("sddmm_sum_planned", &sddmm_sum_planned);
  m.impl("sddmm_sum_group_planned", &sddmm_sum_group_planned);
  m.impl("sparse_attention_plan", &sparse_attention_plan);
  m.impl("sparse_attention_planned", &sparse_attention_planned);
  m.impl("spmm_many_mask", &spmm_many_mask);
  m.impl("sddmm_many_mask", &sddmm_many_mask);
  m.impl("sparse_softmax_many_mask", &sparse_softmax_many_mask);
  m.impl("sparse_softmax_many_mask_scaled", &sparse_softmax_many_mask_scaled);
  m.impl("sparse_softmax_backward_many_mask", &sparse_softmax_backward_many_mask);
  m.impl("csr_transpose_many_mask", &csr_transpose_many_mask);
  m.impl("permute_last", &permute_last);
  m.impl("permute_last_banded", &permute_last_banded);
  m.impl("spmm_permuted", &spmm_permuted);
  m.impl("spmm_transposed_out", &spmm_transposed_out);
  m.impl("left_spmm_group", &left_spmm_group);
  m.impl("left_spmm_group_sum", &left_spmm_group_sum);
  m.impl("left_spmm_permuted", &left_spmm_permuted);
  m.impl("transpose_last2", &transpose_last2);
  m.impl("transpose_last2_as", &transpose_last2_as);
}
