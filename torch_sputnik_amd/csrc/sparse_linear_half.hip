// A sparse layer on HALF-stored activations, all three products on the matrix cores
// (round 5).  The reference runs SparseLinear (modules/sparse_linear.py:18-89) through the
// float32 operators with two layout passes around them (x.transpose(1, 2).contiguous(),
// :89); on half storage -- this library's extension -- the tile kernel of mfma_gemm.h
// reads every operand IN THE LAYOUT THE CALLER HAS IT, so no transposed copy is made at all:
//
//   forward          y[b][o][s]  = sum_i W[o][i] x[b][s][i]       W image k-contiguous, x k-contiguous
//   weight gradient  dW[o][i]    = sum_b sum_s dy[b][o][s] x[b][s][i]   dy k-contiguous, x k-major, sampled at W's mask
//   input gradient   dx[b][s][i] = sum_o dy[b][o][s] W[o][i]      dy k-major, W image k-major, stored in x's type
//
// with x [batch, seq, in] as the module receives it, dy [batch, out, seq] as autograd hands
// it over, and ONE densified image of the weight [out, in] (sputnik_hip_sparse_linear_half_image:
// one launch: every row assembled in LDS and written whole) shared by the forward pass and the input
// gradient.  float32 values and the float32 dy are not rounded to the storage type: they
// enter as half planes whose sum is the value (mfma_gemm.h; dy is split once per backward
// pass, sputnik_hip_half_planes, and both gradients read the planes).
#include "mfma.h"
#include "mfma_gemm.h"
#include "options.h"

namespace sputnik_hip {

using namespace mfma_tiles;

namespace {

bool half_type(int t) { return t == SPUTNIK_HIP_F16 || t == SPUTNIK_HIP_BF16; }

int planes_of(int operand_type, int tile_type) {
  return operand_type == SPUTNIK_HIP_F32 ? sddmm_mfma_planes_of(tile_type) : 1;
}

// The three products take the route together or not at all (they share the image and the
// planes): every reduction a multiple of 64, 16-byte rows, a grid that fills the chip, and
// a density at which the dense tiles beat the vector kernels' 40-50 sampled TFLOP/s.
bool served(int out_f, int in_f, int seq, int batch, int nonzeros, int values_type, int tile_type) {
  if (!half_type(tile_type)) return false;
  if (values_type != SPUTNIK_HIP_F32 && values_type != tile_type) return false;
  if (out_f <= 0 || in_f <= 0 || seq <= 0 || batch <= 0 || nonzeros <= 0) return false;
  if (out_f % kStep != 0 || in_f % kStep != 0 || seq % kStep != 0) return false;
  if (static_cast<int64_t>(out_f) * in_f >= (int64_t{1} << 30) ||
      static_cast<int64_t>(out_f) * seq >= (int64_t{1} << 30) ||
      static_cast<int64_t>(in_f) * seq >= (int64_t{1} << 30))
    return false;
  const int forced = options().spmm_kernel;
  if (forced != 0 && forced != 4) return false;
  if (forced == 4) return true;
  const int64_t tiles = static_cast<int64_t>(ceil_div(out_f, kTile)) * ceil_div(seq, kTile) * batch;
  const double density = static_cast<double>(nonzeros) / (static_cast<double>(out_f) * in_f);
  const int pw = planes_of(values_type, tile_type);
  return out_f >= kTile && in_f >= kTile && seq >= kTile && tiles >= 192 && density >= 0.06 * pw;
}

}  // namespace

// (spmm_mfma.hip)
int densify_into(int m, int k, const int* row_offsets, const int* column_indices, const void* values,
                 int values_type, int tile_type, void* image, int64_t rows_padded, hipStream_t stream);

}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

int sputnik_hip_sparse_linear_half_supported(int out_features, int in_features, int seq, int batch,
                                             int nonzeros, int values_type, int tile_type) {
  return served(out_features, in_features, seq, batch, nonzeros, values_type, tile_type) ? 1 : 0;
}

size_t sputnik_hip_sparse_linear_half_image_bytes(int out_features, int in_features, int values_type,
                                                  int tile_type) {
  if (out_features <= 0 || in_features <= 0 || !half_type(tile_type)) return 0;
  return static_cast<size_t>(planes_of(values_type, tile_type)) * out_features * in_features * 2;
}

int sputnik_hip_sparse_linear_half_image(int out_features, int in_features, int nonzeros,
                                         const int* row_offsets, const int* column_indices,
                                         const void* values, int values_type, int tile_type,
                                         void* image, size_t image_bytes, sputnik_hip_stream_t stream) {
  if (out_features <= 0 || in_features <= 0 || nonzeros < 0 || !half_type(tile_type) ||
      (values_type != SPUTNIK_HIP_F32 && values_type != tile_type))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (image == nullptr || !aligned_to(image, 16) ||
      image_bytes < sputnik_hip_sparse_linear_half_image_bytes(out_features, in_features, values_type, tile_type))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  return densify_into(out_features, in_features, row_offsets, column_indices, values, values_type,
                      tile_type, image, out_features, stream);
}

size_t sputnik_hip_half_planes_bytes(int64_t count, int tile_type) {
  return half_type(tile_type) && count > 0 ? static_cast<size_t>(sddmm_mfma_planes_of(tile_type)) * count * 2 : 0;
}

int sputnik_hip_half_planes(int64_t count, const float* in, int tile_type, void* planes,
                            sputnik_hip_stream_t stream) {
  return sddmm_mfma_split_planes(count, in, tile_type, planes, stream);
}

int sputnik_hip_sparse_linear_half_forward(int out_features, int in_features, int seq, int batch,
                                           const void* image, int values_type, const void* x,
                                           int tile_type, const float* bias, int relu, float* y,
                                           sputnik_hip_stream_t stream) {
  if (!half_type(tile_type) || out_features <= 0 || in_features % kStep != 0 || seq <= 0 || batch <= 0)
    return SPUTNIK_HIP_UNSUPPORTED;
  if (!aligned_to(image, 16) || !aligned_to(x, 16) || !aligned_to(y, 4)) return SPUTNIK_HIP_UNSUPPORTED;
  const int pw = planes_of(values_type, tile_type);
  const GemmOperand a{image, in_features, 0, static_cast<int64_t>(out_features) * in_features};
  const GemmOperand b{x, in_features, static_cast<int64_t>(seq) * in_features, 0};
  GemmOut o{};
  o.dense = y;
  o.ld = seq;
  o.outer_stride = static_cast<int64_t>(out_features) * seq;
  o.bias = bias;
  o.relu = relu;
  return launch_mfma_gemm_typed<false, false, kDense>(tile_type, pw, 1, out_features, seq, in_features,
                                                      batch, batch, false, a, b, o, stream,
                                                      wide_tile(out_features, seq, batch));
}

size_t sputnik_hip_sparse_linear_half_plan_bytes(int out_features, int in_features) {
  return out_features > 0 && in_features > 0 ? sddmm_mfma_plan_bytes(out_features, in_features) : 0;
}

int sputnik_hip_sparse_linear_half_plan(int out_features, int in_features, const int* row_offsets,
                                        const int* column_indices, void* plan,
                                        sputnik_hip_stream_t stream) {
  if (plan == nullptr || !aligned_to(plan, 16)) return SPUTNIK_HIP_INVALID_ARGUMENT;
  return sddmm_mfma_plan(out_features, in_features, row_offsets, column_indices, plan, stream);
}

size_t sputnik_hip_sparse_linear_half_scratch_bytes(int out_features, int in_features, int seq, int batch,
                                                    int nonzeros, int grad_type, int tile_type) {
  if (!half_type(tile_type) || nonzeros <= 0) return 0;
  const int splits = sddmm_mfma_splits(out_features, seq, in_features, batch, planes_of(grad_type, tile_type));
  return splits > 1 ? sizeof(float) * static_cast<size_t>(splits) * nonzeros : 0;
}

// grad_type SPUTNIK_HIP_F32: `grad_output` are the PLANES of dy (sputnik_hip_half_planes on the
// whole [batch, out, seq] tensor); tile_type: dy itself, stored in the tile type.
int sputnik_hip_sparse_linear_half_weight_gradient(int out_features, int in_features, int seq, int batch,
                                                   int nonzeros, const int* row_offsets,
                                                   const int* column_indices, const void* grad_output,
                                                   int grad_type, const void* x, int tile_type,
                                                   float* grad_values, const void* plan, void* scratch,
                                                   size_t scratch_bytes, sputnik_hip_stream_t stream) {
  if (!half_type(tile_type) || seq % kStep != 0 || in_features % 8 != 0 || out_features <= 0 ||
      batch <= 0 || nonzeros <= 0)
    return SPUTNIK_HIP_UNSUPPORTED;
  if (grad_type != SPUTNIK_HIP_F32 && grad_type != tile_type) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (!aligned_to(grad_output, 16) || !aligned_to(x, 16)) return SPUTNIK_HIP_UNSUPPORTED;
  const int pg = planes_of(grad_type, tile_type);
  int splits = sddmm_mfma_splits(out_features, seq, in_features, batch, pg);
  if (splits > 1 && (scratch == nullptr || !aligned_to(scratch, 16) ||
                     scratch_bytes < sizeof(float) * static_cast<size_t>(splits) * nonzeros))
    splits = 1;
  const int64_t per_replica = static_cast<int64_t>(out_features) * seq;
  const GemmOperand a{grad_output, seq, per_replica, per_replica * batch};
  const GemmOperand b{x, in_features, static_cast<int64_t>(seq) * in_features, 0};   // k = s major
  GemmOut o{};
  o.sampled = splits == 1 ? grad_values : static_cast<float*>(scratch);
  o.row_offsets = row_offsets;
  o.column_indices = column_indices;
  o.plan = static_cast<const int*>(plan);
  o.nonzeros = nonzeros;
  o.vector_columns = aligned_to(column_indices, 16) ? 1 : 0;
  const int st = launch_mfma_gemm_typed<false, true, kSampled>(tile_type, pg, 1, out_features, in_features,
                                                               seq, batch, splits, true, a, b, o, stream,
                                                               wide_tile(out_features, in_features, 0));
  if (st != 0 || splits == 1) return st;
  return sum_partial_vectors(nonzeros, splits, static_cast<const float*>(scratch), grad_values, stream);
}

// grad_input stored as grad_input_type (SPUTNIK_HIP_F32 or tile_type), [batch, seq, in].
int sputnik_hip_sparse_linear_half_input_gradient(int out_features, int in_features, int seq, int batch,
                                                  const void* grad_output, int grad_type,
                                                  const void* image, int values_type, int tile_type,
                                                  void* grad_input, int grad_input_type,
                                                  sputnik_hip_stream_t stream) {
  if (!half_type(tile_type) || out_features % kStep != 0 || seq % 8 != 0 || in_features % 8 != 0 ||
      batch <= 0)
    return SPUTNIK_HIP_UNSUPPORTED;
  if (grad_type != SPUTNIK_HIP_F32 && grad_type != tile_type) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (grad_input_type != SPUTNIK_HIP_F32 && grad_input_type != tile_type) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (!aligned_to(grad_output, 16) || !aligned_to(image, 16) || !aligned_to(grad_input, 4))
    return SPUTNIK_HIP_UNSUPPORTED;
  const int pg = planes_of(grad_type, tile_type), pw = planes_of(values_type, tile_type);
  const int64_t per_replica = static_cast<int64_t>(out_features) * seq;
  const GemmOperand a{grad_output, seq, per_replica, per_replica * batch};             // k = o major
  const GemmOperand b{image, in_features, 0, static_cast<int64_t>(out_features) * in_features};   // k = o major
  GemmOut o{};
  o.dense = grad_input;
  o.ld = in_features;
  o.outer_stride = static_cast<int64_t>(seq) * in_features;
  if (grad_input_type == SPUTNIK_HIP_F32)
    return launch_mfma_gemm_typed<true, true, kDense>(tile_type, pg, pw, seq, in_features, out_features,
                                                      batch, batch, false, a, b, o, stream,
                                                      wide_tile(seq, in_features, batch));
  return launch_mfma_gemm_typed<true, true, kDenseHalf>(tile_type, pg, pw, seq, in_features, out_features,
                                                        batch, batch, false, a, b, o, stream,
                                                        wide_tile(seq, in_features, batch));
}

}  // extern "C"
