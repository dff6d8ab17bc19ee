// Shared host/device helpers for libsputnik_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sputnik_hip.h"

namespace sputnik_hip {

// Largest gridDim.y / gridDim.z HIP accepts.
constexpr int kMaxGridYZ = 65535;

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline bool aligned_to(const void* p, size_t bytes) {
  return (reinterpret_cast<uintptr_t>(p) % bytes) == 0;
}

// Widest float vector (4, 2 or 1) that every row start of a row-major
// [rows, width] operand with base `p` and replica stride `stride` supports.
inline int vector_width(const void* p, int64_t width, int64_t stride) {
  if (width % 4 == 0 && stride % 4 == 0 && aligned_to(p, 16)) return 4;
  if (width % 2 == 0 && stride % 2 == 0 && aligned_to(p, 8)) return 2;
  return 1;
}

inline int launch_status() { return static_cast<int>(hipGetLastError()); }

// Fused SpMM epilogue: out[i, :] = act(acc + bias[i]) (both optional).
struct Epilogue {
  const float* bias = nullptr;  // [m], indexed by output row
  int relu = 0;
};

__device__ __forceinline__ float epilogue_scalar(float v, float b, int relu) {
  v += b;
  return relu ? fmaxf(v, 0.f) : v;
}

__device__ __forceinline__ float4 apply_epilogue(float4 v, const Epilogue& e, int row) {
  if (e.bias != nullptr || e.relu) {
    const float b = e.bias != nullptr ? e.bias[row] : 0.f;
    v.x = epilogue_scalar(v.x, b, e.relu);
    v.y = epilogue_scalar(v.y, b, e.relu);
    v.z = epilogue_scalar(v.z, b, e.relu);
    v.w = epilogue_scalar(v.w, b, e.relu);
  }
  return v;
}

// "Many mask" batches (tests/transformer/functions.py of the reference: one mask
// per batch element, shared by its heads): the topologies are concatenated --
// row_offsets [masks][m + 1] (each zero based), row_indices [masks][m],
// column_indices back to back -- and replica r works under topology r / heads.
// A kernel that serves such a batch in ONE launch moves its topology pointers
// by what select_mask returns (wave-uniform: scalar loads); heads = 0 means one
// topology for all replicas (nothing moves).
struct MaskPlace {
  int mask;      // topology number of the replica
  int first;     // its first entry in the concatenated column_indices
  int nonzeros;  // its entry count
};
__device__ __forceinline__ MaskPlace select_mask(int heads, int replica, int m, int nonzeros,
                                                 const int* row_offsets) {
  if (heads <= 0) return MaskPlace{0, 0, nonzeros};
  const int mask = replica / heads;
  int first = 0;
  for (int j = 0; j < mask; ++j) first += row_offsets[static_cast<int64_t>(j) * (m + 1) + m];
  const int count = row_offsets[static_cast<int64_t>(mask) * (m + 1) + m];
  // An EMPTY mask behind all the others would start one past the concatenated
  // column_indices, and the kernels request the window at their (clamped) entry 0
  // before they know that no row has any: it starts on the previous entry instead
  // (never used -- every row is empty).
  if (count == 0 && first > 0) --first;
  return MaskPlace{mask, first, count};
}
// (the kernels then do:  row_offsets += place.mask * (m + 1); column_indices += place.first;
//  row_indices += place.mask * m; nonzeros = place.nonzeros)

template <int VEC>
struct FloatVec;
template <>
struct FloatVec<1> {
  using type = float;
};
template <>
struct FloatVec<2> {
  using type = float2;
};
template <>
struct FloatVec<4> {
  using type = float4;
};

template <int VEC>
__device__ __forceinline__ void load_vec(float (&dst)[VEC], const float* __restrict__ src) {
  using V = typename FloatVec<VEC>::type;
  const V v = *reinterpret_cast<const V*>(src);
  if constexpr (VEC == 1) {
    dst[0] = v;
  } else if constexpr (VEC == 2) {
    dst[0] = v.x;
    dst[1] = v.y;
  } else {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
    dst[3] = v.w;
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ dst, const float (&src)[VEC]) {
  using V = typename FloatVec<VEC>::type;
  if constexpr (VEC == 1) {
    *dst = src[0];
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<V*>(dst) = make_float2(src[0], src[1]);
  } else {
    *reinterpret_cast<V*>(dst) = make_float4(src[0], src[1], src[2], src[3]);
  }
}

}  // namespace sputnik_hip
