// Pieces shared by the quad-form SDDMM kernels (sddmm_tiled.hip: rhs slab in LDS, lhs
// row in registers; sddmm_flat.hip: both operands in LDS): the 16-byte chunk product in
// the operands' storage type, and the quad broadcast that rides on an address add.
#pragma once

#include "wave_utils.h"

namespace sputnik_hip {

using v2f = float __attribute__((ext_vector_type(2)));

template <typename T> struct Dot;
template <> struct Dot<float> {
  using chunk = float __attribute__((ext_vector_type(4)));   // 16 bytes of a row
  static __device__ __forceinline__ void mac(v2f& acc, const chunk& a, const chunk& b) {
    acc = __builtin_elementwise_fma(v2f{a.x, a.y}, v2f{b.x, b.y}, acc);
    acc = __builtin_elementwise_fma(v2f{a.z, a.w}, v2f{b.z, b.w}, acc);
  }
};
using h2v = _Float16 __attribute__((ext_vector_type(2)));
using b2v = __bf16 __attribute__((ext_vector_type(2)));
using HalfChunk = unsigned __attribute__((ext_vector_type(4)));   // 8 half values
// (the words are copied out first: __builtin_bit_cast applied to `a[1]` directly reads
// element 0 of the vector with this compiler)
__device__ __forceinline__ float dot2_f16(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2v, a), __builtin_bit_cast(h2v, b), c, false);
}
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2v, a), __builtin_bit_cast(b2v, b), c, false);
}
template <> struct Dot<_Float16> {
  using chunk = HalfChunk;
  static __device__ __forceinline__ void mac(v2f& acc, const chunk& a, const chunk& b) {
    const unsigned a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
    acc.x = dot2_f16(a0, b0, acc.x);
    acc.y = dot2_f16(a1, b1, acc.y);
    acc.x = dot2_f16(a2, b2, acc.x);
    acc.y = dot2_f16(a3, b3, acc.y);
  }
};
template <> struct Dot<__bf16> {
  using chunk = HalfChunk;
  static __device__ __forceinline__ void mac(v2f& acc, const chunk& a, const chunk& b) {
    const unsigned a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
    acc.x = dot2_bf16(a0, b0, acc.x);
    acc.y = dot2_bf16(a1, b1, acc.y);
    acc.x = dot2_bf16(a2, b2, acc.x);
    acc.y = dot2_bf16(a3, b3, acc.y);
  }
};

template <int S>
__device__ __forceinline__ int quad_bcast_add(int v, int add) {
  // add + (v of lane S of the quad): v_add_u32_dpp quad_perm:[S,S,S,S]
  return __builtin_amdgcn_update_dpp(0, v, S * 0x55, 0xF, 0xF, true) + add;
}

}  // namespace sputnik_hip
