// Shared pieces of the matrix-core kernels (sddmm_mfma.hip, spmm_mfma.hip): the
// 32 x 32 x 16 MFMA on float16 / bfloat16 fragments, tile geometry, the direct
// global->LDS copy.
#pragma once

#include "common.h"
#include "spmm_tiled_common.h"

namespace sputnik_hip {
namespace mfma_tiles {

using tiled::xcd_local_index32;
using f32x16 = float __attribute__((ext_vector_type(16)));

template <typename T>
struct Half8;
template <>
struct Half8<_Float16> {
  using type = _Float16 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(type a, type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Half8<__bf16> {
  using type = __bf16 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(type a, type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

constexpr int kTile = 128;                       // rows and columns of an output tile
constexpr int kStep = 64;                        // k elements per step (128 bytes per row)
constexpr int kOperandBytes = kTile * kStep * 2; // 16 KiB: one operand's tile of a step
constexpr int kStageBytes = 2 * kOperandBytes;   // lhs tile, rhs tile
constexpr int kPitch = kTile + 4;                // floats per row of the epilogue's tile
constexpr int kTileBytes = kTile * kPitch * 4;   // 67 584
constexpr int kLdsBytes = kTileBytes + 4 * (kTile + 4);   // + the tile rows' CSR bounds
static_assert(kTileBytes >= 2 * kStageBytes, "the float tile reuses the stages");

constexpr float kLowPlaneScale = 2048.f;   // 2^11: the float16 split's low plane (see split_planes_kernel)

// ints of the plan's row_ok part (the table sits behind it)
__host__ __device__ inline int64_t plan_rows(int m) { return (static_cast<int64_t>(m) + 3) / 4 * 4; }

// A wave-uniform pointer that the compiler computed on the vector unit (an integer
// division ends there), back in scalar registers: the copies below take their base there.
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return reinterpret_cast<const T*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// direct global->LDS copy of 64 x 16 bytes: LDS destination M0 + lane * 16, per-lane source
__device__ __forceinline__ void copy_piece(const void* base /* wave-uniform */,
                                           unsigned lane_byte_offset, const char* lds_dst) {
  const unsigned lds_addr =
      static_cast<unsigned>(reinterpret_cast<uintptr_t>(AS_LDS(const_cast<char*>(lds_dst))));
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2"
      :
      : "s"(lds_addr), "v"(lane_byte_offset), "s"(base)
      : "memory", "m0");
}


}  // namespace mfma_tiles
}  // namespace sputnik_hip
