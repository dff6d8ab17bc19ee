// Wave64 cross-lane helpers for gfx950 (CDNA4).  Everything here assumes a
// 64-lane wavefront; sub-groups are aligned power-of-two slices of a wave.
#pragma once

#include <hip/hip_runtime.h>

namespace sputnik_hip {

constexpr int kWave = 64;

// DPP controls (gfx9 encoding).
constexpr int kDppQuadXor1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ float readlane_f32(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

struct SumOp {
  __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct MaxOp {
  __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};

// All-reduce inside aligned groups of WIDTH lanes (WIDTH in {1,2,4,8,16,32,64}).
// Up to 16 lanes this is pure DPP (one VALU op per step, no LDS crossbar);
// the 32/64 steps go through v_readlane broadcasts of the four row totals.
template <int WIDTH, typename Op>
__device__ __forceinline__ float group_allreduce(float v, Op op) {
  static_assert(WIDTH >= 1 && WIDTH <= 64 && (WIDTH & (WIDTH - 1)) == 0, "bad width");
  if constexpr (WIDTH >= 2) v = op(v, dpp_f32<kDppQuadXor1>(v));
  if constexpr (WIDTH >= 4) v = op(v, dpp_f32<kDppQuadXor2>(v));
  if constexpr (WIDTH >= 8) v = op(v, dpp_f32<kDppRowHalfMirror>(v));
  if constexpr (WIDTH >= 16) v = op(v, dpp_f32<kDppRowMirror>(v));
  if constexpr (WIDTH == 32) {
    // lanes 0-31: rows 0,1; lanes 32-63: rows 2,3
    const float r0 = readlane_f32(v, 0), r1 = readlane_f32(v, 16);
    const float r2 = readlane_f32(v, 32), r3 = readlane_f32(v, 48);
    v = (__lane_id() < 32) ? op(r0, r1) : op(r2, r3);
  }
  if constexpr (WIDTH == 64) {
    const float r0 = readlane_f32(v, 0), r1 = readlane_f32(v, 16);
    const float r2 = readlane_f32(v, 32), r3 = readlane_f32(v, 48);
    v = op(op(r0, r1), op(r2, r3));
  }
  return v;
}

template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
  return group_allreduce<WIDTH>(v, SumOp{});
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
  return group_allreduce<WIDTH>(v, MaxOp{});
}

// Broadcast from lane `src` of the caller's aligned WIDTH-lane group.
template <int WIDTH, typename T>
__device__ __forceinline__ T group_broadcast(T v, int src) {
  if constexpr (WIDTH == 64) {
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  } else {
    return __shfl(v, src, WIDTH);
  }
}

}  // namespace sputnik_hip
