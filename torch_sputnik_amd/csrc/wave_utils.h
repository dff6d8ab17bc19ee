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

// Transposing reduction inside each 16-lane row: every lane brings 16 partial
// values p[0..15]; lane i (its index in the row) returns the row-wide sum of
// p[i].  Halving exchange: at each of the four steps a lane keeps one half of
// its values, hands the other half to a partner and adds what the partner
// hands over: 8 + 4 + 2 + 1 = 15 DPP adds (plus two selects each) for 16
// sums, against 16 x 4 for sixteen separate all-reduces.  The partners
// (i^8, i^7, i^2, i^1) are the four single-instruction DPP permutations that
// pair lanes differing in bit 3, 2, 1, 0 and together span the row.
constexpr int kDppRowRor8 = 0x128;  // lane i <- lane i ^ 8
__device__ __forceinline__ float row_transpose_sum16(const float (&p)[16], int i) {
  const bool b3 = (i & 8) != 0, b2 = (i & 4) != 0, b1 = (i & 2) != 0, b0 = (i & 1) != 0;
  float q[8], r[4], s[2];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float keep = b3 ? p[u + 8] : p[u], send = b3 ? p[u] : p[u + 8];
    q[u] = keep + dpp_f32<kDppRowRor8>(send);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float keep = b2 ? q[u + 4] : q[u], send = b2 ? q[u] : q[u + 4];
    r[u] = keep + dpp_f32<kDppRowHalfMirror>(send);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float keep = b1 ? r[u + 2] : r[u], send = b1 ? r[u] : r[u + 2];
    s[u] = keep + dpp_f32<kDppQuadXor2>(send);
  }
  const float keep = b0 ? s[1] : s[0], send = b0 ? s[0] : s[1];
  return keep + dpp_f32<kDppQuadXor1>(send);
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
