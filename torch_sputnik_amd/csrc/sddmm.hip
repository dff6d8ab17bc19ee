// SDDMM  out[p] = <lhs[i_p,:], rhs[j_p,:]>  for gfx950.
//
// Replaces sputnik::CudaSddmm as driven by src/sddmm_cuda.cu:45-54.
//
// One wave owns one output row (rows dealt in `row_indices` order).  The
// wave is cut into groups of LPN lanes; a group takes LPN consecutive
// nonzeros of the row at a time: lane t loads column index t (coalesced),
// then for each of the LPN nonzeros the whole group reads the matching rhs
// row with VEC-wide loads (one contiguous LPN*VEC*4-byte segment), multiplies
// with the lhs row fragment it keeps in registers and reduces with DPP
// (no LDS crossbar up to 16 lanes).  Lane t keeps result t, so the LPN
// results leave as one coalesced store.  Long inner dimensions are walked in
// panels of LPN*VEC*KSL elements; later panels add into the output.
#include <type_traits>

#include "common.h"
#include "mfma.h"
#include "options.h"
#include "wave_utils.h"

namespace sputnik_hip {

bool sddmm_tiled_applicable(int m, int k, int n, int nonzeros, const float* lhs,
                            int64_t lhs_stride, const float* rhs, int64_t rhs_stride);
size_t sddmm_tiled_workspace_bytes(int m, int k, int n, int nonzeros, bool summed = false);
int sddmm_tiled_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                     const int* row_offsets, const int* column_indices, void* workspace,
                     hipStream_t stream, bool summed = false, bool with_flat = false, int masks = 1,
                     int64_t mask_plan_ints = 0);
int sddmm_tiled_launch(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                       const int* row_offsets, const int* column_indices, const float* lhs,
                       int64_t lhs_stride, const float* rhs, int64_t rhs_stride, float* out,
                       int64_t out_stride, const void* workspace, hipStream_t stream,
                       int mask_heads = 0, int64_t mask_plan_ints = 0, bool flat = false);
// Bytes between the plans of consecutive masks in a "many mask" workspace.
// (the last word of a region is spare: the masks' start order, mask_start_word)
inline size_t sddmm_many_mask_plan_bytes(int m, int k, int n, int nonzeros) {
  return (sddmm_tiled_workspace_bytes(m, k, n, nonzeros) + sizeof(int) + 255) / 256 * 256;
}

int sddmm_tiled_panels(int m, int k, int n, int nonzeros);
bool sddmm_flat_applicable(int m, int k, int n, int nonzeros, int elem_bytes);   // sddmm_flat.hip
int sddmm_tiled_panel_width(int k);
int sddmm_tiled_launch_partials(int m, int k, int n, int nonzeros, int replicas,
                                const int* row_indices, const int* row_offsets,
                                const int* column_indices, const float* lhs, int64_t lhs_stride,
                                const float* rhs, int64_t rhs_stride, float* partials,
                                const void* workspace, hipStream_t stream);

// native half operands (sddmm_tiled.hip)
bool sddmm_tiled_applicable_half(int m, int k, int n, int nonzeros, const void* lhs,
                                 int64_t lhs_stride, const void* rhs, int64_t rhs_stride);
int sddmm_tiled_passes_half(int k);
int sddmm_tiled_launch_half(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                            const int* row_offsets, const int* column_indices, const void* lhs,
                            int64_t lhs_stride, const void* rhs, int64_t rhs_stride, void* out,
                            int64_t out_stride, int in_type, int out_type, const void* workspace,
                            hipStream_t stream, int mask_heads = 0, int64_t mask_plan_ints = 0,
                            bool flat = false);
bool sddmm_tiled_sum_half_served(int m, int k, int n, int nonzeros);
int sddmm_tiled_launch_partials_half(int m, int k, int n, int nonzeros, int replicas,
                                     const int* row_indices, const int* row_offsets,
                                     const int* column_indices, const void* lhs,
                                     int64_t lhs_stride, const void* rhs, int64_t rhs_stride,
                                     int in_type, float* partials, const void* workspace,
                                     hipStream_t stream);

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

// VEC elements of storage type T, widened to float (T = float: load_vec of common.h).
template <int VEC, typename T>
__device__ __forceinline__ void load_elems(float (&dst)[VEC], const T* __restrict__ src) {
  if constexpr (std::is_same_v<T, float>) {
    load_vec<VEC>(dst, src);
  } else if constexpr (VEC == 1) {
    dst[0] = static_cast<float>(*src);
  } else {
    using V = T __attribute__((ext_vector_type(VEC)));
    const V v = *reinterpret_cast<const V*>(src);
#pragma unroll
    for (int e = 0; e < VEC; ++e) dst[e] = static_cast<float>(v[e]);
  }
}

// T: storage type of lhs / rhs (float; _Float16 / __bf16 = native half operands, widened
// in registers), TO: of the output.  All arithmetic is float.
template <typename T, typename TO, int VEC, int LPN, int KSL>
__global__ __launch_bounds__(kBlock) void sddmm_rowwave_kernel(
    int m, int k, const int* __restrict__ row_indices, const int* __restrict__ row_offsets,
    const int* __restrict__ column_indices, const T* __restrict__ lhs, int64_t lhs_stride,
    const T* __restrict__ rhs, int64_t rhs_stride, TO* __restrict__ out,
    int64_t out_stride, int mask_heads, int first_replica) {
  constexpr int kGroups = kWave / LPN;
  constexpr int kPanel = LPN * VEC * KSL;
  const int wave = threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  const int g = lane / LPN;
  const int l = lane % LPN;
  const int slot = blockIdx.x * kWavesPerBlock + wave;
  if (slot >= m) return;  // wave-uniform
  const int replica = blockIdx.y;
  lhs += replica * lhs_stride;
  rhs += replica * rhs_stride;
  out += replica * out_stride;
  {
    const MaskPlace place = select_mask(mask_heads, first_replica + replica, m, 0, row_offsets);
    row_offsets += static_cast<int64_t>(place.mask) * (m + 1);
    column_indices += place.first;
    row_indices += static_cast<int64_t>(place.mask) * m;
  }

  const int row = row_indices[slot];
  const int p0 = row_offsets[row];
  const int p1 = row_offsets[row + 1];
  const int nblocks = (p1 - p0 + LPN - 1) / LPN;
  const T* __restrict__ lhs_row = lhs + static_cast<int64_t>(row) * k;

  for (int kp = 0; kp < k; kp += kPanel) {
    // lhs fragment of this panel: slice s covers columns kp + (s*LPN + l)*VEC.
    float a[KSL][VEC];
#pragma unroll
    for (int s = 0; s < KSL; ++s) {
      const int c = kp + (s * LPN + l) * VEC;
      if (c < k) {
        load_elems<VEC, T>(a[s], lhs_row + c);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[s][v] = 0.f;
      }
    }

    // (round 5, as the SpMM row gather: a block of entries was a chain of round trips --
    // its columns on demand, then the rhs rows two at a time; now the NEXT block's columns
    // are in flight while this one is worked on and the rhs rows go out kBatch at a time;
    // an entry past the row's end gathers nothing.  Same sums in the same order.)
    constexpr int kBatch = (KSL * VEC <= 8) ? (LPN < 4 ? LPN : 4) : (LPN < 2 ? LPN : 2);
    int j_mine = (g < nblocks && p0 + g * LPN + l < p1) ? column_indices[p0 + g * LPN + l] : 0;
    for (int b = g; b < nblocks; b += kGroups) {
      const int pb = p0 + b * LPN;
      const int q = pb + l;
      const int q_next = q + kGroups * LPN;
      const int j_next = (q_next < p1) ? column_indices[q_next] : 0;
      const int cnt = min(LPN, p1 - pb);
      float result = 0.f;
      for (int t0 = 0; t0 < cnt; t0 += kBatch) {
        float bv[kBatch][KSL][VEC];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int j = __shfl(j_mine, t0 + u, LPN);
          const T* __restrict__ rhs_row = rhs + static_cast<int64_t>(j) * k;
#pragma unroll
          for (int s = 0; s < KSL; ++s) {
            const int c = kp + (s * LPN + l) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) bv[u][s][v] = 0.f;
            if (c < k && t0 + u < cnt) load_elems<VEC, T>(bv[u][s], rhs_row + c);
          }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          float partial = 0.f;
#pragma unroll
          for (int s = 0; s < KSL; ++s)
#pragma unroll
            for (int v = 0; v < VEC; ++v) partial = fmaf(a[s][v], bv[u][s][v], partial);
          const float total = group_sum<LPN>(partial);
          if (l == t0 + u) result = total;
        }
      }
      j_mine = j_next;
      if (q < p1) {
        if (kp == 0) {
          out[q] = static_cast<TO>(result);
        } else {
          out[q] = static_cast<TO>(static_cast<float>(out[q]) + result);
        }
      }
    }
  }
}

template <typename T, typename TO, int VEC, int LPN, int KSL>
int launch(int m, int k, int replicas, const int* row_indices, const int* row_offsets,
           const int* column_indices, const T* lhs, int64_t lhs_stride, const T* rhs,
           int64_t rhs_stride, TO* out, int64_t out_stride, hipStream_t stream,
           int mask_heads) {
  const int gx = ceil_div(m, kWavesPerBlock);
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    hipLaunchKernelGGL((sddmm_rowwave_kernel<T, TO, VEC, LPN, KSL>), dim3(gx, ry), dim3(kBlock), 0,
                       stream, m, k, row_indices, row_offsets, column_indices,
                       lhs + r0 * lhs_stride, lhs_stride, rhs + r0 * rhs_stride, rhs_stride,
                       out + r0 * out_stride, out_stride, mask_heads, r0);
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

template <int VEC, typename T = float, typename TO = float>
int launch_vec(int m, int k, int replicas, const int* row_indices, const int* row_offsets,
               const int* column_indices, const T* lhs, int64_t lhs_stride, const T* rhs,
               int64_t rhs_stride, TO* out, int64_t out_stride, hipStream_t stream,
               int mask_heads = 0) {
  const int lanes = ceil_div(k, VEC);
#define SPUTNIK_HIP_SD(LPN, KSL)                                                                \
  return launch<T, TO, VEC, LPN, KSL>(m, k, replicas, row_indices, row_offsets, column_indices, \
                                      lhs, lhs_stride, rhs, rhs_stride, out, out_stride, stream, \
                                      mask_heads)
  if (lanes <= 4) SPUTNIK_HIP_SD(4, 1);
  if (lanes <= 8) SPUTNIK_HIP_SD(8, 1);
  if (lanes <= 16) SPUTNIK_HIP_SD(16, 1);
  if (lanes <= 32) SPUTNIK_HIP_SD(32, 1);
  if (lanes <= 64) SPUTNIK_HIP_SD(64, 1);
  if (lanes <= 128) SPUTNIK_HIP_SD(64, 2);
  SPUTNIK_HIP_SD(64, 4);
#undef SPUTNIK_HIP_SD
}

// out[i] = partials[0][i] + partials[1][i] + ... in that order (deterministic).
template <int VEC>
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(int count /* of VEC-wide pieces */,
                                                              int parts, int64_t stride,
                                                              const float* __restrict__ partials,
                                                              float* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  float acc[VEC];
  load_vec<VEC>(acc, partials + static_cast<int64_t>(i) * VEC);
#pragma unroll 4
  for (int z = 1; z < parts; ++z) {
    float v[VEC];
    load_vec<VEC>(v, partials + z * stride + static_cast<int64_t>(i) * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) out[static_cast<int64_t>(i) * VEC + e] = acc[e];
}

// The same for up to four sums at once (grid y = the sum): the weight gradients of a group
// of projections, whose four-microsecond sums are launch time and little else.
constexpr int kMaxSumGroup = 4;
struct SumGroup {
  const float* partials[kMaxSumGroup];
  float* out[kMaxSumGroup];
  int length[kMaxSumGroup];   // floats per vector (= distance between the partial vectors)
  int parts[kMaxSumGroup];
};
__global__ __launch_bounds__(kBlock) void sum_partials_group_kernel(SumGroup g) {
  const int which = blockIdx.y;
  // (selected with compares, not by indexing the argument: no copy of it in scratch memory)
  const float* __restrict__ partials = which == 0 ? g.partials[0] : which == 1 ? g.partials[1]
                                       : which == 2 ? g.partials[2] : g.partials[3];
  float* __restrict__ out = which == 0 ? g.out[0] : which == 1 ? g.out[1] : which == 2 ? g.out[2] : g.out[3];
  const int length = which == 0 ? g.length[0] : which == 1 ? g.length[1] : which == 2 ? g.length[2] : g.length[3];
  const int parts = which == 0 ? g.parts[0] : which == 1 ? g.parts[1] : which == 2 ? g.parts[2] : g.parts[3];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= length) return;
  float acc = partials[i];
#pragma unroll 4
  for (int z = 1; z < parts; ++z) acc += partials[static_cast<int64_t>(z) * length + i];
  out[i] = acc;
}

// out[i] = partials[0][i] + ... + partials[parts - 1][i]  (one launch, index order)
int sum_partials_launch(int nonzeros, int parts, const float* partials, float* out, hipStream_t stream) {
  if (nonzeros % 4 == 0 && aligned_to(out, 16) && aligned_to(partials, 16)) {
    hipLaunchKernelGGL(sum_partials_kernel<4>, dim3(ceil_div(nonzeros / 4, kBlock)), dim3(kBlock), 0,
                       stream, nonzeros / 4, parts, static_cast<int64_t>(nonzeros), partials, out);
  } else {
    hipLaunchKernelGGL(sum_partials_kernel<1>, dim3(ceil_div(nonzeros, kBlock)), dim3(kBlock), 0, stream,
                       nonzeros, parts, static_cast<int64_t>(nonzeros), partials, out);
  }
  return launch_status();
}

// Widest vector (4, 2 or 1 ELEMENTS of `elem` bytes) every row start supports.
int vector_width_of(const void* p, int64_t width, int64_t stride, size_t elem) {
  if (width % 4 == 0 && stride % 4 == 0 && aligned_to(p, 4 * elem)) return 4;
  if (width % 2 == 0 && stride % 2 == 0 && aligned_to(p, 2 * elem)) return 2;
  return 1;
}

// Row-wave kernel on half operands.  single: the whole k in one panel (64 lanes x VEC x 4).
template <typename T, typename TO>
int rowwave_half(int m, int k, int replicas, const int* row_indices, const int* row_offsets,
                 const int* column_indices, const void* lhs, int64_t lhs_stride, const void* rhs,
                 int64_t rhs_stride, void* out, int64_t out_stride, hipStream_t stream) {
  const T* l = static_cast<const T*>(lhs);
  const T* r = static_cast<const T*>(rhs);
  TO* o = static_cast<TO*>(out);
  const int vec = min(vector_width_of(lhs, k, lhs_stride, sizeof(T)),
                      vector_width_of(rhs, k, rhs_stride, sizeof(T)));
  switch (vec) {
    case 4: return launch_vec<4, T, TO>(m, k, replicas, row_indices, row_offsets, column_indices, l,
                                        lhs_stride, r, rhs_stride, o, out_stride, stream);
    case 2: return launch_vec<2, T, TO>(m, k, replicas, row_indices, row_offsets, column_indices, l,
                                        lhs_stride, r, rhs_stride, o, out_stride, stream);
    default: return launch_vec<1, T, TO>(m, k, replicas, row_indices, row_offsets, column_indices, l,
                                         lhs_stride, r, rhs_stride, o, out_stride, stream);
  }
}

}  // namespace

int sum_partial_vectors(int nonzeros, int parts, const float* partials, float* out, hipStream_t stream) {
  return sum_partials_launch(nonzeros, parts, partials, out, stream);
}

}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

size_t sputnik_hip_sddmm_workspace_bytes(int m, int k, int n, int nonzeros) {
  if (m <= 0 || k <= 0 || n <= 0 || nonzeros <= 0) return 0;
  return sddmm_tiled_workspace_bytes(m, k, n, nonzeros);
}

namespace {

// Does this call take the LDS-tiled kernels?  Small calls are launch-latency
// bound: the row-wave kernel is one launch, the tiled path a pre-pass plus a
// kernel that first stages its slab.  Measured cross-over
// (tools/small_sddmm.py): about 2.7e8 multiply-adds at k = 64, 1.3e8 at k =
// 128, 3e7 at k = 512, i.e. nnz * k^2 * replicas ~ 2^34.
// Test knob SPUTNIK_HIP_SDDMM_KERNEL (options.h: read once): "tiled" / "wave".
// Round 5: the one-number rule (nnz * k^2 * replicas < 2^34: row-wave) was up to 1.9 x off
// in a band of mid-sized batched calls -- 1024^2 at density 0.05 x 64 replicas, k = 64: 71 us
// on the row-wave kernel where the tiled path takes 40; tools/small_sddmm.py, 60 shapes,
// profiles/r5_sddmm_tiled_vs_wave.jsonl: mean regret 14 %, worst 86 %.  The float product now
// compares two estimates fitted to that sweep (mean regret 1.5 %, worst 26 %), in us:
//   row-wave: a wave per (row, replica), 0.85 ns each, plus the entries' dot products
//     (8.2 / 28 / 55 / 90 us per million entries at panel width 64 / 128 / 256 / 512), at
//     least the time ONE wave needs for an average row (0.72 ns per element), at least 13;
//   tiled: 20 (13 with the plan made ahead), 10 us per million entries of pre-pass, the
//     rounds of slab-staging workgroups (9 us per round of 512 at widths <= 128, 4 / 15 per
//     round of 256 at 256 / 512), 1.8 us per million entries and 64 elements.
// `summed`: the form summed over the replicas, whose panels (its own width) run side by
// side in ONE launch.  And a workgroup is no faster than its own entries allow, 17 ps per
// entry and element of the panel -- a mask of few workgroups with many entries each (256^2
// at density 0.5, k = 256, 8 replicas: 16 workgroups, 56 us against 28 on the row-wave
// kernel) is not carried by the chip-wide rate.
bool float_call_is_small(int m, int k, int n, int nonzeros, int replicas, bool planned,
                         bool summed = false) {
  const int width = summed ? (sddmm_tiled_panels(m, k, n, nonzeros) > 0
                                  ? k / sddmm_tiled_panels(m, k, n, nonzeros) : 0)
                           : sddmm_tiled_panel_width(k);
  if (width == 0 || m <= 0 || (width != 64 && width != 128 && width != 256 && width != 512)) return true;
  const double entries = static_cast<double>(nonzeros) * replicas, panels = k / width;
  const double per_million = width <= 64 ? 8.2 : width <= 128 ? 28.0 : width <= 256 ? 55.0 : 90.0;
  const double wave = std::max({13.0, 0.85e-3 * m * replicas + per_million * 1e-6 * entries * panels,
                                0.72e-3 * nonzeros / m * k});
  const int slab_rows = width <= 64 ? 256 : width <= 256 ? 128 : 64;
  const double per_replica = static_cast<double>(ceil_div(m, 256)) * ceil_div(n, slab_rows);
  const double rounds = per_replica * replicas / (width <= 128 ? 512 : 256) * panels;
  const double per_round = width <= 128 ? 9.0 : width <= 256 ? 4.0 : 15.0;
  const double own_entries = 1.7e-5 * (nonzeros / per_replica) * width * (summed ? 1.0 : panels);
  const double tiled = (planned ? 13.0 : 20.0 + 10e-6 * nonzeros) +
                       std::max(per_round * rounds + 1.8e-6 * entries * (k / 64.0), own_entries);
  // (fitted with a 10 % lean towards the tiled path; the row-wave kernel has since learnt to
  // keep the next block's columns and four rhs rows in flight -- 1.2-2 x faster on long rows --
  // and on the sweep repeated with it, profiles/r5_sddmm_tiled_vs_wave_final.jsonl, the
  // estimates as they stand decide best WITHOUT the lean: mean regret 0.9 %, worst 23 %)
  return tiled >= wave;
}

bool takes_tiled(int m, int k, int n, int nonzeros, int replicas /* < 0: unknown */,
                 const float* lhs, int64_t lhs_stride, const float* rhs, int64_t rhs_stride,
                 const void* workspace, size_t workspace_bytes, bool summed = false,
                 bool planned = false) {
  const bool force_tiled = options().sddmm_kernel == 1;
  const bool force_wave = options().sddmm_kernel == 2;
  const bool small = replicas >= 0 && float_call_is_small(m, k, n, nonzeros, replicas, planned, summed);
  return !force_wave && (force_tiled || !small) && workspace != nullptr &&
         aligned_to(workspace, 16) &&
         sddmm_tiled_applicable(m, k, n, nonzeros, lhs, lhs_stride, rhs, rhs_stride) &&
         workspace_bytes >= sddmm_tiled_workspace_bytes(m, k, n, nonzeros, summed);
}

int sddmm_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
               const int* row_offsets, const int* column_indices, const float* lhs,
               int64_t lhs_stride, const float* rhs, int64_t rhs_stride, float* out,
               int64_t out_stride, void* workspace, size_t workspace_bytes, bool planned,
               hipStream_t stream, int mask_heads = 0, const int* mask_nonzeros = nullptr) {
  // mask_heads > 0 ("many mask", many_mask.hip): concatenated topologies, replica r
  // under number r / mask_heads; `nonzeros` = the largest mask's count (kernel choice
  // and plan size), mask_nonzeros (host) = every mask's count; `workspace` holds one
  // plan per mask, sddmm_many_mask_plan_bytes apart.
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || replicas == 0) return 0;
  if (k == 0) {
    for (int r = 0; r < replicas; ++r) {
      const hipError_t e =
          hipMemsetAsync(out + r * out_stride, 0, sizeof(float) * static_cast<size_t>(nonzeros), stream);
      if (e != hipSuccess) return static_cast<int>(e);
    }
    return 0;
  }
  const int masks = mask_heads > 0 ? replicas / mask_heads : 1;
  const size_t plan_bytes = sddmm_many_mask_plan_bytes(m, k, n, nonzeros);
  if (takes_tiled(m, k, n, nonzeros, replicas, lhs, lhs_stride, rhs, rhs_stride, workspace,
                  mask_heads > 0 ? workspace_bytes / masks : workspace_bytes, /*summed=*/false,
                  planned) &&
      (mask_heads == 0 || plan_bytes * masks <= workspace_bytes)) {
    if (!planned) {
      // (many mask: the tables of all topologies in ONE launch, round 4; the masks share
      // m, n and k, so they share the table's shape)
      const int st = sddmm_tiled_plan(m, k, n, nonzeros, row_indices, row_offsets, column_indices,
                                      workspace, stream, /*summed=*/false, /*with_flat=*/false,
                                      mask_heads > 0 ? masks : 1,
                                      static_cast<int64_t>(plan_bytes / sizeof(int)));
      if (st != 0) return st;
    }
    // (a plan that was made ahead of the call carries the pair-flat kernel's lists too)
    return sddmm_tiled_launch(m, k, n, nonzeros, replicas, row_indices, row_offsets,
                              column_indices, lhs, lhs_stride, rhs, rhs_stride, out, out_stride,
                              workspace, stream, mask_heads, static_cast<int64_t>(plan_bytes / sizeof(int)),
                              /*flat=*/planned);
  }
  int vec = vector_width(lhs, k, lhs_stride);
  vec = min(vec, vector_width(rhs, k, rhs_stride));
  switch (vec) {
    case 4:
      return launch_vec<4>(m, k, replicas, row_indices, row_offsets, column_indices, lhs,
                           lhs_stride, rhs, rhs_stride, out, out_stride, stream, mask_heads);
    case 2:
      return launch_vec<2>(m, k, replicas, row_indices, row_offsets, column_indices, lhs,
                           lhs_stride, rhs, rhs_stride, out, out_stride, stream, mask_heads);
    default:
      return launch_vec<1>(m, k, replicas, row_indices, row_offsets, column_indices, lhs,
                           lhs_stride, rhs, rhs_stride, out, out_stride, stream, mask_heads);
  }
}

}  // namespace

namespace {

// SDDMM on float16 / bfloat16 operands read as they are (in_type), output float or
// in_type.  A half OUTPUT needs the product in one pass (later passes would add into
// rounded values): SPUTNIK_HIP_UNSUPPORTED otherwise, the caller takes a float output.
int sddmm_exec_half(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                    const int* row_offsets, const int* column_indices, const void* lhs,
                    int64_t lhs_stride, const void* rhs, int64_t rhs_stride, void* out,
                    int64_t out_stride, int in_type, int out_type, void* workspace,
                    size_t workspace_bytes, bool planned, hipStream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if ((in_type != SPUTNIK_HIP_F16 && in_type != SPUTNIK_HIP_BF16) ||
      (out_type != SPUTNIK_HIP_F32 && out_type != in_type))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || replicas == 0) return 0;
  const size_t out_elem = out_type == SPUTNIK_HIP_F32 ? 4 : 2;
  if (!aligned_to(lhs, 2) || !aligned_to(rhs, 2) || !aligned_to(out, out_elem))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (k == 0) {
    for (int r = 0; r < replicas; ++r) {
      const hipError_t e = hipMemsetAsync(static_cast<char*>(out) + r * out_stride * out_elem, 0,
                                          out_elem * static_cast<size_t>(nonzeros), stream);
      if (e != hipSuccess) return static_cast<int>(e);
    }
    return 0;
  }
  const bool half_out = out_type != SPUTNIK_HIP_F32;
  const bool force_tiled = options().sddmm_kernel == 1;
  const bool force_wave = options().sddmm_kernel == 2;
  // (the float product's estimates: on the half sweep, tools/small_sddmm.py --half, the
  // one-number rule left 17 % mean regret, worst 2.9 x; these leave 1.1 %)
  const bool small = float_call_is_small(m, k, n, nonzeros, replicas, planned);
  const bool tiled = !force_wave && (force_tiled || !small) && workspace != nullptr &&
                     aligned_to(workspace, 16) &&
                     sddmm_tiled_applicable_half(m, k, n, nonzeros, lhs, lhs_stride, rhs, rhs_stride) &&
                     workspace_bytes >= sddmm_tiled_workspace_bytes(m, k, n, nonzeros) &&
                     (!half_out || sddmm_tiled_passes_half(k) == 1);
  if (tiled) {
    if (!planned) {
      const int st = sddmm_tiled_plan(m, k, n, nonzeros, row_indices, row_offsets, column_indices,
                                      workspace, stream);
      if (st != 0) return st;
    }
    return sddmm_tiled_launch_half(m, k, n, nonzeros, replicas, row_indices, row_offsets,
                                   column_indices, lhs, lhs_stride, rhs, rhs_stride, out, out_stride,
                                   in_type, out_type, workspace, stream, 0, 0, /*flat=*/planned);
  }
  if (half_out && k > 1024) return SPUTNIK_HIP_UNSUPPORTED;   // (row-wave panel: 64 lanes x 4 x 4)
  if (half_out && min(vector_width_of(lhs, k, lhs_stride, 2), vector_width_of(rhs, k, rhs_stride, 2)) * 256 < k)
    return SPUTNIK_HIP_UNSUPPORTED;
#define SPUTNIK_HIP_RW(T, TO)                                                                   \
  return rowwave_half<T, TO>(m, k, replicas, row_indices, row_offsets, column_indices, lhs,      \
                             lhs_stride, rhs, rhs_stride, out, out_stride, stream)
  if (in_type == SPUTNIK_HIP_F16 && !half_out) SPUTNIK_HIP_RW(_Float16, float);
  if (in_type == SPUTNIK_HIP_F16) SPUTNIK_HIP_RW(_Float16, _Float16);
  if (!half_out) SPUTNIK_HIP_RW(__bf16, float);
  SPUTNIK_HIP_RW(__bf16, __bf16);
#undef SPUTNIK_HIP_RW
}

}  // namespace

// One launch for all masks of a "many mask" batch (many_mask.hip; C linkage like its
// surroundings, hidden visibility: not part of the ABI).
int sputnik_hip_internal_sddmm_many_mask(int masks, int m, int k, int n, const int* nonzeros, int largest, int replicas,
                    const int* row_indices, const int* row_offsets, const int* column_indices,
                    const float* lhs, int64_t lhs_stride, const float* rhs, int64_t rhs_stride,
                    float* out, int64_t out_stride, void* workspace, size_t workspace_bytes,
                    hipStream_t stream) {
  return sddmm_exec(m, k, n, largest, replicas, row_indices, row_offsets, column_indices, lhs,
                    lhs_stride, rhs, rhs_stride, out, out_stride, workspace, workspace_bytes,
                    /*planned=*/false, stream, replicas / masks, nonzeros);
}

int sputnik_hip_sddmm_batched(int m, int k, int n, int nonzeros, int replicas,
                              const int* row_indices, const int* row_offsets,
                              const int* column_indices, const float* lhs, int64_t lhs_stride,
                              const float* rhs, int64_t rhs_stride, float* out,
                              int64_t out_stride, void* workspace, size_t workspace_bytes,
                              sputnik_hip_stream_t stream) {
  return sddmm_exec(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices, lhs,
                    lhs_stride, rhs, rhs_stride, out, out_stride, workspace, workspace_bytes,
                    /*planned=*/false, stream);
}

namespace {

// The summed form's workspace: the vector kernels' tables, then (256-byte aligned) the plan
// of the matrix-core route when SOME call on this shape could take it (the queries know
// neither the operands' storage type nor the replica count).
bool mfma_for_some_call(int m, int k, int n, int nonzeros) {
  return sddmm_mfma_shape(m, k, n, nonzeros, /*replicas=*/1 << 20);
}
size_t mfma_plan_offset(int m, int k, int n, int nonzeros) {
  return (sddmm_tiled_workspace_bytes(m, k, n, nonzeros, /*summed=*/true) + 255) / 256 * 256;
}
size_t sum_workspace_bytes(int m, int k, int n, int nonzeros) {
  if (!mfma_for_some_call(m, k, n, nonzeros))
    return sddmm_tiled_workspace_bytes(m, k, n, nonzeros, /*summed=*/true);
  return mfma_plan_offset(m, k, n, nonzeros) + sddmm_mfma_plan_bytes(m, n);
}

// Partial vectors a summed call needs: one per (replica, panel) on the tiled path.
int64_t sum_parts(int m, int k, int n, int nonzeros, int replicas) {
  const bool tiled_shape = sddmm_tiled_workspace_bytes(m, k, n, nonzeros) != 0;
  return static_cast<int64_t>(replicas) * (tiled_shape ? sddmm_tiled_panels(m, k, n, nonzeros) : 1);
}

// `leave_parts` != nullptr: the call stops in front of its last launch and says how many
// partial vectors of `nonzeros` floats it left in `scratch` (0: `out` is complete) -- for
// sputnik_hip_sddmm_sum_group_planned, which adds the vectors of several calls in one launch.
int sddmm_sum_exec(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                   const int* row_offsets, const int* column_indices, const float* lhs,
                   int64_t lhs_stride, const float* rhs, int64_t rhs_stride, float* out,
                   void* workspace, size_t workspace_bytes, bool planned, void* scratch,
                   size_t scratch_bytes, hipStream_t stream, int* leave_parts = nullptr) {
  if (leave_parts != nullptr) *leave_parts = 0;
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0) return 0;
  if (k == 0 || replicas == 0) {
    const hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * static_cast<size_t>(nonzeros), stream);
    return static_cast<int>(e);
  }
  bool tiled = takes_tiled(m, k, n, nonzeros, replicas, lhs, lhs_stride, rhs, rhs_stride,
                           workspace, workspace_bytes, /*summed=*/true);
  // (the tiled form puts every (replica, panel) pair on grid z: beyond 65535 of them --
  // a SparseLinear weight gradient over a very large batch -- the row-wave kernel, which
  // walks z in chunks, writes the per-replica partial products instead)
  if (tiled && static_cast<int64_t>(replicas) * sddmm_tiled_panels(m, k, n, nonzeros) > kMaxGridYZ)
    tiled = false;
  const int panels = tiled ? sddmm_tiled_panels(m, k, n, nonzeros) : 1;
  const int64_t parts = static_cast<int64_t>(replicas) * panels;
  if (parts == 1)
    // One replica and one panel: the plain product, which plans for itself -- a plan made
    // for the summed form has that form's slabs and no pair-flat lists behind the tables,
    // and the plain dispatch may choose differently from the `tiled` above.
    return sddmm_exec(m, k, n, nonzeros, 1, row_indices, row_offsets, column_indices, lhs,
                      lhs_stride, rhs, rhs_stride, out, 0, tiled ? workspace : nullptr,
                      tiled ? workspace_bytes : 0, /*planned=*/false, stream);
  if (scratch == nullptr || !aligned_to(scratch, 16) ||
      scratch_bytes < sizeof(float) * static_cast<size_t>(parts) * nonzeros)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  float* partials = static_cast<float*>(scratch);
  int st;
  if (tiled) {
    if (!planned) {
      st = sddmm_tiled_plan(m, k, n, nonzeros, row_indices, row_offsets, column_indices,
                            workspace, stream, /*summed=*/true);
      if (st != 0) return st;
    }
    st = sddmm_tiled_launch_partials(m, k, n, nonzeros, replicas, row_indices, row_offsets,
                                     column_indices, lhs, lhs_stride, rhs, rhs_stride, partials,
                                     workspace, stream);
  } else {
    // (row-wave kernel: the workspace, planned for the summed form or not, is not used)
    st = sddmm_exec(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices, lhs,
                    lhs_stride, rhs, rhs_stride, partials, nonzeros, nullptr, 0, false, stream);
  }
  if (st != 0) return st;
  const int nparts = static_cast<int>(parts);
  if (leave_parts != nullptr) {
    *leave_parts = nparts;
    return 0;
  }
  if (nonzeros % 4 == 0 && aligned_to(out, 16)) {
    hipLaunchKernelGGL(sum_partials_kernel<4>, dim3(ceil_div(nonzeros / 4, kBlock)), dim3(kBlock),
                       0, stream, nonzeros / 4, nparts, static_cast<int64_t>(nonzeros), partials,
                       out);
  } else {
    hipLaunchKernelGGL(sum_partials_kernel<1>, dim3(ceil_div(nonzeros, kBlock)), dim3(kBlock), 0,
                       stream, nonzeros, nparts, static_cast<int64_t>(nonzeros), partials, out);
  }
  return launch_status();
}

// Long reductions over a mask that occupies every tile: the dense tiles on the matrix
// cores, sampled at the mask (sddmm_mfma.hip; half operands only).  Its plan sits behind
// the vector kernels' tables in the summed form's workspace (made here when the call is
// not planned; without the room the kernel finds a tile's entries by a walk); the partial
// vectors of the workgroups that share a tile go to `scratch`, without which (or with too
// little of it) one workgroup per tile writes straight into `out`.
int sum_on_matrix_cores(int m, int k, int n, int nonzeros, int replicas, const int* row_offsets,
                        const int* column_indices, const void* lhs, int64_t lhs_stride,
                        const void* rhs, int64_t rhs_stride, int in_type, float* out,
                        void* workspace, size_t workspace_bytes, bool planned, void* scratch,
                        size_t scratch_bytes, hipStream_t stream, int planes,
                        int64_t lhs_plane_stride, int64_t rhs_plane_stride) {
  int splits = sddmm_mfma_splits(m, k, n, replicas, planes);
  if (splits > 1 && (scratch == nullptr || !aligned_to(scratch, 16) ||
                     scratch_bytes < sizeof(float) * static_cast<size_t>(splits) * nonzeros))
    splits = 1;
  void* plan = nullptr;
  const size_t plan_at = mfma_plan_offset(m, k, n, nonzeros);
  if (workspace != nullptr && aligned_to(workspace, 16) &&
      workspace_bytes >= plan_at + sddmm_mfma_plan_bytes(m, n)) {
    plan = static_cast<char*>(workspace) + plan_at;
    if (!planned) {
      const int st = sddmm_mfma_plan(m, n, row_offsets, column_indices, plan, stream);
      if (st != 0) return st;
    }
  }
  float* dst = splits == 1 ? out : static_cast<float*>(scratch);
  const int st = sddmm_mfma_launch(m, k, n, nonzeros, replicas, row_offsets, column_indices, lhs,
                                   lhs_stride, rhs, rhs_stride, in_type, dst, splits, plan, stream,
                                   planes, lhs_plane_stride, rhs_plane_stride);
  if (st != 0 || splits == 1) return st;
  if (nonzeros % 4 == 0 && aligned_to(out, 16)) {
    hipLaunchKernelGGL(sum_partials_kernel<4>, dim3(ceil_div(nonzeros / 4, kBlock)), dim3(kBlock),
                       0, stream, nonzeros / 4, splits, static_cast<int64_t>(nonzeros), dst, out);
  } else {
    hipLaunchKernelGGL(sum_partials_kernel<1>, dim3(ceil_div(nonzeros, kBlock)), dim3(kBlock), 0,
                       stream, nonzeros, splits, static_cast<int64_t>(nonzeros), dst, out);
  }
  return launch_status();
}

// The same on float16 / bfloat16 operands read as they are; partial vectors and the
// sum are float32.
int sddmm_sum_exec_half(int m, int k, int n, int nonzeros, int replicas, const int* row_indices,
                        const int* row_offsets, const int* column_indices, const void* lhs,
                        int64_t lhs_stride, const void* rhs, int64_t rhs_stride, int in_type,
                        float* out, void* workspace, size_t workspace_bytes, bool planned,
                        void* scratch, size_t scratch_bytes, hipStream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (in_type != SPUTNIK_HIP_F16 && in_type != SPUTNIK_HIP_BF16) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0) return 0;
  if (k == 0 || replicas == 0) {
    const hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * static_cast<size_t>(nonzeros), stream);
    return static_cast<int>(e);
  }
  if (sddmm_mfma_applicable(m, k, n, nonzeros, replicas, lhs, lhs_stride, rhs, rhs_stride))
    return sum_on_matrix_cores(m, k, n, nonzeros, replicas, row_offsets, column_indices, lhs,
                               lhs_stride, rhs, rhs_stride, in_type, out, workspace, workspace_bytes,
                               planned, scratch, scratch_bytes, stream, 1, 0, 0);
  const bool force_tiled = options().sddmm_kernel == 1;
  const bool force_wave = options().sddmm_kernel == 2;
  const bool small = float_call_is_small(m, k, n, nonzeros, replicas, planned, /*summed=*/true);
  bool tiled = !force_wave && (force_tiled || !small) && workspace != nullptr &&
               aligned_to(workspace, 16) &&
               sddmm_tiled_applicable_half(m, k, n, nonzeros, lhs, lhs_stride, rhs, rhs_stride) &&
               sddmm_tiled_sum_half_served(m, k, n, nonzeros) &&
               workspace_bytes >= sddmm_tiled_workspace_bytes(m, k, n, nonzeros, /*summed=*/true);
  if (tiled && static_cast<int64_t>(replicas) * sddmm_tiled_panels(m, k, n, nonzeros) > kMaxGridYZ)
    tiled = false;
  const int panels = tiled ? sddmm_tiled_panels(m, k, n, nonzeros) : 1;
  const int64_t parts = static_cast<int64_t>(replicas) * panels;
  if (parts == 1)   // (the plain product, planning for itself: see sddmm_sum_exec)
    return sddmm_exec_half(m, k, n, nonzeros, 1, row_indices, row_offsets, column_indices, lhs,
                           lhs_stride, rhs, rhs_stride, out, 0, in_type, SPUTNIK_HIP_F32,
                           tiled ? workspace : nullptr, tiled ? workspace_bytes : 0,
                           /*planned=*/false, stream);
  if (scratch == nullptr || !aligned_to(scratch, 16) ||
      scratch_bytes < sizeof(float) * static_cast<size_t>(parts) * nonzeros)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  float* partials = static_cast<float*>(scratch);
  int st;
  if (tiled) {
    if (!planned) {
      st = sddmm_tiled_plan(m, k, n, nonzeros, row_indices, row_offsets, column_indices,
                            workspace, stream, /*summed=*/true);
      if (st != 0) return st;
    }
    st = sddmm_tiled_launch_partials_half(m, k, n, nonzeros, replicas, row_indices, row_offsets,
                                          column_indices, lhs, lhs_stride, rhs, rhs_stride, in_type,
                                          partials, workspace, stream);
  } else {
    st = sddmm_exec_half(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices, lhs,
                         lhs_stride, rhs, rhs_stride, partials, nonzeros, in_type, SPUTNIK_HIP_F32,
                         nullptr, 0, false, stream);
  }
  if (st != 0) return st;
  const int nparts = static_cast<int>(parts);
  if (nonzeros % 4 == 0 && aligned_to(out, 16)) {
    hipLaunchKernelGGL(sum_partials_kernel<4>, dim3(ceil_div(nonzeros / 4, kBlock)), dim3(kBlock),
                       0, stream, nonzeros / 4, nparts, static_cast<int64_t>(nonzeros), partials,
                       out);
  } else {
    hipLaunchKernelGGL(sum_partials_kernel<1>, dim3(ceil_div(nonzeros, kBlock)), dim3(kBlock), 0,
                       stream, nonzeros, nparts, static_cast<int64_t>(nonzeros), partials, out);
  }
  return launch_status();
}

}  // namespace

size_t sputnik_hip_sddmm_sum_workspace_bytes(int m, int k, int n, int nonzeros) {
  if (m <= 0 || k <= 0 || n <= 0 || nonzeros <= 0) return 0;
  return sum_workspace_bytes(m, k, n, nonzeros);
}

int sputnik_hip_sddmm_sum_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                               const int* row_offsets, const int* column_indices,
                               void* workspace, size_t workspace_bytes,
                               sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || k == 0) return 0;
  if (workspace == nullptr || !aligned_to(workspace, 16)) return 0;
  if (mfma_for_some_call(m, k, n, nonzeros) &&
      workspace_bytes >= mfma_plan_offset(m, k, n, nonzeros) + sddmm_mfma_plan_bytes(m, n)) {
    const int st = sddmm_mfma_plan(m, n, row_offsets, column_indices,
                                   static_cast<char*>(workspace) + mfma_plan_offset(m, k, n, nonzeros),
                                   stream);
    if (st != 0) return st;
  }
  const size_t need = sddmm_tiled_workspace_bytes(m, k, n, nonzeros, /*summed=*/true);
  if (need == 0 || workspace_bytes < need)
    return 0;  // nothing (else) to plan: the row-wave kernel needs no workspace
  return sddmm_tiled_plan(m, k, n, nonzeros, row_indices, row_offsets, column_indices, workspace,
                          stream, /*summed=*/true);
}

size_t sputnik_hip_sddmm_sum_scratch_bytes(int m, int k, int n, int nonzeros, int replicas) {
  if (m <= 0 || k <= 0 || n <= 0 || nonzeros <= 0 || replicas <= 0) return 0;
  int64_t parts = sum_parts(m, k, n, nonzeros, replicas);
  // (the query knows no storage type: room for the matrix-core route of half operands too)
  if (sddmm_mfma_shape(m, k, n, nonzeros, replicas))
    parts = max(parts, static_cast<int64_t>(sddmm_mfma_splits(m, k, n, replicas)));
  return parts <= 1 ? 0 : sizeof(float) * static_cast<size_t>(parts) * nonzeros;
}

int sputnik_hip_sddmm_sum_batched(int m, int k, int n, int nonzeros, int replicas,
                                  const int* row_indices, const int* row_offsets,
                                  const int* column_indices, const float* lhs,
                                  int64_t lhs_stride, const float* rhs, int64_t rhs_stride,
                                  float* out, void* workspace, size_t workspace_bytes,
                                  void* scratch, size_t scratch_bytes,
                                  sputnik_hip_stream_t stream) {
  return sddmm_sum_exec(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices,
                        lhs, lhs_stride, rhs, rhs_stride, out, workspace, workspace_bytes,
                        /*planned=*/false, scratch, scratch_bytes, stream);
}

int sputnik_hip_sddmm_sum_batched_planned(int m, int k, int n, int nonzeros, int replicas,
                                          const int* row_indices, const int* row_offsets,
                                          const int* column_indices, const float* lhs,
                                          int64_t lhs_stride, const float* rhs,
                                          int64_t rhs_stride, float* out, const void* workspace,
                                          size_t workspace_bytes, void* scratch,
                                          size_t scratch_bytes, sputnik_hip_stream_t stream) {
  return sddmm_sum_exec(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices,
                        lhs, lhs_stride, rhs, rhs_stride, out, const_cast<void*>(workspace),
                        workspace_bytes, /*planned=*/true, scratch, scratch_bytes, stream);
}

int sputnik_hip_sddmm_sum_group_planned(int m, int k, int n, int replicas, int count,
                                        const sputnik_hip_sddmm_sum_problem* problems,
                                        int64_t lhs_stride, int64_t rhs_stride,
                                        sputnik_hip_stream_t stream) {
  if (count < 0 || count > kMaxSumGroup || (count > 0 && problems == nullptr))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  SumGroup g = {};
  int sums = 0, longest = 0;
  for (int p = 0; p < count; ++p) {
    const sputnik_hip_sddmm_sum_problem& q = problems[p];
    int parts = 0;
    const int st = sddmm_sum_exec(m, k, n, q.nonzeros, replicas, q.row_indices, q.row_offsets,
                                  q.column_indices, q.lhs, lhs_stride, q.rhs, rhs_stride, q.out,
                                  const_cast<void*>(q.workspace), q.workspace_bytes, /*planned=*/true,
                                  q.scratch, q.scratch_bytes, stream, &parts);
    if (st != 0) return st;
    if (parts == 0) continue;   // (`out` is complete: an empty mask, or one replica of one panel)
    g.partials[sums] = static_cast<const float*>(q.scratch);
    g.out[sums] = q.out;
    g.length[sums] = q.nonzeros;
    g.parts[sums] = parts;
    longest = max(longest, q.nonzeros);
    ++sums;
  }
  if (sums == 0) return 0;
  hipLaunchKernelGGL(sum_partials_group_kernel, dim3(ceil_div(longest, kBlock), sums), dim3(kBlock), 0,
                     stream, g);
  return launch_status();
}

int sputnik_hip_sddmm_sum_typed(int m, int k, int n, int nonzeros, int replicas,
                                const int* row_indices, const int* row_offsets,
                                const int* column_indices, const void* lhs, int64_t lhs_stride,
                                const void* rhs, int64_t rhs_stride, int in_type, float* out,
                                void* workspace, size_t workspace_bytes, int planned,
                                void* scratch, size_t scratch_bytes, sputnik_hip_stream_t stream) {
  if (in_type == SPUTNIK_HIP_F32)
    return sddmm_sum_exec(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices,
                          static_cast<const float*>(lhs), lhs_stride,
                          static_cast<const float*>(rhs), rhs_stride, out, workspace,
                          workspace_bytes, planned != 0, scratch, scratch_bytes, stream);
  return sddmm_sum_exec_half(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices,
                             lhs, lhs_stride, rhs, rhs_stride, in_type, out, workspace,
                             workspace_bytes, planned != 0, scratch, scratch_bytes, stream);
}

namespace {
// (float32, half) operand pairs of the summed product: bytes of the float32 operand's two
// half planes in front of the partial vectors in `scratch`
size_t mixed_planes_bytes(int rows, int k, int replicas, int half_type) {
  return (static_cast<size_t>(sddmm_mfma_planes_of(half_type)) * replicas * rows * k * 2 + 255) / 256 * 256;
}
bool mixed_pair(int lhs_type, int rhs_type) {
  const bool lhs_half = lhs_type == SPUTNIK_HIP_F16 || lhs_type == SPUTNIK_HIP_BF16;
  const bool rhs_half = rhs_type == SPUTNIK_HIP_F16 || rhs_type == SPUTNIK_HIP_BF16;
  return (lhs_type == SPUTNIK_HIP_F32 && rhs_half) || (rhs_type == SPUTNIK_HIP_F32 && lhs_half);
}
}  // namespace

size_t sputnik_hip_sddmm_sum_mixed_scratch_bytes(int m, int k, int n, int nonzeros, int replicas,
                                                 int lhs_type, int rhs_type) {
  if (lhs_type == rhs_type) return sputnik_hip_sddmm_sum_scratch_bytes(m, k, n, nonzeros, replicas);
  if (m <= 0 || k <= 0 || n <= 0 || nonzeros <= 0 || replicas <= 0) return 0;
  if (!mixed_pair(lhs_type, rhs_type) || !sddmm_mfma_shape(m, k, n, nonzeros, replicas)) return 0;
  const int rows = lhs_type == SPUTNIK_HIP_F32 ? m : n;
  const int half_type = lhs_type == SPUTNIK_HIP_F32 ? rhs_type : lhs_type;
  return mixed_planes_bytes(rows, k, replicas, half_type) +
         sizeof(float) * static_cast<size_t>(sddmm_mfma_splits(m, k, n, replicas,
                                                               sddmm_mfma_planes_of(half_type))) * nonzeros;
}

int sputnik_hip_sddmm_sum_mixed(int m, int k, int n, int nonzeros, int replicas,
                                const int* row_indices, const int* row_offsets,
                                const int* column_indices, const void* lhs, int lhs_type,
                                int64_t lhs_stride, const void* rhs, int rhs_type,
                                int64_t rhs_stride, float* out, void* workspace,
                                size_t workspace_bytes, int planned, void* scratch,
                                size_t scratch_bytes, sputnik_hip_stream_t stream) {
  if (lhs_type == rhs_type)
    return sputnik_hip_sddmm_sum_typed(m, k, n, nonzeros, replicas, row_indices, row_offsets,
                                       column_indices, lhs, lhs_stride, rhs, rhs_stride, lhs_type,
                                       out, workspace, workspace_bytes, planned, scratch,
                                       scratch_bytes, stream);
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (!mixed_pair(lhs_type, rhs_type)) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || k == 0 || replicas == 0) return SPUTNIK_HIP_UNSUPPORTED;
  const bool lhs_float = lhs_type == SPUTNIK_HIP_F32;
  const int half_type = lhs_float ? rhs_type : lhs_type;
  const int rows = lhs_float ? m : n;
  const void* wide = lhs_float ? lhs : rhs;
  const int64_t wide_stride = lhs_float ? lhs_stride : rhs_stride;
  // the float32 operand is split in one flat pass: its replicas must lie back to back
  if (replicas > 1 && wide_stride != static_cast<int64_t>(rows) * k) return SPUTNIK_HIP_UNSUPPORTED;
  const size_t planes_bytes = mixed_planes_bytes(rows, k, replicas, half_type);
  if (scratch == nullptr || !aligned_to(scratch, 16) || scratch_bytes < planes_bytes ||
      !aligned_to(wide, 16))
    return SPUTNIK_HIP_UNSUPPORTED;
  const int64_t plane_elems = static_cast<int64_t>(replicas) * rows * k;
  char* hi = static_cast<char*>(scratch);
  const void* a = lhs_float ? static_cast<const void*>(hi) : lhs;
  const void* b = lhs_float ? rhs : static_cast<const void*>(hi);
  const int64_t a_stride = lhs_float ? static_cast<int64_t>(rows) * k : lhs_stride;
  const int64_t b_stride = lhs_float ? rhs_stride : static_cast<int64_t>(rows) * k;
  if (!sddmm_mfma_applicable(m, k, n, nonzeros, replicas, a, a_stride, b, b_stride))
    return SPUTNIK_HIP_UNSUPPORTED;
  const int st = sddmm_mfma_split_planes(plane_elems, static_cast<const float*>(wide), half_type, hi,
                                         stream);
  if (st != 0) return st;
  return sum_on_matrix_cores(m, k, n, nonzeros, replicas, row_offsets, column_indices, a, a_stride, b,
                             b_stride, half_type, out, workspace, workspace_bytes, planned != 0,
                             static_cast<char*>(scratch) + planes_bytes, scratch_bytes - planes_bytes,
                             stream, sddmm_mfma_planes_of(half_type), lhs_float ? plane_elems : 0,
                             lhs_float ? 0 : plane_elems);
}

int sputnik_hip_sddmm_plan(int m, int k, int n, int nonzeros, const int* row_indices,
                           const int* row_offsets, const int* column_indices, void* workspace,
                           size_t workspace_bytes, sputnik_hip_stream_t stream) {
  if (m < 0 || k < 0 || n < 0 || nonzeros < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || k == 0) return 0;
  // operand alignment is checked again by the planned call; here only the shape matters
  if (workspace == nullptr || !aligned_to(workspace, 16) ||
      sddmm_tiled_workspace_bytes(m, k, n, nonzeros) == 0 ||
      workspace_bytes < sddmm_tiled_workspace_bytes(m, k, n, nonzeros))
    return 0;  // nothing to plan: the row-wave kernel needs no workspace
  return sddmm_tiled_plan(m, k, n, nonzeros, row_indices, row_offsets, column_indices, workspace,
                          stream, /*summed=*/false, /*with_flat=*/true);
}

int sputnik_hip_sddmm_batched_planned(int m, int k, int n, int nonzeros, int replicas,
                                      const int* row_indices, const int* row_offsets,
                                      const int* column_indices, const float* lhs,
                                      int64_t lhs_stride, const float* rhs, int64_t rhs_stride,
                                      float* out, int64_t out_stride, const void* workspace,
                                      size_t workspace_bytes, sputnik_hip_stream_t stream) {
  return sddmm_exec(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices, lhs,
                    lhs_stride, rhs, rhs_stride, out, out_stride, const_cast<void*>(workspace),
                    workspace_bytes, /*planned=*/true, stream);
}

int sputnik_hip_sddmm_typed(int m, int k, int n, int nonzeros, int replicas,
                            const int* row_indices, const int* row_offsets,
                            const int* column_indices, const void* lhs, int64_t lhs_stride,
                            const void* rhs, int64_t rhs_stride, int in_type, void* out,
                            int64_t out_stride, int out_type, void* workspace,
                            size_t workspace_bytes, int planned, sputnik_hip_stream_t stream) {
  if (in_type == SPUTNIK_HIP_F32) {
    if (out_type != SPUTNIK_HIP_F32) return SPUTNIK_HIP_INVALID_ARGUMENT;
    return sddmm_exec(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices,
                      static_cast<const float*>(lhs), lhs_stride, static_cast<const float*>(rhs),
                      rhs_stride, static_cast<float*>(out), out_stride, workspace, workspace_bytes,
                      planned != 0, stream);
  }
  return sddmm_exec_half(m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices, lhs,
                         lhs_stride, rhs, rhs_stride, out, out_stride, in_type, out_type, workspace,
                         workspace_bytes, planned != 0, stream);
}

const char* sputnik_hip_sddmm_kernel_name(int m, int k, int n, int nonzeros, int replicas,
                                          int elem_bytes, int planned) {
  if (m <= 0 || k <= 0 || n <= 0 || nonzeros <= 0 || replicas <= 0) return "none";
  const bool force_tiled = options().sddmm_kernel == 1;
  const bool force_wave = options().sddmm_kernel == 2;
  const bool small = float_call_is_small(m, k, n, nonzeros, replicas, planned != 0);
  const bool shape = sddmm_tiled_workspace_bytes(m, k, n, nonzeros) != 0 &&
                     static_cast<int64_t>(n) * k * elem_bytes < (int64_t{1} << 32) &&
                     static_cast<int64_t>(m) * k * elem_bytes < (int64_t{1} << 32);
  if (force_wave || !(force_tiled || !small) || !shape) return "sddmm_rowwave_kernel";
  if (planned != 0 && sddmm_flat_applicable(m, k, n, nonzeros, elem_bytes)) return "sddmm_flat_kernel";
  if (elem_bytes == 2 || sddmm_tiled_panel_width(k) == 64) return "sddmm_quad_kernel";
  return "sddmm_stationary_kernel";
}

int sputnik_hip_sddmm(int m, int k, int n, int nonzeros, const int* row_indices,
                      const int* row_offsets, const int* column_indices, const float* lhs,
                      const float* rhs, float* out, sputnik_hip_stream_t stream) {
  return sputnik_hip_sddmm_batched(m, k, n, nonzeros, 1, row_indices, row_offsets,
                                   column_indices, lhs, 0, rhs, 0, out, 0, nullptr, 0, stream);
}

}  // extern "C"
