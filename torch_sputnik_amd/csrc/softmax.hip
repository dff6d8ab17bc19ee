// Sparse softmax over the stored entries of each CSR row, for gfx950.
//
// Replaces sputnik::SparseSoftmax as driven by src/softmax_cuda.cu:35-43.
//
// A group of 16 / 32 / 64 lanes owns a run of consecutive rows; a row moves
// through registers once, as ALIGNED 16-byte pieces (one dwordx4 per lane and
// instruction, whatever the row's start), with the next row's pieces in flight
// while the current one is reduced with DPP (max, then sum of exp): 8 bytes of
// HBM traffic per entry, the algorithmic minimum.  See the kernel's comment.
#include <type_traits>
#include <utility>

#include "common.h"
#include "options.h"
#include "wave_utils.h"

namespace sputnik_hip {
namespace {

constexpr int kBlock = 256;

using f4 = float __attribute__((ext_vector_type(4)));

template <typename F, int... Is>
__device__ __forceinline__ void unrolled_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// f(integral_constant<0>), ..., f(integral_constant<N-1>): register arrays indexed
// by the counter stay in registers.
template <int N, typename F>
__device__ __forceinline__ void unrolled(F&& f) {
  unrolled_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Storage types.  float is the reference's (src/softmax_cuda.cu:38-42); _Float16 and
// __bf16 are the native half-precision forms (round 3): read and written as such --
// half the HBM traffic -- and widened / narrowed (round to nearest even) in
// registers; all arithmetic stays float.
using h8 = _Float16 __attribute__((ext_vector_type(8)));
using b8 = __bf16 __attribute__((ext_vector_type(8)));
using f8 = float __attribute__((ext_vector_type(8)));
// A piece is 16 bytes of storage whatever the type (the bytes a lane has in flight
// per load decide the bandwidth of a latency-bound stream: 8-byte pieces of half
// values ran at HALF the byte rate of the float kernel): 4 float or 8 half entries.
template <typename T> struct Piece { using raw = f4; using wide = f4; static constexpr int kEntries = 4; };
template <> struct Piece<_Float16> { using raw = h8; using wide = f8; static constexpr int kEntries = 8; };
template <> struct Piece<__bf16> { using raw = b8; using wide = f8; static constexpr int kEntries = 8; };

// One piece at `base + byte_offset`: wave-uniform base (SGPR pair) plus a 32-bit
// per-lane offset, the cheapest addressing form (no 64-bit VALU arithmetic).
template <typename T, bool NT>
__device__ __forceinline__ typename Piece<T>::raw load_piece(const char* __restrict__ base,
                                                             unsigned byte_offset) {
  using P = typename Piece<T>::raw;
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const P*>(base + byte_offset));
  return *reinterpret_cast<const P*>(base + byte_offset);
}
// (the rows in flight stay in storage form: half the registers of the widened one)
template <typename T>
__device__ __forceinline__ typename Piece<T>::wide widen_piece(typename Piece<T>::raw raw) {
  if constexpr (std::is_same_v<T, float>) return raw;
  else return __builtin_convertvector(raw, typename Piece<T>::wide);
}
template <typename T, bool NT>
__device__ __forceinline__ void store_piece(T* __restrict__ dst, typename Piece<T>::wide v) {
  using P = typename Piece<T>::raw;
  P raw;
  if constexpr (std::is_same_v<T, float>) raw = v;
  else raw = __builtin_convertvector(v, P);
  if constexpr (NT) __builtin_nontemporal_store(raw, reinterpret_cast<P*>(dst));
  else *reinterpret_cast<P*>(dst) = raw;
}
template <typename T>
__device__ __forceinline__ float widen(T v) { return static_cast<float>(v); }

// Forward (BACKWARD = false): y = softmax(scale * x) over the stored entries of
// each row.  Backward: dx = scale * y * (dy - sum_row(dy * y)).
//
// A group of LPR lanes owns `rows_per_group` CONSECUTIVE rows (neighbouring
// groups stream neighbouring memory) and walks them with the next row's data
// already in flight.  A row moves through registers exactly once -- 8 bytes of
// HBM traffic per entry forward, 12 backward, the algorithmic minimum -- as
// 16-byte pieces that are ALIGNED in memory: the first piece starts up to three
// entries before the row, lane l owns pieces l, l + LPR, ... (V of them), so every
// load is one unconditional `dwordx4` per lane (a piece outside the row is
// clamped into the buffer and masked by index arithmetic, one unsigned compare
// per entry) and every interior store is one `dwordx4`; only the two pieces a
// row shares with its neighbours are stored entry by entry.  All index
// arithmetic is 32-bit and relative to the replica's base; the loop has no
// divergent branch.  (Half types: a piece is 8 entries, read "4" as "8" above.)
// Rows that do not fit the window (LPR * 4 * V - 3 entries),
// rows that touch the partial first / last piece of the whole buffer, and
// buffers whose inputs and output are aligned differently take three strided
// passes (re-reads served by L2).  exp is the hardware v_exp_f32 path (__expf):
// relative error about 5e-6 at |x - max| ~ 88, far inside the 1e-4 budget; the
// accurate library expf made the kernel VALU-bound.
template <typename T, int LPR, int BASE, int V, bool BACKWARD, int DEPTH, int NT>
__global__ __launch_bounds__(kBlock) void sparse_softmax_rows_kernel(
    int m, int rows_per_group, int nonzeros, const T* __restrict__ a, int64_t a_stride,
    const T* __restrict__ b, int64_t b_stride, const int* __restrict__ row_offsets,
    T* __restrict__ out, int64_t out_stride, float scale, int same_phase, int mask_heads,
    int first_replica, unsigned long long mask_members) {
  // "many mask", one launch per CLASS of masks (softmax_many_mask below): bit i set = mask i
  // runs at this launch's lane count; the workgroups of the others leave at once
  if (mask_members != 0 &&
      !((mask_members >> ((first_replica + static_cast<int>(blockIdx.y)) / mask_heads)) & 1ull))
    return;
  constexpr int E = Piece<T>::kEntries;   // entries per 16-byte piece
  using W = typename Piece<T>::wide;
  constexpr int kWindow = LPR * E * V;
  const int l = threadIdx.x % LPR;
  const int group = (blockIdx.x * kBlock + threadIdx.x) / LPR;
  const int replica = blockIdx.y;
  const int replicas = gridDim.y;
  // "many mask": the topology of replica r is number r / mask_heads of the
  // concatenated ones (row offsets [masks][m + 1], each zero based)
  if (mask_heads > 0) row_offsets += static_cast<int64_t>((first_replica + replica) / mask_heads) * (m + 1);
  a += replica * a_stride;
  if constexpr (BACKWARD) b += replica * b_stride;
  out += replica * out_stride;

  // This replica's alignment: its element e sits (e + phase) % E entries into a
  // 16-byte piece.  [qmin, qmax]: first / last piece start that lies wholly
  // inside the buffer (relative to this replica: the buffer begins `replica`
  // strides earlier and ends `replicas - 1 - replica` strides + nonzeros later).
  const int phase = static_cast<int>((reinterpret_cast<uintptr_t>(a) / sizeof(T)) % E);
  const int64_t before = static_cast<int64_t>(replica) * a_stride;
  const int64_t after = static_cast<int64_t>(replicas - 1 - replica) * a_stride + nonzeros;
  const int lo_lim = -static_cast<int>(before < (1 << 30) ? before : (1 << 30));
  const int hi_lim = static_cast<int>(after < (1 << 30) ? after : (1 << 30));
  const int qmin = lo_lim + ((-(lo_lim + phase)) & (E - 1));
  const int qmax = ((hi_lim - E + phase) & ~(E - 1)) - phase;
  // (pointers one piece before the replica: byte offsets (q + E) * sizeof(T) are never negative)
  const char* a_bytes = reinterpret_cast<const char*>(a - E);
  const char* b_bytes = reinterpret_cast<const char*>(b - E);

  // A wave's 64 / LPR groups take ADJACENT rows in every step (row = first row of
  // the wave + step * groups + group), so one store instruction of the wave
  // covers one contiguous stretch of memory and the 16-byte pieces that two
  // neighbouring rows share are written within the same instruction pair.
  // Lanes of groups past the last row keep empty rows so that every lane reaches
  // the convergent reductions below.
  constexpr int kGroups = kWave / LPR;  // per wave
  const int wave_index = (blockIdx.x * kBlock + threadIdx.x) / kWave;
  const int g = (threadIdx.x % kWave) / LPR;
  const int row0 = wave_index * (rows_per_group * kGroups) + g;
  (void)group;

  struct Row {
    int p0, len, s;   // first entry, length, start of the first aligned piece
    bool fast;
    bool extra;       // wave-uniform: some row of this step reaches past BASE pieces
    typename Piece<T>::raw x[V], y[V];
  };
  // The run's row bounds, loaded ONCE (lane i of the group holds those of its
  // i-th row): inside the loop a row's bounds come from a lane broadcast, so no
  // load of the loop depends on another one (a dependent load would also wait
  // for the previous row's stores: vmcnt retires in order).
  const int mine = row0 + l * kGroups;
  const bool have = l < rows_per_group && mine < m;
  const int o_lo = have ? row_offsets[mine] : 0;
  const int o_hi = have ? row_offsets[mine + 1] : 0;
  auto fetch = [&](int it, Row& w) {
    const int p0 = it < rows_per_group ? __shfl(o_lo, it, LPR) : 0;
    const int p1 = it < rows_per_group ? __shfl(o_hi, it, LPR) : 0;
    w.p0 = p0;
    w.len = p1 - p0;
    w.s = p0 - ((p0 + phase) & (E - 1));
    const int last_piece = ((p1 - 1 + phase) & ~(E - 1)) - phase;
    w.fast = same_phase && p1 - w.s <= kWindow && w.s >= qmin && last_piece <= qmax;
    // Pieces BASE .. V-1 exist for the few rows that are much longer than the
    // mean: they are requested and worked on only in the steps in which some row
    // of the wave needs them (a wave-uniform branch: no exp, no load and no store
    // is spent on them otherwise).
    w.extra = BASE < V && __builtin_amdgcn_ballot_w64(w.fast && p1 - w.s > BASE * E * LPR) != 0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (v >= BASE && !w.extra) continue;
      const int q = w.s + E * l + v * (E * LPR);
      const unsigned off =
          static_cast<unsigned>(min(max(q, qmin), qmax) + E) * static_cast<unsigned>(sizeof(T));
      w.x[v] = load_piece<T, (NT & 1) != 0>(a_bytes, off);
      if constexpr (BACKWARD) w.y[v] = load_piece<T, (NT & 1) != 0>(b_bytes, off);
    }
  };

  // DEPTH rows in flight ahead of the one being reduced: a statically indexed
  // ring of DEPTH + 1 register sets (the loop body is unrolled once per set).
  Row ring[DEPTH + 1];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, ring[d]);
  auto process = [&](Row& row) {
    const int p0 = row.p0, len = row.len;
    if (row.fast) {
      struct {
        int s;
        bool extra;
        W x[V], y[V];
      } cur;
      cur.s = row.s;
      cur.extra = row.extra;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (v >= BASE && !cur.extra) continue;
        cur.x[v] = widen_piece<T>(row.x[v]);
        if constexpr (BACKWARD) cur.y[v] = widen_piece<T>(row.y[v]);
      }
      // entry i of piece v belongs to the row iff 0 <= q + i - p0 < len
      bool valid[V][E];
      W res[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (v >= BASE && !cur.extra) continue;
        const int t = cur.s + E * l + v * (E * LPR) - p0;
#pragma unroll
        for (int i = 0; i < E; ++i)
          valid[v][i] = static_cast<unsigned>(t + i) < static_cast<unsigned>(len);
      }
      if constexpr (!BACKWARD) {
        float mx = -INFINITY;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (v >= BASE && !cur.extra) continue;
#pragma unroll
          for (int i = 0; i < E; ++i) {
            cur.x[v][i] = valid[v][i] ? cur.x[v][i] * scale : -INFINITY;
            mx = fmaxf(mx, cur.x[v][i]);
          }
        }
        mx = group_max<LPR>(mx);
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (v >= BASE && !cur.extra) continue;
#pragma unroll
          for (int i = 0; i < E; ++i) {
            cur.x[v][i] = __expf(cur.x[v][i] - mx);   // exp(-inf) = 0 for the masked slots
            sum += cur.x[v][i];
          }
        }
        sum = group_sum<LPR>(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);   // 1 ulp
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (v < BASE || cur.extra) res[v] = cur.x[v] * inv;
      } else {
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (v >= BASE && !cur.extra) continue;
#pragma unroll
          for (int i = 0; i < E; ++i) dot += valid[v][i] ? cur.x[v][i] * cur.y[v][i] : 0.f;
        }
        dot = group_sum<LPR>(dot);
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (v < BASE || cur.extra) res[v] = scale * cur.x[v] * (cur.y[v] - dot);
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (v >= BASE && !cur.extra) continue;
        const int q = cur.s + E * l + v * (E * LPR);
        if (valid[v][0] && valid[v][E - 1]) {
          store_piece<T, (NT & 2) != 0>(out + q, res[v]);
        } else {
#pragma unroll
          for (int i = 0; i < E; ++i)
            if (valid[v][i]) out[q + i] = static_cast<T>(res[v][i]);
        }
      }
    } else if (len > 0) {
      // three strided passes (this group's lanes only)
      const int p1 = p0 + len;
      if constexpr (!BACKWARD) {
        float mx = -INFINITY;
        for (int q = p0 + l; q < p1; q += LPR) mx = fmaxf(mx, widen(a[q]) * scale);
        mx = group_max<LPR>(mx);
        float sum = 0.f;
        for (int q = p0 + l; q < p1; q += LPR) sum += __expf(widen(a[q]) * scale - mx);
        sum = group_sum<LPR>(sum);
        const float inv = 1.f / sum;
        for (int q = p0 + l; q < p1; q += LPR)
          out[q] = static_cast<T>(__expf(widen(a[q]) * scale - mx) * inv);
      } else {
        float dot = 0.f;
        for (int q = p0 + l; q < p1; q += LPR) dot = fmaf(widen(a[q]), widen(b[q]), dot);
        dot = group_sum<LPR>(dot);
        for (int q = p0 + l; q < p1; q += LPR)
          out[q] = static_cast<T>(scale * widen(a[q]) * (widen(b[q]) - dot));
      }
    }
  };
  for (int it0 = 0; it0 < rows_per_group; it0 += DEPTH + 1) {
    unrolled<DEPTH + 1>([&](auto j) {
      constexpr int J = decltype(j)::value;
      if (it0 + J < rows_per_group) {
        fetch(it0 + J + DEPTH, ring[(J + DEPTH) % (DEPTH + 1)]);  // past the run: an empty row
        process(ring[J]);
      }
    });
  }
}

template <typename T>
inline int phase_of(const T* p) {
  // alignment class of the first replica (the kernel recomputes it per replica)
  return static_cast<int>((reinterpret_cast<uintptr_t>(p) / sizeof(T)) % Piece<T>::kEntries);
}

template <typename T, int LPR, int BASE, int V, bool BACKWARD>
int launch_rows(int m, int nonzeros, int replicas, const T* a, int64_t a_stride, const T* b,
                int64_t b_stride, const int* row_offsets, T* out, int64_t out_stride,
                float scale, hipStream_t stream, int mask_heads, unsigned long long mask_members) {
  // Two rows per group once the grid fills the chip (256 CUs x 8 workgroups):
  // measured at config 3's mask with 64 and 512 replicas (tools/softmax_sweep.sh),
  // short workgroups whose dispatch staggers the read and the write phases beat
  // longer runs per group (1 / 2 / 4 / 16 rows: 104 / 99 / 102 / 110 us at 512).
  const int groups_per_block = kBlock / LPR;
  const int64_t total_rows = static_cast<int64_t>(m) * replicas;
  int rows_per_group = static_cast<int>(total_rows / (int64_t{256} * 8 * groups_per_block));
  rows_per_group = max(1, min(rows_per_group, 2));
  if (options().softmax_rpg > 0) rows_per_group = min(options().softmax_rpg, 16);  // developer knob
  const int depth = options().softmax_depth;                                       // developer knob
  const int gx = ceil_div(ceil_div(m, rows_per_group), groups_per_block);
  // The fast path needs the inputs and the output of every replica aligned alike
  // and 32-bit byte offsets.  Its window of clamped 16-byte pieces is derived from
  // ONE stride and applied to every operand, so the strides must be EQUAL (fresh
  // allocations: the common case), not merely congruent: with a broadcast gradient
  // (stride 0) or a padded input the unconditional loads of the other operand would
  // reach up to a window outside its buffer.
  const bool strides_alike = a_stride == out_stride && (!BACKWARD || a_stride == b_stride);
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    const T* a_r = a + r0 * a_stride;
    const T* b_r = BACKWARD ? b + r0 * b_stride : nullptr;
    T* out_r = out + r0 * out_stride;
    // (half types: every replica must also start on an element boundary of its piece
    // grid, i.e. the stride keeps the phase arithmetic in whole elements: always true)
    const int same_phase = strides_alike && nonzeros >= 2 * Piece<T>::kEntries && nonzeros < (1 << 28) &&
                           phase_of(a_r) == phase_of(out_r) &&
                           (!BACKWARD || phase_of(a_r) == phase_of(b_r));
#define SPUTNIK_HIP_SOFTMAX_LAUNCH(DEPTH, NT)                                                     \
  hipLaunchKernelGGL((sparse_softmax_rows_kernel<T, LPR, BASE, V, BACKWARD, DEPTH, NT>), dim3(gx, ry), \
                     dim3(kBlock), 0, stream, m, rows_per_group, nonzeros, a_r, a_stride, b_r,   \
                     b_stride, row_offsets, out_r, out_stride, scale, same_phase, mask_heads, r0, \
                     mask_members)
    // Nontemporal STORES in the forward pass when the output is larger than the
    // caches can hand to the next kernel anyway (measured, 1024^2 mask at density
    // 0.1: 512 replicas, 215 MB out, 88.4 -> 75.8 us; 64 replicas 12.5 -> 12.2 us
    // alone but +8 us on the attention step that reads the 27 MB straight back).
    // In the backward pass they cost 4-5 %, nontemporal loads 15-35 % in both.
    // Developer knob: 0 = none, 1 = loads, 2 = stores, 3 = both.
    const bool large_out =
        static_cast<int64_t>(ry) * nonzeros * static_cast<int64_t>(sizeof(T)) >= (int64_t{128} << 20);
    const int nt = options().softmax_nt >= 0 ? options().softmax_nt
                                            : (!BACKWARD && large_out) ? 2 : 0;
    if (depth == 3) SPUTNIK_HIP_SOFTMAX_LAUNCH(3, 0);
    else if (depth == 2) SPUTNIK_HIP_SOFTMAX_LAUNCH(2, 0);
    else if (nt == 1) SPUTNIK_HIP_SOFTMAX_LAUNCH(1, 1);
    else if (nt == 2) SPUTNIK_HIP_SOFTMAX_LAUNCH(1, 2);
    else if (nt == 3) SPUTNIK_HIP_SOFTMAX_LAUNCH(1, 3);
    else SPUTNIK_HIP_SOFTMAX_LAUNCH(1, 0);
#undef SPUTNIK_HIP_SOFTMAX_LAUNCH
    const int st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

// BASE pieces of four entries per lane cover the typical row (mean + ~2.2
// standard deviations of a random pattern + the alignment slack: every register
// slot costs an exp); up to V pieces are there for the few longer rows and are
// paid for only in the steps that need them.  Rows beyond LPR * 4 * V take the
// strided passes.
// Entries the window of a row has to hold for a mask of `nonzeros` entries in m rows: the
// mean row + ~2.2 standard deviations of a random pattern + the alignment slack.
inline int64_t entries_a_row_needs(int m, int64_t nonzeros) {
  const int64_t mean = nonzeros / m;
  int64_t dev = 1;
  while (dev * dev < 5 * mean) ++dev;   // ~ 2.2 * sqrt(mean)
  return mean + dev + 3;
}
// The float kernels' classes by that need (the cases of dispatch_rows below).
inline int float_class_of(int64_t need) {
  return need <= 64 ? 0 : need <= 128 ? 1 : need <= 192 ? 2 : need <= 256 ? 3 : need <= 512 ? 4 : 5;
}

template <typename T, bool BACKWARD>
int dispatch_rows(int m, int nonzeros, int replicas, const T* a, int64_t a_stride,
                  const T* b, int64_t b_stride, const int* row_offsets, T* out,
                  int64_t out_stride, float scale, hipStream_t stream, int mask_heads = 0,
                  int typical_nonzeros = -1, unsigned long long mask_members = 0) {
  // (many masks: `nonzeros` is the width of a value row -- what may be read --, the
  // window is sized for `typical_nonzeros`, the largest mask of the launch)
  const int64_t need = entries_a_row_needs(m, typical_nonzeros >= 0 ? typical_nonzeros : nonzeros);
#define SPUTNIK_HIP_SOFTMAX_CASE(LPR, BASE, V)                                                    \
  return launch_rows<T, LPR, BASE, V, BACKWARD>(m, nonzeros, replicas, a, a_stride, b, b_stride,  \
                                                row_offsets, out, out_stride, scale, stream,      \
                                                mask_heads, mask_members)
  if constexpr (Piece<T>::kEntries == 8) {   // half types: 8 entries per piece and lane
    if (need <= 128) SPUTNIK_HIP_SOFTMAX_CASE(16, 1, 2);
    if (need <= 256) SPUTNIK_HIP_SOFTMAX_CASE(16, 2, 3);
    if (need <= 512) SPUTNIK_HIP_SOFTMAX_CASE(32, 2, 3);
    SPUTNIK_HIP_SOFTMAX_CASE(64, 2, 2);
  } else {
    if (need <= 64) SPUTNIK_HIP_SOFTMAX_CASE(16, 1, 2);
    if (need <= 128) SPUTNIK_HIP_SOFTMAX_CASE(16, 2, 3);
    if (need <= 192) SPUTNIK_HIP_SOFTMAX_CASE(16, 3, 4);
    if (need <= 256) SPUTNIK_HIP_SOFTMAX_CASE(32, 2, 4);
    if (need <= 512) SPUTNIK_HIP_SOFTMAX_CASE(32, 4, 4);
    SPUTNIK_HIP_SOFTMAX_CASE(64, 4, 4);
  }
#undef SPUTNIK_HIP_SOFTMAX_CASE
}

// Storage type dispatch (SPUTNIK_HIP_F32 / F16 / BF16 of include/sputnik_hip.h): every
// value operand of one call has the same type.
template <bool BACKWARD>
int dispatch_typed(int dtype, int m, int nonzeros, int replicas, const void* a, int64_t a_stride,
                   const void* b, int64_t b_stride, const int* row_offsets, void* out,
                   int64_t out_stride, float scale, hipStream_t stream, int mask_heads = 0,
                   int typical_nonzeros = -1) {
#define SPUTNIK_HIP_SOFTMAX_T(T)                                                                  \
  return dispatch_rows<T, BACKWARD>(m, nonzeros, replicas, static_cast<const T*>(a), a_stride,    \
                                    static_cast<const T*>(b), b_stride, row_offsets,              \
                                    static_cast<T*>(out), out_stride, scale, stream, mask_heads,  \
                                    typical_nonzeros)
  switch (dtype) {
    case SPUTNIK_HIP_F32:
      if (!aligned_to(a, 4) || !aligned_to(out, 4) || (BACKWARD && !aligned_to(b, 4)))
        return SPUTNIK_HIP_INVALID_ARGUMENT;
      SPUTNIK_HIP_SOFTMAX_T(float);
    case SPUTNIK_HIP_F16:
      if (!aligned_to(a, 2) || !aligned_to(out, 2) || (BACKWARD && !aligned_to(b, 2)))
        return SPUTNIK_HIP_INVALID_ARGUMENT;
      SPUTNIK_HIP_SOFTMAX_T(_Float16);
    case SPUTNIK_HIP_BF16:
      if (!aligned_to(a, 2) || !aligned_to(out, 2) || (BACKWARD && !aligned_to(b, 2)))
        return SPUTNIK_HIP_INVALID_ARGUMENT;
      SPUTNIK_HIP_SOFTMAX_T(__bf16);
    default: return SPUTNIK_HIP_INVALID_ARGUMENT;
  }
#undef SPUTNIK_HIP_SOFTMAX_T
}

}  // namespace

// One launch for all masks of a "many mask" batch (many_mask.hip): value rows
// [replicas][width], replica r under the topology number r / heads.
// Round 5: one launch per CLASS of masks instead of one launch at the lane count of the
// largest.  The masks of a batch differ in size (tests/test_attention_many_masks.py:26-36 of
// the reference draws a sparsity per batch element), and a row of 51 entries walked by the 64
// lanes x 4 pieces that a row of 512 needs keeps one piece in sixteen busy: 8 masks of
// density 0.05 - 0.5 x 8 heads, S = 1024, ran at 36 us where the same entries in equal masks
// take 22.  Every launch covers all replicas; the workgroups of the masks outside its class
// (a bit per mask in a kernel argument: up to 64 masks, more keep the one launch) leave at
// once.  `mask_nonzeros`: the masks' entry counts, a HOST array.
int softmax_many_mask(bool backward, int m, int width, int largest_nonzeros, int replicas, int heads,
                      const float* a, int64_t a_stride, const float* b, int64_t b_stride,
                      const int* row_offsets, float* out, int64_t out_stride, float scale,
                      hipStream_t stream, int masks, const int* mask_nonzeros) {
  if (m == 0 || width == 0 || replicas == 0 || largest_nonzeros == 0) return 0;
  const auto launch = [&](int typical, unsigned long long members) {
    return backward ? dispatch_rows<float, true>(m, width, replicas, a, a_stride, b, b_stride,
                                                 row_offsets, out, out_stride, scale, stream, heads,
                                                 typical, members)
                    : dispatch_rows<float, false>(m, width, replicas, a, a_stride, nullptr, 0,
                                                  row_offsets, out, out_stride, scale, stream, heads,
                                                  typical, members);
  };
  if (mask_nonzeros == nullptr || masks < 2 || masks > 64 || replicas > kMaxGridYZ)
    return launch(largest_nonzeros, 0ull);
  constexpr int kClasses = 6;
  unsigned long long members[kClasses] = {};
  int typical[kClasses] = {};
  for (int i = 0; i < masks; ++i) {
    if (mask_nonzeros[i] <= 0) continue;   // (no entries: nothing to normalise)
    const int c = float_class_of(entries_a_row_needs(m, mask_nonzeros[i]));
    members[c] |= 1ull << i;
    typical[c] = std::max(typical[c], mask_nonzeros[i]);
  }
  for (int c = 0; c < kClasses; ++c) {
    if (members[c] == 0) continue;
    const int st = launch(typical[c], members[c]);
    if (st != 0) return st;
  }
  return 0;
}

}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

int sputnik_hip_sparse_softmax_typed(int m, int n, int nonzeros, int replicas, const void* values,
                                     int64_t values_stride, const int* row_indices,
                                     const int* row_offsets, const int* column_indices,
                                     float scale, void* out, int64_t out_stride, int dtype,
                                     sputnik_hip_stream_t stream) {
  (void)n;
  (void)row_indices;
  (void)column_indices;
  if (m < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (dtype != SPUTNIK_HIP_F32 && dtype != SPUTNIK_HIP_F16 && dtype != SPUTNIK_HIP_BF16)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || replicas == 0) return 0;
  return dispatch_typed<false>(dtype, m, nonzeros, replicas, values, values_stride, nullptr, 0,
                               row_offsets, out, out_stride, scale, stream);
}

int sputnik_hip_sparse_softmax_backward_typed(int m, int nonzeros, int replicas,
                                              const void* softmax_out, int64_t out_stride,
                                              const void* grad_out, int64_t grad_out_stride,
                                              const int* row_offsets, float scale,
                                              void* grad_values, int64_t grad_values_stride,
                                              int dtype, sputnik_hip_stream_t stream) {
  if (m < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (dtype != SPUTNIK_HIP_F32 && dtype != SPUTNIK_HIP_F16 && dtype != SPUTNIK_HIP_BF16)
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || nonzeros == 0 || replicas == 0) return 0;
  return dispatch_typed<true>(dtype, m, nonzeros, replicas, softmax_out, out_stride, grad_out,
                              grad_out_stride, row_offsets, grad_values, grad_values_stride, scale,
                              stream);
}

int sputnik_hip_sparse_softmax_scaled_batched(int m, int n, int nonzeros, int replicas,
                                              const float* values, int64_t values_stride,
                                              const int* row_indices, const int* row_offsets,
                                              const int* column_indices, float scale, float* out,
                                              int64_t out_stride, sputnik_hip_stream_t stream) {
  return sputnik_hip_sparse_softmax_typed(m, n, nonzeros, replicas, values, values_stride,
                                          row_indices, row_offsets, column_indices, scale, out,
                                          out_stride, SPUTNIK_HIP_F32, stream);
}

int sputnik_hip_sparse_softmax_batched(int m, int n, int nonzeros, int replicas,
                                       const float* values, int64_t values_stride,
                                       const int* row_indices, const int* row_offsets,
                                       const int* column_indices, float* out,
                                       int64_t out_stride, sputnik_hip_stream_t stream) {
  return sputnik_hip_sparse_softmax_scaled_batched(m, n, nonzeros, replicas, values,
                                                   values_stride, row_indices, row_offsets,
                                                   column_indices, 1.0f, out, out_stride, stream);
}

int sputnik_hip_sparse_softmax_backward_batched(int m, int nonzeros, int replicas,
                                                const float* softmax_out, int64_t out_stride,
                                                const float* grad_out, int64_t grad_out_stride,
                                                const int* row_offsets, float scale,
                                                float* grad_values, int64_t grad_values_stride,
                                                sputnik_hip_stream_t stream) {
  return sputnik_hip_sparse_softmax_backward_typed(m, nonzeros, replicas, softmax_out, out_stride,
                                                   grad_out, grad_out_stride, row_offsets, scale,
                                                   grad_values, grad_values_stride,
                                                   SPUTNIK_HIP_F32, stream);
}

int sputnik_hip_sparse_softmax(int m, int n, int nonzeros, const float* values,
                               const int* row_indices, const int* row_offsets,
                               const int* column_indices, float* out,
                               sputnik_hip_stream_t stream) {
  return sputnik_hip_sparse_softmax_batched(m, n, nonzeros, 1, values, 0, row_indices,
                                            row_offsets, column_indices, out, 0, stream);
}

}  // extern "C"
