// Fused sparse attention forward for gfx950:
//
//   out[i, :] = sum_j softmax_j(scale * <q_i, k_j>) * v_j      j over the stored
//                                                               columns of mask row i
//
// One kernel for the chain the reference runs as three library calls and an
// elementwise pass -- sddmm, "/ sqrt(d)", sparse_softmax, spmm
// (modules/sparse_attention.py:66-82) -- so the [replicas, nnz] score and
// weight arrays never exist: HBM traffic is Q, K, V and the output only.
//
// Decomposition (the 64-column SpMM kernel's, spmm_tiled64.hip): a workgroup
// owns 128 query rows and walks the key/value rows in chunks of 128; each
// chunk's K rows and V rows are staged into LDS by direct global->LDS copies
// (double buffered).  A 16-lane row group owns one query row: its q fragment
// (pre-multiplied by the scale), running maximum, running sum and 4 output
// columns per lane stay in registers for the whole walk (online softmax).
// Per 16-entry window of a row's columns inside the chunk:
//   1. 16 partial dot products per lane against the K rows (ds_read_b128 of a
//      DPP-broadcast address), then ONE transposing DPP reduction that leaves
//      score u in lane u;
//   2. window maximum and sum with two 16-lane DPP all-reduces, one exp per
//      lane, rescale of the accumulators;
//   3. the weights go back out to all lanes entry by entry, paired with the
//      tile offset in one 64-bit DPP broadcast, against the V rows (same
//      offsets: the V tile sits at a fixed distance from the K tile).
// Needs ascending columns inside rows (checked by the shared pre-pass); row
// blocks that fail take an order-independent path (K, V gathered from L2).
//
// Round 3: step 1 in the QUAD form of sddmm_tiled.hip -- the four quads of a row group
// work on four different entries, lane (quad q, t) holds a quarter of the scaled q row
// (16 elements, chunk order rotated by q: no LDS bank conflict) and two quad_perm adds
// close a dot product; lane (q, t) ends up with the score of entry 4t + q, which is
// where that entry's column lives.  The kernel was 80 % busy issuing vector
// instructions (profiles/r3e_pmc_sq_attention_ops.json: 46.5 M per launch), a third
// of them the broadcasts and the 45-instruction transposing reduction of the 16-lane
// form.  Steps 2 and 3 are unchanged (step 3 finds entry u in lane 4 (u % 4) + u / 4).
#include "spmm_tiled_common.h"

namespace sputnik_hip {
namespace {

using namespace tiled;

constexpr int kD = 64;      // head dimension served by this kernel
constexpr int kWaves = 16;  // waves per workgroup
constexpr int kRQ = 2;      // row quads per wave (4 query rows each)
constexpr int kBK = 128;    // key/value rows per LDS stage
constexpr int kBM = kWaves * kRQ * 4;
constexpr int kThreads = kWaves * kWave;
constexpr int kTileFloats = kBK * kD;  // one of K / V: 32 KiB
constexpr int kWin = 2;                // 16-entry windows prefetched per row and chunk
constexpr int kCopiesPerWave = (kBK / 4) / kWaves;  // 1 KiB copies of 4 rows each
static_assert((kBK / 4) % kWaves == 0, "stage copies split evenly over the waves");

__device__ __forceinline__ void stage_kv(float* __restrict__ tile, const float* __restrict__ k,
                                         const float* __restrict__ v, int n, int jc, int wave,
                                         int lane) {
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int j = 0; j < kCopiesPerWave; ++j) {
    const int r0 = (wave + j * kWaves) * 4;
    const int src_row = min(jc + r0 + g, n - 1);  // past the last key: re-read the last row
    const unsigned off = (static_cast<unsigned>(src_row) * kD + i * 4u) * 4u;
    lds_dma_row(k, off, tile + r0 * kD);
    lds_dma_row(v, off, tile + kTileFloats + r0 * kD);
  }
}

using f4v = float __attribute__((ext_vector_type(4)));
using v2f = float __attribute__((ext_vector_type(2)));

struct RowAcc {
  f4v q[4];    // scale * q, elements 16t + 4((c + quad) % 4) .. +3 for c = 0..3 (lane = (quad, t))
  float4 acc;  // unnormalised output columns 4i .. 4i+3
  float mx, l;
};

template <int S>
__device__ __forceinline__ int quad_bcast_add(int v, int add) {
  // add + (v of lane S of the quad): v_add_u32_dpp quad_perm:[S,S,S,S]
  return __builtin_amdgcn_update_dpp(0, v, S * 0x55, 0xF, 0xF, true) + add;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  float s = a.x * b.x;
  s = fmaf(a.y, b.y, s);
  s = fmaf(a.z, b.z, s);
  return fmaf(a.w, b.w, s);
}

// Online-softmax update of one row with `e`-weighted V contributions still to
// be added by the caller: returns the factor the old accumulators were scaled by.
__device__ __forceinline__ void rescale(RowAcc& r, float m_new) {
  const float alpha = __expf(r.mx - m_new);  // mx = -inf gives 0
  r.l *= alpha;
  r.acc.x *= alpha;
  r.acc.y *= alpha;
  r.acc.z *= alpha;
  r.acc.w *= alpha;
  r.mx = m_new;
}

__global__ __launch_bounds__(kThreads) void sparse_attention_kernel(
    int m, int n, int nonzeros, int slots, int nchunks, const int* __restrict__ row_indices,
    const int* __restrict__ row_offsets, const int* __restrict__ column_indices,
    const int* __restrict__ table, const int* __restrict__ row_ok, const float* __restrict__ q,
    int64_t q_stride, const float* __restrict__ k, int64_t k_stride, const float* __restrict__ v,
    int64_t v_stride, float scale, float* __restrict__ out, int64_t out_stride,
    float* __restrict__ lse, int64_t lse_stride) {
  __shared__ float tile[2][2 * kTileFloats];  // [buffer][K rows | V rows]

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int g = lane >> 4, i = lane & 15;
  const int qd = i >> 2, tq = i & 3;   // quad of the row group, lane of the quad
  const int e16 = 4 * tq + qd;         // this lane's entry of a 16-entry window
  // (the row blocks of a replica read the same K and V: one XCD, see xcd_local_index)
  const unsigned long long work = xcd_local_index();
  const int mblock = static_cast<int>(work % gridDim.x);
  const int replica = static_cast<int>(work / gridDim.x);
  q += replica * q_stride;
  k += replica * k_stride;
  v += replica * v_stride;
  out += replica * out_stride;
  if (lse != nullptr) lse += replica * lse_stride;
  const int slot0 = mblock * kBM + wave * (kRQ * 4);
  const int last = nonzeros - 1;

  RowAcc st[kRQ];
  int my_row[kRQ];
#pragma unroll
  for (int t = 0; t < kRQ; ++t) {
    const int entry = dealt_index(slot0 + 4 * t + g, slots, kBM);
    my_row[t] = entry < m ? row_indices[entry] : -1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f4v qf = {0.f, 0.f, 0.f, 0.f};
      if (my_row[t] >= 0)
        qf = *reinterpret_cast<const f4v*>(q + static_cast<int64_t>(my_row[t]) * kD + 16 * tq +
                                           4 * ((c + qd) & 3));
      st[t].q[c] = qf * scale;
    }
    st[t].acc = make_float4(0.f, 0.f, 0.f, 0.f);
    st[t].mx = -INFINITY;
    st[t].l = 0.f;
  }

  auto finish = [&]() {
#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      if (my_row[t] < 0) continue;
      const float inv = st[t].l > 0.f ? 1.f / st[t].l : 0.f;  // rows without entries give zeros
      *reinterpret_cast<float4*>(out + static_cast<int64_t>(my_row[t]) * kD + 4 * i) =
          make_float4(st[t].acc.x * inv, st[t].acc.y * inv, st[t].acc.z * inv,
                      st[t].acc.w * inv);
      if (lse != nullptr && i == 0)
        lse[my_row[t]] = st[t].l > 0.f ? st[t].mx + __logf(st[t].l) : -INFINITY;
    }
  };

  // Row blocks whose columns do not ascend inside rows: order-independent path,
  // one entry at a time, K and V rows gathered from global memory.
  if (!block_rows_ok(row_ok, mblock * kBM, kBM)) {
    for (int t = 0; t < kRQ; ++t) {
      const int p0 = my_row[t] >= 0 ? row_offsets[my_row[t]] : 0;
      const int p1 = my_row[t] >= 0 ? row_offsets[my_row[t] + 1] : 0;
      float4 q16 = make_float4(0.f, 0.f, 0.f, 0.f);   // elements 4i .. 4i+3 of scale * q
      if (my_row[t] >= 0) {
        const float4 qf =
            *reinterpret_cast<const float4*>(q + static_cast<int64_t>(my_row[t]) * kD + 4 * i);
        q16 = make_float4(qf.x * scale, qf.y * scale, qf.z * scale, qf.w * scale);
      }
      for (int p = p0; p < p1; ++p) {
        const int64_t base = static_cast<int64_t>(column_indices[p]) * kD + 4 * i;
        const float4 kf = *reinterpret_cast<const float4*>(k + base);
        const float4 vf = *reinterpret_cast<const float4*>(v + base);
        const float s = group_sum<16>(dot4(q16, kf));
        if (s > st[t].mx) rescale(st[t], s);
        const float e = __expf(s - st[t].mx);
        st[t].l += e;
        st[t].acc.x = fmaf(e, vf.x, st[t].acc.x);
        st[t].acc.y = fmaf(e, vf.y, st[t].acc.y);
        st[t].acc.z = fmaf(e, vf.z, st[t].acc.z);
        st[t].acc.w = fmaf(e, vf.w, st[t].acc.w);
      }
    }
    finish();
    return;
  }

  const int* __restrict__ my_table = table + slot0 + g;
  int ps[kRQ], pe[kRQ], wcol[kRQ][kWin];
#pragma unroll
  for (int t = 0; t < kRQ; ++t) {
    ps[t] = my_table[4 * t];
    pe[t] = my_table[slots + 4 * t];
#pragma unroll
    for (int w = 0; w < kWin; ++w) wcol[t][w] = column_indices[min(ps[t] + 16 * w + e16, last)];
  }

  stage_kv(tile[0], k, v, n, 0, wave, lane);
  wait_vm<0>();
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunks;
    if (more) stage_kv(tile[buf ^ 1], k, v, n, (c + 1) * kBK, wave, lane);

    int pe_next[kRQ], ncol[kRQ][kWin];
#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      pe_next[t] = more ? my_table[static_cast<int64_t>(c + 2) * slots + 4 * t] : pe[t];
#pragma unroll
      for (int w = 0; w < kWin; ++w)
        ncol[t][w] = more ? column_indices[min(pe[t] + 16 * w + e16, last)] : 0;
    }

    const char* __restrict__ k_base = reinterpret_cast<const char*>(&tile[buf][0] + i * 4);
    const char* __restrict__ v_base = k_base + kTileFloats * sizeof(float);
    const int jc = c * kBK;
    const int k_lds = static_cast<int>(static_cast<unsigned>(
        reinterpret_cast<uintptr_t>(AS_LDS(&tile[buf][0]))));   // LDS byte address of the K tile

#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      const int cnt = pe[t] - ps[t];  // this group's row; the same in its 16 lanes

      auto window = [&](int ecol, int w0) {
        const int left = cnt - w0;
        if (left <= 0) return;
        const bool valid = e16 < left;
        const int roff = valid ? ((ecol - jc) * (kD * 4)) : 0;

        // 1. scores, quad form: step S = entries 4S .. 4S+3, one per quad
        float s = 0.f;
        const int kq = k_lds + 64 * tq;   // this lane's quarter of tile row 0
        auto scores4 = [&](auto Sc) {
          constexpr int kS = decltype(Sc)::value;
          f4v b[4];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            b[c] = *reinterpret_cast<const __attribute__((address_space(3))) f4v*>(
                static_cast<unsigned>(quad_bcast_add<kS>(roff, kq + 16 * ((c + qd) & 3))));
          v2f a2 = {0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            a2 = __builtin_elementwise_fma(v2f{st[t].q[c].x, st[t].q[c].y}, v2f{b[c].x, b[c].y}, a2);
            a2 = __builtin_elementwise_fma(v2f{st[t].q[c].z, st[t].q[c].w}, v2f{b[c].z, b[c].w}, a2);
          }
          const float total = group_sum<4>(a2.x + a2.y);
          s = (tq == kS) ? total : s;
        };
        scores4(std::integral_constant<int, 0>{});
        if (left > 4) scores4(std::integral_constant<int, 1>{});
        if (left > 8) scores4(std::integral_constant<int, 2>{});
        if (left > 12) scores4(std::integral_constant<int, 3>{});
        s = valid ? s : -INFINITY;

        // 2. online softmax over the window (at least one entry is valid)
        const float m_new = fmaxf(st[t].mx, group_max<16>(s));
        rescale(st[t], m_new);
        const float e = valid ? __expf(s - m_new) : 0.f;
        st[t].l += group_sum<16>(e);

        // 3. weighted V rows; padded entries carry weight 0 and offset 0
        const entry_pair ent = make_entry(roff, e);
        float a4[4] = {st[t].acc.x, st[t].acc.y, st[t].acc.z, st[t].acc.w};
        auto values4 = [&](auto G) {
          constexpr int kG = decltype(G)::value;
          // (entries kG .. kG+3 of the window sit in lanes kG/4, 4 + kG/4, 8 + kG/4, 12 + kG/4)
          const entry_pair e0 = row_bcast_entry<0 + kG / 4>(ent), e1 = row_bcast_entry<4 + kG / 4>(ent);
          const entry_pair e2 = row_bcast_entry<8 + kG / 4>(ent), e3 = row_bcast_entry<12 + kG / 4>(ent);
          const float4 b0 = *reinterpret_cast<const float4*>(v_base + entry_off(e0));
          const float4 b1 = *reinterpret_cast<const float4*>(v_base + entry_off(e1));
          const float4 b2 = *reinterpret_cast<const float4*>(v_base + entry_off(e2));
          const float4 b3 = *reinterpret_cast<const float4*>(v_base + entry_off(e3));
          SPUTNIK_HIP_FMA4(a4, entry_val(e0), b0);
          SPUTNIK_HIP_FMA4(a4, entry_val(e1), b1);
          SPUTNIK_HIP_FMA4(a4, entry_val(e2), b2);
          SPUTNIK_HIP_FMA4(a4, entry_val(e3), b3);
        };
        values4(std::integral_constant<int, 0>{});
        if (left > 4) values4(std::integral_constant<int, 4>{});
        if (left > 8) values4(std::integral_constant<int, 8>{});
        if (left > 12) values4(std::integral_constant<int, 12>{});
        st[t].acc = make_float4(a4[0], a4[1], a4[2], a4[3]);
      };
#pragma unroll
      for (int w = 0; w < kWin; ++w) window(wcol[t][w], 16 * w);
      // more than 32 entries of one row inside one chunk: fetch on demand
      const int longest = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                              max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
      for (int w0 = 16 * kWin; w0 < longest; w0 += 16)
        window(column_indices[min(ps[t] + w0 + e16, last)], w0);
    }

#pragma unroll
    for (int t = 0; t < kRQ; ++t) {
      ps[t] = pe[t];
      pe[t] = pe_next[t];
#pragma unroll
      for (int w = 0; w < kWin; ++w) wcol[t][w] = ncol[t][w];
    }
    wait_vm<0>();     // the next K/V tiles have landed
    __syncthreads();  // ... for every wave, and the current buffer is free
  }
  finish();
}

inline int slots_of(int m) { return ceil_div(m, kBM) * kBM; }
inline int chunks_of(int n) { return ceil_div(n, kBK); }

bool supported(int m, int n, int d, int nonzeros) {
  return d == kD && m > 0 && n > 0 && nonzeros > 0 &&
         static_cast<int64_t>(n) * kD * 4 < (int64_t{1} << 32);
}

}  // namespace
}  // namespace sputnik_hip

using namespace sputnik_hip;

extern "C" {

int sputnik_hip_sparse_attention_supported(int m, int n, int d, int nonzeros) {
  return supported(m, n, d, nonzeros) ? 1 : 0;
}

size_t sputnik_hip_sparse_attention_workspace_bytes(int m, int n, int d, int nonzeros) {
  if (!supported(m, n, d, nonzeros)) return 0;
  return row_ok_bytes(slots_of(m)) +
         sizeof(int) * static_cast<size_t>(chunks_of(n) + 1) * slots_of(m);
}

namespace {

int attention_exec(int m, int n, int d, int nonzeros, int replicas, const int* row_indices,
                   const int* row_offsets, const int* column_indices, const float* q,
                   int64_t q_stride, const float* k, int64_t k_stride, const float* v,
                   int64_t v_stride, float scale, float* out, int64_t out_stride, float* lse,
                   int64_t lse_stride, void* workspace, size_t workspace_bytes, bool planned,
                   hipStream_t stream) {
  if (m < 0 || n < 0 || d < 0 || nonzeros < 0 || replicas < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (m == 0 || replicas == 0) return 0;
  if (nonzeros == 0 || n == 0) {  // every row is empty: zeros (and -inf log-sum-exp)
    for (int r = 0; r < replicas; ++r) {
      hipError_t e = hipMemsetAsync(out + r * out_stride, 0, sizeof(float) * m * d, stream);
      if (e != hipSuccess) return static_cast<int>(e);
      if (lse != nullptr) {
        e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(lse + r * lse_stride),
                              static_cast<int>(0xff800000u), m, stream);
        if (e != hipSuccess) return static_cast<int>(e);
      }
    }
    return 0;
  }
  if (!supported(m, n, d, nonzeros) || !aligned_to(q, 16) || !aligned_to(k, 16) ||
      !aligned_to(v, 16) || !aligned_to(out, 16) || q_stride % 4 != 0 || k_stride % 4 != 0 ||
      v_stride % 4 != 0 || out_stride % 4 != 0)
    return SPUTNIK_HIP_UNSUPPORTED;
  if (workspace == nullptr || !aligned_to(workspace, 16) ||
      workspace_bytes < sputnik_hip_sparse_attention_workspace_bytes(m, n, d, nonzeros))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  const int slots = slots_of(m), nchunks = chunks_of(n);
  int* row_ok = static_cast<int*>(workspace);
  int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + row_ok_bytes(slots));
  int st = 0;
  if (!planned) {
    hipLaunchKernelGGL((spmm_chunk_table_kernel<kBK>), dim3(ceil_div(slots, 4)),
                       dim3(256), 0, stream, m, n, slots, kBM, nchunks, row_indices, row_offsets,
                       column_indices, table, row_ok);
    st = launch_status();
    if (st != 0) return st;
  }
  for (int r0 = 0; r0 < replicas; r0 += kMaxGridYZ) {
    const int ry = min(replicas - r0, kMaxGridYZ);
    hipLaunchKernelGGL(sparse_attention_kernel, dim3(slots / kBM, ry), dim3(kThreads), 0, stream,
                       m, n, nonzeros, slots, nchunks, row_indices, row_offsets, column_indices,
                       table, row_ok, q + r0 * q_stride, q_stride, k + r0 * k_stride, k_stride,
                       v + r0 * v_stride, v_stride, scale, out + r0 * out_stride, out_stride,
                       lse != nullptr ? lse + r0 * lse_stride : nullptr, lse_stride);
    st = launch_status();
    if (st != 0) return st;
  }
  return 0;
}

}  // namespace

int sputnik_hip_sparse_attention_forward(int m, int n, int d, int nonzeros, int replicas,
                                         const int* row_indices, const int* row_offsets,
                                         const int* column_indices, const float* q,
                                         int64_t q_stride, const float* k, int64_t k_stride,
                                         const float* v, int64_t v_stride, float scale,
                                         float* out, int64_t out_stride, float* lse,
                                         int64_t lse_stride, void* workspace,
                                         size_t workspace_bytes, sputnik_hip_stream_t stream) {
  return attention_exec(m, n, d, nonzeros, replicas, row_indices, row_offsets, column_indices, q,
                        q_stride, k, k_stride, v, v_stride, scale, out, out_stride, lse,
                        lse_stride, workspace, workspace_bytes, /*planned=*/false, stream);
}

int sputnik_hip_sparse_attention_plan(int m, int n, int d, int nonzeros, const int* row_indices,
                                      const int* row_offsets, const int* column_indices,
                                      void* workspace, size_t workspace_bytes,
                                      sputnik_hip_stream_t stream) {
  if (m < 0 || n < 0 || d < 0 || nonzeros < 0) return SPUTNIK_HIP_INVALID_ARGUMENT;
  if (!supported(m, n, d, nonzeros)) return SPUTNIK_HIP_UNSUPPORTED;
  if (workspace == nullptr || !aligned_to(workspace, 16) ||
      workspace_bytes < sputnik_hip_sparse_attention_workspace_bytes(m, n, d, nonzeros))
    return SPUTNIK_HIP_INVALID_ARGUMENT;
  const int slots = slots_of(m);
  int* row_ok = static_cast<int*>(workspace);
  int* table = reinterpret_cast<int*>(static_cast<char*>(workspace) + row_ok_bytes(slots));
  hipLaunchKernelGGL((spmm_chunk_table_kernel<kBK>), dim3(ceil_div(slots, 4)), dim3(256),
                     0, stream, m, n, slots, kBM, chunks_of(n), row_indices, row_offsets,
                     column_indices, table, row_ok);
  return launch_status();
}

int sputnik_hip_sparse_attention_forward_planned(
    int m, int n, int d, int nonzeros, int replicas, const int* row_indices,
    const int* row_offsets, const int* column_indices, const float* q, int64_t q_stride,
    const float* k, int64_t k_stride, const float* v, int64_t v_stride, float scale, float* out,
    int64_t out_stride, float* lse, int64_t lse_stride, const void* workspace,
    size_t workspace_bytes, sputnik_hip_stream_t stream) {
  return attention_exec(m, n, d, nonzeros, replicas, row_indices, row_offsets, column_indices, q,
                        q_stride, k, k_stride, v, v_stride, scale, out, out_stride, lse,
                        lse_stride, const_cast<void*>(workspace), workspace_bytes,
                        /*planned=*/true, stream);
}

}  // extern "C"
