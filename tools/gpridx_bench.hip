// VGPR-index-mode (s_set_gpr_idx_*) experiments on gfx950 (developer tool, not shipped).
//
//   1. correctness: does an M0-relative destination / src2 work for v_pk_fma_f32
//      and v_fmac_f32, and which other instructions does the mode touch
//      (v_mov_b64_dpp, v_add_u32, ds_read)?
//   2. cost: the 8-column inner step of the SpMM kernel (one 64-bit DPP broadcast,
//      one address add, two ds_read_b128, four v_pk_fma_f32 per nonzero), software
//      pipelined, with and without the index-mode instructions around the FMAs.
//
//   hipcc --offload-arch=gfx950 -O2 tools/gpridx_bench.hip -o /tmp/gpridx && /tmp/gpridx
#include <hip/hip_runtime.h>

#include "gpridx_v2_body.inc"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

// ---------------------------------------------------------------------------
// 1. correctness
// ---------------------------------------------------------------------------
// acc = v[64..79] (16 registers, zeroed).  For idx in 0,2,..,14 with the mode
// on: v_pk_fma_f32 v[64:65] += (idx+1) * 1.0, M0-relative dst and src2.  Expect
// acc[idx] = acc[idx+1] = idx + 1.
__global__ void k_correct_pk(float* out) {
  float r[16];
  asm volatile(
      "v_mov_b32 v64, 0\nv_mov_b32 v65, 0\nv_mov_b32 v66, 0\nv_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\nv_mov_b32 v69, 0\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0\n"
      "v_mov_b32 v72, 0\nv_mov_b32 v73, 0\nv_mov_b32 v74, 0\nv_mov_b32 v75, 0\n"
      "v_mov_b32 v76, 0\nv_mov_b32 v77, 0\nv_mov_b32 v78, 0\nv_mov_b32 v79, 0\n"
      "v_mov_b32 v4, 1.0\nv_mov_b32 v5, 1.0\n"
      "s_mov_b32 s20, 0\n"
      "1:\n"
      "s_add_u32 s21, s20, 1\n"
      "v_cvt_f32_u32 v2, s21\n"
      "v_mov_b32 v3, v2\n"
      "s_set_gpr_idx_on s20, 0xc\n"
      "v_pk_fma_f32 v[64:65], v[2:3], v[4:5], v[64:65]\n"
      "s_set_gpr_idx_off\n"
      "s_add_u32 s20, s20, 2\n"
      "s_cmp_lt_u32 s20, 16\n"
      "s_cbranch_scc1 1b\n"
      "v_mov_b32 %0, v64\nv_mov_b32 %1, v65\nv_mov_b32 %2, v66\nv_mov_b32 %3, v67\n"
      "v_mov_b32 %4, v68\nv_mov_b32 %5, v69\nv_mov_b32 %6, v70\nv_mov_b32 %7, v71\n"
      "v_mov_b32 %8, v72\nv_mov_b32 %9, v73\nv_mov_b32 %10, v74\nv_mov_b32 %11, v75\n"
      "v_mov_b32 %12, v76\nv_mov_b32 %13, v77\nv_mov_b32 %14, v78\nv_mov_b32 %15, v79\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]),
        "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]),
        "=v"(r[14]), "=v"(r[15])
      :
      : "memory", "s20", "s21", "m0", "v2", "v3", "v4", "v5", "v64", "v65", "v66", "v67", "v68",
        "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) out[i] = r[i];
}

// The same with v_fmac_f32 (VOP2: src2 is the destination) under mode DST only
// (0x8) -- does the implied src2 follow the destination?
__global__ void k_correct_fmac(float* out, int mode_both) {
  float r[16];
  asm volatile(
      "v_mov_b32 v64, 0\nv_mov_b32 v65, 0\nv_mov_b32 v66, 0\nv_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\nv_mov_b32 v69, 0\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0\n"
      "v_mov_b32 v72, 0\nv_mov_b32 v73, 0\nv_mov_b32 v74, 0\nv_mov_b32 v75, 0\n"
      "v_mov_b32 v76, 0\nv_mov_b32 v77, 0\nv_mov_b32 v78, 0\nv_mov_b32 v79, 0\n"
      "v_mov_b32 v4, 1.0\n"
      "s_mov_b32 s20, 0\n"
      "1:\n"
      "s_add_u32 s21, s20, 1\n"
      "v_cvt_f32_u32 v2, s21\n"
      "s_set_gpr_idx_on s20, 0x8\n"
      "v_fmac_f32 v64, v2, v4\n"
      "v_fmac_f32 v64, v2, v4\n"
      "s_set_gpr_idx_off\n"
      "s_add_u32 s20, s20, 1\n"
      "s_cmp_lt_u32 s20, 16\n"
      "s_cbranch_scc1 1b\n"
      "v_mov_b32 %0, v64\nv_mov_b32 %1, v65\nv_mov_b32 %2, v66\nv_mov_b32 %3, v67\n"
      "v_mov_b32 %4, v68\nv_mov_b32 %5, v69\nv_mov_b32 %6, v70\nv_mov_b32 %7, v71\n"
      "v_mov_b32 %8, v72\nv_mov_b32 %9, v73\nv_mov_b32 %10, v74\nv_mov_b32 %11, v75\n"
      "v_mov_b32 %12, v76\nv_mov_b32 %13, v77\nv_mov_b32 %14, v78\nv_mov_b32 %15, v79\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]),
        "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]),
        "=v"(r[14]), "=v"(r[15])
      :
      : "memory", "s20", "s21", "m0", "v2", "v3", "v4", "v5", "v64", "v65", "v66", "v67", "v68",
        "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) out[i] = r[i];
}

// Which other instructions does mode DST|SRC2 (index 2) touch?  v_mov_b64_dpp
// v[64:65], v_add_u32 v68, v_mov_b32 v72, ds_read_b32 v76: the value lands in
// register +0 (untouched by the mode) or +2.
__global__ void k_touch(float* out) {
  __shared__ float lds[64];
  lds[threadIdx.x % 64] = 7.f;
  __syncthreads();
  float r[16];
  asm volatile(
      "v_mov_b32 v64, 0\nv_mov_b32 v65, 0\nv_mov_b32 v66, 0\nv_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\nv_mov_b32 v69, 0\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0\n"
      "v_mov_b32 v72, 0\nv_mov_b32 v73, 0\nv_mov_b32 v74, 0\nv_mov_b32 v75, 0\n"
      "v_mov_b32 v76, 0\nv_mov_b32 v77, 0\nv_mov_b32 v78, 0\nv_mov_b32 v79, 0\n"
      "v_mov_b32 v2, 1.0\nv_mov_b32 v3, 2.0\nv_mov_b32 v4, 3.0\nv_mov_b32 v5, 0\n"
      "s_mov_b32 s20, 2\n"
      "s_set_gpr_idx_on s20, 0xc\n"
      "v_mov_b64_dpp v[64:65], v[2:3] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32 v68, v4, v4\n"
      "v_mov_b32 v72, v4\n"
      "ds_read_b32 v76, v5\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_set_gpr_idx_off\n"
      "v_mov_b32 %0, v64\nv_mov_b32 %1, v65\nv_mov_b32 %2, v66\nv_mov_b32 %3, v67\n"
      "v_mov_b32 %4, v68\nv_mov_b32 %5, v69\nv_mov_b32 %6, v70\nv_mov_b32 %7, v71\n"
      "v_mov_b32 %8, v72\nv_mov_b32 %9, v73\nv_mov_b32 %10, v74\nv_mov_b32 %11, v75\n"
      "v_mov_b32 %12, v76\nv_mov_b32 %13, v77\nv_mov_b32 %14, v78\nv_mov_b32 %15, v79\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]),
        "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]),
        "=v"(r[14]), "=v"(r[15])
      :
      : "memory", "s20", "s21", "m0", "v2", "v3", "v4", "v5", "v64", "v65", "v66", "v67", "v68",
        "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) out[i] = r[i];
}

// ---------------------------------------------------------------------------
// 2. cost of the inner step
// ---------------------------------------------------------------------------
// Registers: v[6:7] entry window (offset 0, value), v40 lane base, strips in
// v[10:25] / v[48:63], broadcast pairs v[30:37], addresses v[44:47], accumulators
// v[64:127] (8 rows x 8 columns); s[40:43] = accumulator index of entries 0..3.
#define NZP8(U, PAIR_LO, PAIR_HI, ADDR, B0, B3, B4, B7)                                             \
  "v_mov_b64_dpp v[" PAIR_LO ":" PAIR_HI "], v[6:7] row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_add_u32 " ADDR ", v" PAIR_LO ", v40\n"                                                         \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n"                                                       \
  "ds_read_b128 v[" B4 ":" B7 "], " ADDR " offset:1024\n"
// four v_pk_fma_f32 into v[64:71] (M0-relative when the mode is on); A = register
// PAIR whose HIGH half is the value: op_sel picks the high half for both lanes
#define PK8(AP, B0, B2, B4, B6)                                                                \
  "v_pk_fma_f32 v[64:65], v[" AP "], v[" B0 "], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"   \
  "v_pk_fma_f32 v[66:67], v[" AP "], v[" B2 "], v[66:67] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"   \
  "v_pk_fma_f32 v[68:69], v[" AP "], v[" B4 "], v[68:69] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"   \
  "v_pk_fma_f32 v[70:71], v[" AP "], v[" B6 "], v[70:71] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"

#define ACC64                                                                                     \
  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",      \
      "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",  \
      "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",       \
      "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",     \
      "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",     \
      "v124", "v125", "v126", "v127"
#define WORK                                                                                      \
  "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22",      \
      "v23", "v24", "v25", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v44", "v45",  \
      "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",  \
      "v59", "v60", "v61", "v62", "v63", "v40", "v6", "v7"

#define DEF_STEP(NAME, BODY)                                                                      \
  __global__ __launch_bounds__(1024) void NAME(unsigned long long* out, int iters, int rnd) {     \
    __shared__ float lds[16384];                                                                  \
    for (int i = threadIdx.x; i < 16384; i += blockDim.x)                                         \
      lds[i] = rnd ? (float)((i * 2654435761u) >> 8) * (1.f / 16777216.f) : 1.f;                  \
    __syncthreads();                                                                              \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    unsigned int base = (threadIdx.x % 64) * 16;                                                  \
    /* entry window: (LDS row offset, value) per lane of a 16-lane row */                         \
    unsigned int woff = rnd ? ((threadIdx.x * 7u + 3u) % 31u) * 2048u : 0u;                       \
    float wval = rnd ? (float)(((threadIdx.x % 16) * 40503u) & 1023u) * (1.f / 1024.f) : 1.f;     \
    asm volatile("v_mov_b32 v40, %0\nv_mov_b32 v6, %2\nv_mov_b32 v7, %3\n"                        \
                 "s_mov_b32 s40, 0\ns_mov_b32 s41, 8\ns_mov_b32 s42, 24\ns_mov_b32 s43, 56\ns_mov_b32 s44, 0x38180800\n"     \
                 "s_mov_b32 s20, %1\n1:\n" BODY BODY BODY BODY                                    \
                 "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"                \
                 "s_waitcnt lgkmcnt(0)\n"                                                         \
                 :                                                                                \
                 : "v"(base), "s"(iters), "v"(woff), "v"(wval)                                    \
                 : "memory", "m0", "s20", "s40", "s41", "s42", "s43", "s44", "s45", WORK, ACC64);               \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    if (threadIdx.x % 64 == 0)                                                                    \
      out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                           \
    if (lds[threadIdx.x] < -1.f) out[0] = 0;                                                      \
  }

// a) no index mode: every entry into row 0 (the static-register form, pipelined:
//    the reads of two entries are issued before the FMAs of the previous two)
DEF_STEP(k_pipe_plain,
         NZP8("2", "34", "35", "v46", "48", "51", "52", "55") NZP8("3", "36", "37", "v47", "56", "59", "60", "63")
         "s_waitcnt lgkmcnt(6)\n" PK8("30:31", "10:11", "12:13", "14:15", "16:17")
         "s_waitcnt lgkmcnt(4)\n" PK8("32:33", "18:19", "20:21", "22:23", "24:25")
         NZP8("0", "30", "31", "v44", "10", "13", "14", "17") NZP8("1", "32", "33", "v45", "18", "21", "22", "25")
         "s_waitcnt lgkmcnt(6)\n" PK8("34:35", "48:49", "50:51", "52:53", "54:55")
         "s_waitcnt lgkmcnt(4)\n" PK8("36:37", "56:57", "58:59", "60:61", "62:63"))

// b) index mode switched on and off around every pair of entries, one
//    s_set_gpr_idx_idx between the two
DEF_STEP(k_pipe_idx,
         NZP8("2", "34", "35", "v46", "48", "51", "52", "55") NZP8("3", "36", "37", "v47", "56", "59", "60", "63")
         "s_waitcnt lgkmcnt(6)\ns_set_gpr_idx_on s40, 0xc\n" PK8("30:31", "10:11", "12:13", "14:15", "16:17")
         "s_waitcnt lgkmcnt(4)\ns_set_gpr_idx_idx s41\n" PK8("32:33", "18:19", "20:21", "22:23", "24:25")
         "s_set_gpr_idx_off\n"
         NZP8("0", "30", "31", "v44", "10", "13", "14", "17") NZP8("1", "32", "33", "v45", "18", "21", "22", "25")
         "s_waitcnt lgkmcnt(6)\ns_set_gpr_idx_on s42, 0xc\n" PK8("34:35", "48:49", "50:51", "52:53", "54:55")
         "s_waitcnt lgkmcnt(4)\ns_set_gpr_idx_idx s43\n" PK8("36:37", "56:57", "58:59", "60:61", "62:63")
         "s_set_gpr_idx_off\n")

// c) the same plus the scalar work of a flat-stream loop per pair of entries:
//    two byte extractions for the indices, a position compare and a branch
DEF_STEP(k_pipe_idx_salu,
         NZP8("2", "34", "35", "v46", "48", "51", "52", "55") NZP8("3", "36", "37", "v47", "56", "59", "60", "63")
         "s_bfe_u32 s40, s44, 0x80000\ns_bfe_u32 s41, s44, 0x80008\n"
         "s_waitcnt lgkmcnt(6)\ns_set_gpr_idx_on s40, 0xc\n" PK8("30:31", "10:11", "12:13", "14:15", "16:17")
         "s_waitcnt lgkmcnt(4)\ns_set_gpr_idx_idx s41\n" PK8("32:33", "18:19", "20:21", "22:23", "24:25")
         "s_set_gpr_idx_off\ns_cmp_eq_u32 s20, 0\ns_cbranch_scc1 2f\n"
         NZP8("0", "30", "31", "v44", "10", "13", "14", "17") NZP8("1", "32", "33", "v45", "18", "21", "22", "25")
         "s_bfe_u32 s42, s44, 0x80010\ns_bfe_u32 s43, s44, 0x80018\n"
         "s_waitcnt lgkmcnt(6)\ns_set_gpr_idx_on s42, 0xc\n" PK8("34:35", "48:49", "50:51", "52:53", "54:55")
         "s_waitcnt lgkmcnt(4)\ns_set_gpr_idx_idx s43\n" PK8("36:37", "56:57", "58:59", "60:61", "62:63")
         "s_set_gpr_idx_off\ns_cmp_eq_u32 s20, 0\ns_cbranch_scc1 2f\n2:\n")

// d) column-grouped loop: the B strip of a column is read ONCE for all the rows of the wave
//    that hold the column (entries sorted by column); per entry only a 32-bit value broadcast
//    and the four FMAs.  Body = one 16-entry window with 9 column groups (density 0.1, 8 rows
//    per wave) / 4 groups (density 0.5).  DEF_STEP repeats the body 4 times per iteration.
DEF_STEP(k_colgroup_d10, V2_BODY_D10)
DEF_STEP(k_colgroup_d50, V2_BODY_D50)
// (the value broadcast of the NEXT entry issued in front of the FMAs of this one)
DEF_STEP(k_colgroup_d10_ahead, V2_BODY_D10_AHEAD)
DEF_STEP(k_colgroup_d50_ahead, V2_BODY_D50_AHEAD)

// e) 256-column tile (4 columns per lane, 16 rows per wave): entries in PAIRS -- one
//    chunk test and one index-mode region per two entries, one ds_read_b128 and two
//    v_pk_fma_f32 per entry, FMAs three pairs behind the reads.  8 entries per body.
DEF_STEP(k_pair256, "s_mov_b32 s45, 0x7fffffff\n" V4_PAIR_BODY)

// f) HALF dense operand kept as half in LDS (8 columns per lane = ONE ds_read_b128 per
//    entry): fp16 through v_fma_mix_f32 (f32 value x f16 B + f32 sum, 8 per entry), and
//    bf16 widened by a shift / a mask outside the index-mode region (8 VALU) in front of
//    the four v_pk_fma_f32.  Same scalar work as c).
#define NZH8(U, PAIR_LO, PAIR_HI, ADDR, B0, B3)                                                     \
  "v_mov_b64_dpp v[" PAIR_LO ":" PAIR_HI "], v[6:7] row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_add_u32 " ADDR ", v" PAIR_LO ", v40\n"                                                         \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n"
#define MIX8(AV, B0, B1, B2, B3)                                                      \
  "v_fma_mix_f32 v64, v" AV ", v" B0 ", v64 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n"       \
  "v_fma_mix_f32 v65, v" AV ", v" B0 ", v65 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"       \
  "v_fma_mix_f32 v66, v" AV ", v" B1 ", v66 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n"       \
  "v_fma_mix_f32 v67, v" AV ", v" B1 ", v67 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"       \
  "v_fma_mix_f32 v68, v" AV ", v" B2 ", v68 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n"       \
  "v_fma_mix_f32 v69, v" AV ", v" B2 ", v69 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"       \
  "v_fma_mix_f32 v70, v" AV ", v" B3 ", v70 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n"       \
  "v_fma_mix_f32 v71, v" AV ", v" B3 ", v71 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"
// bf16 -> f32: odd columns keep the high half (mask), even columns shift up; W0..W7 = 8
// consecutive registers taking the widened strip, written from the top so the strip may
// start in the same registers
#define WIDEN8(B0, B1, B2, B3, W0, W1, W2, W3, W4, W5, W6, W7)                        \
  "v_and_b32 v" W7 ", 0xffff0000, v" B3 "\nv_lshlrev_b32 v" W6 ", 16, v" B3 "\n"      \
  "v_and_b32 v" W5 ", 0xffff0000, v" B2 "\nv_lshlrev_b32 v" W4 ", 16, v" B2 "\n"      \
  "v_and_b32 v" W3 ", 0xffff0000, v" B1 "\nv_lshlrev_b32 v" W2 ", 16, v" B1 "\n"      \
  "v_and_b32 v" W1 ", 0xffff0000, v" B0 "\nv_lshlrev_b32 v" W0 ", 16, v" B0 "\n"

DEF_STEP(k_half_mix,
         NZH8("2", "34", "35", "v46", "48", "51") NZH8("3", "36", "37", "v47", "56", "59")
         "s_bfe_u32 s40, s44, 0x80000\ns_bfe_u32 s41, s44, 0x80008\n"
         "s_waitcnt lgkmcnt(3)\ns_set_gpr_idx_on s40, 0xc\n" MIX8("31", "10", "11", "12", "13")
         "s_waitcnt lgkmcnt(2)\ns_set_gpr_idx_idx s41\n" MIX8("33", "18", "19", "20", "21")
         "s_set_gpr_idx_off\ns_cmp_eq_u32 s20, 0\ns_cbranch_scc1 2f\n"
         NZH8("0", "30", "31", "v44", "10", "13") NZH8("1", "32", "33", "v45", "18", "21")
         "s_bfe_u32 s42, s44, 0x80010\ns_bfe_u32 s43, s44, 0x80018\n"
         "s_waitcnt lgkmcnt(3)\ns_set_gpr_idx_on s42, 0xc\n" MIX8("35", "48", "49", "50", "51")
         "s_waitcnt lgkmcnt(2)\ns_set_gpr_idx_idx s43\n" MIX8("37", "56", "57", "58", "59")
         "s_set_gpr_idx_off\ns_cmp_eq_u32 s20, 0\ns_cbranch_scc1 2f\n2:\n")

DEF_STEP(k_half_widen,
         NZH8("2", "34", "35", "v46", "48", "51") NZH8("3", "36", "37", "v47", "56", "59")
         "s_bfe_u32 s40, s44, 0x80000\ns_bfe_u32 s41, s44, 0x80008\n"
         "s_waitcnt lgkmcnt(3)\n" WIDEN8("10", "11", "12", "13", "10", "11", "12", "13", "14", "15", "16", "17")
         "s_waitcnt lgkmcnt(2)\n" WIDEN8("18", "19", "20", "21", "18", "19", "20", "21", "22", "23", "24", "25")
         "s_set_gpr_idx_on s40, 0xc\n" PK8("30:31", "10:11", "12:13", "14:15", "16:17")
         "s_set_gpr_idx_idx s41\n" PK8("32:33", "18:19", "20:21", "22:23", "24:25")
         "s_set_gpr_idx_off\ns_cmp_eq_u32 s20, 0\ns_cbranch_scc1 2f\n"
         NZH8("0", "30", "31", "v44", "10", "13") NZH8("1", "32", "33", "v45", "18", "21")
         "s_bfe_u32 s42, s44, 0x80010\ns_bfe_u32 s43, s44, 0x80018\n"
         "s_waitcnt lgkmcnt(3)\n" WIDEN8("48", "49", "50", "51", "48", "49", "50", "51", "52", "53", "54", "55")
         "s_waitcnt lgkmcnt(2)\n" WIDEN8("56", "57", "58", "59", "56", "57", "58", "59", "60", "61", "62", "63")
         "s_set_gpr_idx_on s42, 0xc\n" PK8("34:35", "48:49", "50:51", "52:53", "54:55")
         "s_set_gpr_idx_idx s43\n" PK8("36:37", "56:57", "58:59", "60:61", "62:63")
         "s_set_gpr_idx_off\ns_cmp_eq_u32 s20, 0\ns_cbranch_scc1 2f\n2:\n")

// v_fma_mix_f32 under DST|SRC2 index mode: acc[idx] += (idx+1) * half(1.0), both halves
// of the B register in turn -> acc[idx] = 2*(idx+1) if dst and src2 both follow M0.
__global__ void k_correct_mix(float* out) {
  float r[16];
  asm volatile(
      "v_mov_b32 v64, 0\nv_mov_b32 v65, 0\nv_mov_b32 v66, 0\nv_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\nv_mov_b32 v69, 0\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0\n"
      "v_mov_b32 v72, 0\nv_mov_b32 v73, 0\nv_mov_b32 v74, 0\nv_mov_b32 v75, 0\n"
      "v_mov_b32 v76, 0\nv_mov_b32 v77, 0\nv_mov_b32 v78, 0\nv_mov_b32 v79, 0\n"
      "v_mov_b32 v4, 0x3c003c00\n"
      "s_mov_b32 s20, 0\n"
      "1:\n"
      "s_add_u32 s21, s20, 1\n"
      "v_cvt_f32_u32 v2, s21\n"
      "s_set_gpr_idx_on s20, 0xc\n"
      "v_fma_mix_f32 v64, v2, v4, v64 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n"
      "v_fma_mix_f32 v64, v2, v4, v64 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"
      "s_set_gpr_idx_off\n"
      "s_add_u32 s20, s20, 1\n"
      "s_cmp_lt_u32 s20, 16\n"
      "s_cbranch_scc1 1b\n"
      "v_mov_b32 %0, v64\nv_mov_b32 %1, v65\nv_mov_b32 %2, v66\nv_mov_b32 %3, v67\n"
      "v_mov_b32 %4, v68\nv_mov_b32 %5, v69\nv_mov_b32 %6, v70\nv_mov_b32 %7, v71\n"
      "v_mov_b32 %8, v72\nv_mov_b32 %9, v73\nv_mov_b32 %10, v74\nv_mov_b32 %11, v75\n"
      "v_mov_b32 %12, v76\nv_mov_b32 %13, v77\nv_mov_b32 %14, v78\nv_mov_b32 %15, v79\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]),
        "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]),
        "=v"(r[14]), "=v"(r[15])
      :
      : "memory", "s20", "s21", "m0", "v2", "v3", "v4", "v5", "v64", "v65", "v66", "v67", "v68",
        "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) out[i] = r[i];
}

typedef void (*kern_t)(unsigned long long*, int, int);

static void run(const char* name, kern_t k, int rnd, int per_iter = 16) {
  const int iters = 20000;
  for (int waves : {4, 8, 16}) {
    unsigned long long* d;
    CHECK(hipMalloc(&d, sizeof(unsigned long long) * 256 * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 0, 0, d, 500, rnd);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 0, 0, d, iters, rnd);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // entries per loop iteration: 4 bodies of 4, or 4 windows of 16
    const double entries_per_simd = double(iters) * per_iter * (waves / 4.0);
    printf("%-18s rnd=%d waves/SIMD=%d  wall=%.3f ms  ns/entry/SIMD=%.2f\n", name, rnd, waves / 4, ms,
           ms * 1e6 / entries_per_simd);
    CHECK(hipFree(d));
  }
}

int main() {
  float* d;
  CHECK(hipMalloc(&d, 64 * sizeof(float)));
  float h[16];
  hipLaunchKernelGGL(k_correct_pk, dim3(1), dim3(64), 0, 0, d);
  CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("pk_fma dst|src2 relative (want 1 1 3 3 5 5 ...):");
  for (float v : h) printf(" %g", v);
  printf("\n");
  hipLaunchKernelGGL(k_correct_fmac, dim3(1), dim3(64), 0, 0, d, 0);
  CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("fmac dst relative, two per index (want 2 4 6 ... if src2 follows dst):");
  for (float v : h) printf(" %g", v);
  printf("\n");
  hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, 0, d);
  CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("mode on, index 2: mov_b64_dpp->v64 add->v68 mov->v72 ds_read->v76:");
  for (float v : h) printf(" %g", v);
  printf("\n");
  hipLaunchKernelGGL(k_correct_mix, dim3(1), dim3(64), 0, 0, d);
  CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("fma_mix dst|src2 relative, both halves (want 2 4 6 ...):");
  for (float v : h) printf(" %g", v);
  printf("\n");
  for (int rnd = 0; rnd < 2; ++rnd) {
    run("half_mix(fp16 B)", k_half_mix, rnd);
    run("half_widen(bf16 B)", k_half_widen, rnd);
    run("pipe_plain", k_pipe_plain, rnd);
    run("pipe_idx", k_pipe_idx, rnd);
    run("pipe_idx_salu", k_pipe_idx_salu, rnd);
    run("pair256(per 256-col entry)", k_pair256, rnd, 32);
    run("colgroup_d10", k_colgroup_d10, rnd, 64);
    run("colgroup_d50", k_colgroup_d50, rnd, 64);
    run("colgroup_d10_ahead", k_colgroup_d10_ahead, rnd, 64);
    run("colgroup_d50_ahead", k_colgroup_d50_ahead, rnd, 64);
  }
  return 0;
}
