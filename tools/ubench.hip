// Instruction-cost microbenchmarks for gfx950 (developer tool, not shipped).
// Each kernel runs ITERS iterations of a 16-instruction block on every wave of
// a 256-workgroup grid and reports shader cycles per instruction per SIMD for
// 1, 2 and 4 waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

#define REP16(X) X X X X X X X X X X X X X X X X

#define LOOP_BEGIN "s_mov_b32 s20, %1\n1:\n"
#define LOOP_END "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"

#define DEF_KERNEL(NAME, BODY, CLOBBERS...)                                          \
  __global__ void NAME(unsigned long long* out, int iters) {                          \
    __shared__ float lds[16384];                                                      \
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;                 \
    __syncthreads();                                                                  \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                             \
    unsigned int base = (threadIdx.x % 64) * 16;                                      \
    asm volatile("v_mov_b32 v40, %0\n"                                                \
                 "s_mov_b32 s30, 0x3f800000\ns_mov_b32 s31, 0x3f800000\n"             \
                 "s_mov_b32 s38, 5\nv_mov_b32 v5, 0\nv_mov_b32 v6, 0\nv_mov_b32 v7, 0\n"                                                 \
                 LOOP_BEGIN BODY BODY BODY BODY LOOP_END                                             \
                 "s_waitcnt lgkmcnt(0)\n"                                             \
                 : : "v"(base), "s"(iters)                                            \
                 : "memory", "s20", "s30", "s31", "s38", "s33", "s34", "s35", "s36", "v40", "v5", "v6", "v7", CLOBBERS); \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    if (lds[threadIdx.x] < -1.f) out[0] = 0;                                          \
  }

#define V16 "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25"
#define V32 V16, "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v42", "v43"

// a) v_fma_f32, all-VGPR operands, 16 independent accumulators
DEF_KERNEL(k_fma_vgpr,
           "v_fma_f32 v10, v2, v3, v10\nv_fma_f32 v11, v2, v3, v11\nv_fma_f32 v12, v2, v3, v12\nv_fma_f32 v13, v2, v3, v13\n"
           "v_fma_f32 v14, v2, v3, v14\nv_fma_f32 v15, v2, v3, v15\nv_fma_f32 v16, v2, v3, v16\nv_fma_f32 v17, v2, v3, v17\n"
           "v_fma_f32 v18, v2, v3, v18\nv_fma_f32 v19, v2, v3, v19\nv_fma_f32 v20, v2, v3, v20\nv_fma_f32 v21, v2, v3, v21\n"
           "v_fma_f32 v22, v2, v3, v22\nv_fma_f32 v23, v2, v3, v23\nv_fma_f32 v24, v2, v3, v24\nv_fma_f32 v25, v2, v3, v25\n",
           V16)

// b) v_fma_f32 with an SGPR multiplier
DEF_KERNEL(k_fma_sgpr,
           "v_fma_f32 v10, s30, v3, v10\nv_fma_f32 v11, s30, v3, v11\nv_fma_f32 v12, s30, v3, v12\nv_fma_f32 v13, s30, v3, v13\n"
           "v_fma_f32 v14, s30, v3, v14\nv_fma_f32 v15, s30, v3, v15\nv_fma_f32 v16, s30, v3, v16\nv_fma_f32 v17, s30, v3, v17\n"
           "v_fma_f32 v18, s30, v3, v18\nv_fma_f32 v19, s30, v3, v19\nv_fma_f32 v20, s30, v3, v20\nv_fma_f32 v21, s30, v3, v21\n"
           "v_fma_f32 v22, s30, v3, v22\nv_fma_f32 v23, s30, v3, v23\nv_fma_f32 v24, s30, v3, v24\nv_fma_f32 v25, s30, v3, v25\n",
           V16)

// c) v_pk_fma_f32, all-VGPR
DEF_KERNEL(k_pkfma_vgpr,
           "v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[10:11]\nv_pk_fma_f32 v[12:13], v[2:3], v[4:5], v[12:13]\n"
           "v_pk_fma_f32 v[14:15], v[2:3], v[4:5], v[14:15]\nv_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[16:17]\n"
           "v_pk_fma_f32 v[18:19], v[2:3], v[4:5], v[18:19]\nv_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[20:21]\n"
           "v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[22:23]\nv_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[24:25]\n"
           "v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[26:27]\nv_pk_fma_f32 v[28:29], v[2:3], v[4:5], v[28:29]\n"
           "v_pk_fma_f32 v[30:31], v[2:3], v[4:5], v[30:31]\nv_pk_fma_f32 v[32:33], v[2:3], v[4:5], v[32:33]\n"
           "v_pk_fma_f32 v[34:35], v[2:3], v[4:5], v[34:35]\nv_pk_fma_f32 v[36:37], v[2:3], v[4:5], v[36:37]\n"
           "v_pk_fma_f32 v[38:39], v[2:3], v[4:5], v[38:39]\nv_pk_fma_f32 v[42:43], v[2:3], v[4:5], v[42:43]\n",
           V32)

// d) v_pk_fma_f32 with an SGPR pair broadcast (the form hipcc emits for a*b+c with scalar a)
DEF_KERNEL(k_pkfma_sgpr,
           "v_pk_fma_f32 v[10:11], s[30:31], v[4:5], v[10:11] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[12:13], s[30:31], v[4:5], v[12:13] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[14:15], s[30:31], v[4:5], v[14:15] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[16:17], s[30:31], v[4:5], v[16:17] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[18:19], s[30:31], v[4:5], v[18:19] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[20:21], s[30:31], v[4:5], v[20:21] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[22:23], s[30:31], v[4:5], v[22:23] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[24:25], s[30:31], v[4:5], v[24:25] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[26:27], s[30:31], v[4:5], v[26:27] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[28:29], s[30:31], v[4:5], v[28:29] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[30:31], s[30:31], v[4:5], v[30:31] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[32:33], s[30:31], v[4:5], v[32:33] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[34:35], s[30:31], v[4:5], v[34:35] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[36:37], s[30:31], v[4:5], v[36:37] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[38:39], s[30:31], v[4:5], v[38:39] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[42:43], s[30:31], v[4:5], v[42:43] op_sel_hi:[0,1,1]\n",
           V32)

// e) v_readlane_b32 with an SGPR lane select
DEF_KERNEL(k_readlane,
           REP16("v_readlane_b32 s33, v3, s38\n"), "v10")

// f) v_lshl_add_u32 v, s, 10, v
DEF_KERNEL(k_lshl_add,
           "v_lshl_add_u32 v10, s38, 10, v40\nv_lshl_add_u32 v11, s38, 10, v40\nv_lshl_add_u32 v12, s38, 10, v40\nv_lshl_add_u32 v13, s38, 10, v40\n"
           "v_lshl_add_u32 v14, s38, 10, v40\nv_lshl_add_u32 v15, s38, 10, v40\nv_lshl_add_u32 v16, s38, 10, v40\nv_lshl_add_u32 v17, s38, 10, v40\n"
           "v_lshl_add_u32 v18, s38, 10, v40\nv_lshl_add_u32 v19, s38, 10, v40\nv_lshl_add_u32 v20, s38, 10, v40\nv_lshl_add_u32 v21, s38, 10, v40\n"
           "v_lshl_add_u32 v22, s38, 10, v40\nv_lshl_add_u32 v23, s38, 10, v40\nv_lshl_add_u32 v24, s38, 10, v40\nv_lshl_add_u32 v25, s38, 10, v40\n",
           V16)

// g) ds_read_b128 stream: 8 reads (32 VGPRs) then wait; block = 16 "instructions" = 2 x this
DEF_KERNEL(k_ds_read_b128,
           "ds_read_b128 v[10:13], v40\nds_read_b128 v[14:17], v40 offset:1024\nds_read_b128 v[18:21], v40 offset:2048\nds_read_b128 v[22:25], v40 offset:3072\n"
           "ds_read_b128 v[26:29], v40 offset:4096\nds_read_b128 v[30:33], v40 offset:5120\nds_read_b128 v[34:37], v40 offset:6144\nds_read_b128 v[10:13], v40 offset:7168\n"
           "ds_read_b128 v[14:17], v40 offset:8192\nds_read_b128 v[18:21], v40 offset:9216\nds_read_b128 v[22:25], v40 offset:10240\nds_read_b128 v[26:29], v40 offset:11264\n"
           "ds_read_b128 v[30:33], v40 offset:12288\nds_read_b128 v[34:37], v40 offset:13312\nds_read_b128 v[10:13], v40 offset:14336\nds_read_b128 v[14:17], v40 offset:15360\n"
           "s_waitcnt lgkmcnt(0)\n",
           V32)

// h) the SpMM inner step as compiled: per nonzero 2 readlane, s_sub, lshl_add, ds_read_b128,
//    2 pk_fma (4 nonzeros per block -> counts as 16 "instructions" = 4 nonzeros x 4 FMA-lanes)
#define NZ(J, A, ADDR, B0, B1, B2, B3)                                     \
  "v_readlane_b32 " J ", v3, s38\n"                                          \
  "v_readlane_b32 " A ", v4, s38\n"                                          \
  "s_and_b32 " J ", " J ", 15\n"                                             \
  "v_lshl_add_u32 " ADDR ", " J ", 10, v40\n"                                \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n"
DEF_KERNEL(k_spmm_step,
           NZ("s33", "s34", "v44", "10", "11", "12", "13")
           NZ("s35", "s36", "v45", "14", "15", "16", "17")
           "s_waitcnt lgkmcnt(1)\n"
           "v_pk_fma_f32 v[26:27], s[34:35], v[10:11], v[26:27] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[28:29], s[34:35], v[12:13], v[28:29] op_sel_hi:[0,1,1]\n"
           "s_waitcnt lgkmcnt(0)\n"
           "v_pk_fma_f32 v[26:27], s[36:37], v[14:15], v[26:27] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[28:29], s[36:37], v[16:17], v[28:29] op_sel_hi:[0,1,1]\n"
           NZ("s33", "s34", "v44", "18", "19", "20", "21")
           NZ("s35", "s36", "v45", "22", "23", "24", "25")
           "s_waitcnt lgkmcnt(1)\n"
           "v_pk_fma_f32 v[30:31], s[34:35], v[18:19], v[30:31] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[32:33], s[34:35], v[20:21], v[32:33] op_sel_hi:[0,1,1]\n"
           "s_waitcnt lgkmcnt(0)\n"
           "v_pk_fma_f32 v[30:31], s[36:37], v[22:23], v[30:31] op_sel_hi:[0,1,1]\n"
           "v_pk_fma_f32 v[32:33], s[36:37], v[24:25], v[32:33] op_sel_hi:[0,1,1]\n",
           V32, "v44", "v45", "s37")

// i) same step with plain v_fma_f32 (4 per nonzero) instead of v_pk_fma_f32
DEF_KERNEL(k_spmm_step_fma,
           NZ("s33", "s34", "v44", "10", "11", "12", "13")
           NZ("s35", "s36", "v45", "14", "15", "16", "17")
           "s_waitcnt lgkmcnt(1)\n"
           "v_fma_f32 v26, s34, v10, v26\nv_fma_f32 v27, s34, v11, v27\nv_fma_f32 v28, s34, v12, v28\nv_fma_f32 v29, s34, v13, v29\n"
           "s_waitcnt lgkmcnt(0)\n"
           "v_fma_f32 v26, s36, v14, v26\nv_fma_f32 v27, s36, v15, v27\nv_fma_f32 v28, s36, v16, v28\nv_fma_f32 v29, s36, v17, v29\n"
           NZ("s33", "s34", "v44", "18", "19", "20", "21")
           NZ("s35", "s36", "v45", "22", "23", "24", "25")
           "s_waitcnt lgkmcnt(1)\n"
           "v_fma_f32 v30, s34, v18, v30\nv_fma_f32 v31, s34, v19, v31\nv_fma_f32 v32, s34, v20, v32\nv_fma_f32 v33, s34, v21, v33\n"
           "s_waitcnt lgkmcnt(0)\n"
           "v_fma_f32 v30, s36, v22, v30\nv_fma_f32 v31, s36, v23, v31\nv_fma_f32 v32, s36, v24, v32\nv_fma_f32 v33, s36, v25, v33\n",
           V32, "v44", "v45", "s37")

// j) v_fmac_f32 with a DPP row_newbcast source (value broadcast inside each 16-lane row)
DEF_KERNEL(k_fmac_dpp,
           "v_fmac_f32_dpp v10, v3, v4 row_newbcast:0 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v11, v3, v4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
           "v_fmac_f32_dpp v12, v3, v4 row_newbcast:2 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v13, v3, v4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
           "v_fmac_f32_dpp v14, v3, v4 row_newbcast:4 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v15, v3, v4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
           "v_fmac_f32_dpp v16, v3, v4 row_newbcast:6 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v17, v3, v4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
           "v_fmac_f32_dpp v18, v3, v4 row_newbcast:8 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v19, v3, v4 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
           "v_fmac_f32_dpp v20, v3, v4 row_newbcast:10 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v21, v3, v4 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
           "v_fmac_f32_dpp v22, v3, v4 row_newbcast:12 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v23, v3, v4 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
           "v_fmac_f32_dpp v24, v3, v4 row_newbcast:14 row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp v25, v3, v4 row_newbcast:15 row_mask:0xf bank_mask:0xf\n",
           V16)

// k) plain v_fmac_f32 (VOP2, 4-byte encoding)
DEF_KERNEL(k_fmac_vop2,
           "v_fmac_f32 v10, v3, v4\nv_fmac_f32 v11, v3, v4\nv_fmac_f32 v12, v3, v4\nv_fmac_f32 v13, v3, v4\n"
           "v_fmac_f32 v14, v3, v4\nv_fmac_f32 v15, v3, v4\nv_fmac_f32 v16, v3, v4\nv_fmac_f32 v17, v3, v4\n"
           "v_fmac_f32 v18, v3, v4\nv_fmac_f32 v19, v3, v4\nv_fmac_f32 v20, v3, v4\nv_fmac_f32 v21, v3, v4\n"
           "v_fmac_f32 v22, v3, v4\nv_fmac_f32 v23, v3, v4\nv_fmac_f32 v24, v3, v4\nv_fmac_f32 v25, v3, v4\n",
           V16)

// l) SpMM step, DPP form: per nonzero v_add_u32_dpp (address) + ds_read_b128 + 4 v_fmac_f32_dpp
#define NZD(U, ADDR, B0, B3) \
  "v_and_b32_dpp " ADDR ", v5, v6 row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_add_u32 " ADDR ", " ADDR ", v40\n" \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n"
#define FMD(U, ACC, B) "v_fmac_f32_dpp " ACC ", v4, " B " row_newbcast:" U " row_mask:0xf bank_mask:0xf\n"
DEF_KERNEL(k_spmm_step_dpp,
           NZD("0", "v44", "10", "13") NZD("1", "v45", "14", "17") NZD("2", "v46", "18", "21") NZD("3", "v47", "22", "25")
           "s_waitcnt lgkmcnt(3)\n" FMD("0", "v26", "v10") FMD("0", "v27", "v11") FMD("0", "v28", "v12") FMD("0", "v29", "v13")
           "s_waitcnt lgkmcnt(2)\n" FMD("1", "v26", "v14") FMD("1", "v27", "v15") FMD("1", "v28", "v16") FMD("1", "v29", "v17")
           "s_waitcnt lgkmcnt(1)\n" FMD("2", "v26", "v18") FMD("2", "v27", "v19") FMD("2", "v28", "v20") FMD("2", "v29", "v21")
           "s_waitcnt lgkmcnt(0)\n" FMD("3", "v26", "v22") FMD("3", "v27", "v23") FMD("3", "v28", "v24") FMD("3", "v29", "v25"),
           V32, "v44", "v45", "v46", "v47")

// m) same, address formed by ONE v_add_u32_dpp (lane offset + broadcast row offset)
#define NZE(U, ADDR, B0, B3) \
  "v_add_u32_dpp " ADDR ", v5, v40 row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n"
DEF_KERNEL(k_spmm_step_dpp1,
           NZE("0", "v44", "10", "13") NZE("1", "v45", "14", "17") NZE("2", "v46", "18", "21") NZE("3", "v47", "22", "25")
           "s_waitcnt lgkmcnt(3)\n" FMD("0", "v26", "v10") FMD("0", "v27", "v11") FMD("0", "v28", "v12") FMD("0", "v29", "v13")
           "s_waitcnt lgkmcnt(2)\n" FMD("1", "v26", "v14") FMD("1", "v27", "v15") FMD("1", "v28", "v16") FMD("1", "v29", "v17")
           "s_waitcnt lgkmcnt(1)\n" FMD("2", "v26", "v18") FMD("2", "v27", "v19") FMD("2", "v28", "v20") FMD("2", "v29", "v21")
           "s_waitcnt lgkmcnt(0)\n" FMD("3", "v26", "v22") FMD("3", "v27", "v23") FMD("3", "v28", "v24") FMD("3", "v29", "v25"),
           V32, "v44", "v45", "v46", "v47")

// n) SALU throughput: 16 independent s_and_b32
DEF_KERNEL(k_salu,
           "s_and_b32 s33, s38, 63\ns_and_b32 s34, s38, 63\ns_and_b32 s35, s38, 63\ns_and_b32 s36, s38, 63\n"
           "s_and_b32 s33, s38, 63\ns_and_b32 s34, s38, 63\ns_and_b32 s35, s38, 63\ns_and_b32 s36, s38, 63\n"
           "s_and_b32 s33, s38, 63\ns_and_b32 s34, s38, 63\ns_and_b32 s35, s38, 63\ns_and_b32 s36, s38, 63\n"
           "s_and_b32 s33, s38, 63\ns_and_b32 s34, s38, 63\ns_and_b32 s35, s38, 63\ns_and_b32 s36, s38, 63\n",
           "v10")

// o) SALU next to VALU: 8 v_fmac + 8 s_and interleaved (do they overlap?)
DEF_KERNEL(k_salu_valu,
           "v_fmac_f32 v10, v3, v4\ns_and_b32 s33, s38, 63\nv_fmac_f32 v11, v3, v4\ns_and_b32 s34, s38, 63\n"
           "v_fmac_f32 v12, v3, v4\ns_and_b32 s35, s38, 63\nv_fmac_f32 v13, v3, v4\ns_and_b32 s36, s38, 63\n"
           "v_fmac_f32 v14, v3, v4\ns_and_b32 s33, s38, 63\nv_fmac_f32 v15, v3, v4\ns_and_b32 s34, s38, 63\n"
           "v_fmac_f32 v16, v3, v4\ns_and_b32 s35, s38, 63\nv_fmac_f32 v17, v3, v4\ns_and_b32 s36, s38, 63\n",
           V16)

// p) scalar-operand SpMM step: per nonzero s_and, v_lshl_add_u32 (SGPR), ds_read_b128, 2 v_pk_fma (SGPR pair)
#define NZS(J, ADDR, B0, B3) \
  "s_and_b32 " J ", s38, 15\n" \
  "v_lshl_add_u32 " ADDR ", " J ", 10, v40\n" \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n"
DEF_KERNEL(k_spmm_step_scalar,
           NZS("s33", "v44", "10", "13") NZS("s34", "v45", "14", "17") NZS("s35", "v46", "18", "21") NZS("s36", "v47", "22", "25")
           "s_waitcnt lgkmcnt(3)\n"
           "v_pk_fma_f32 v[26:27], s[30:31], v[10:11], v[26:27] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[28:29], s[30:31], v[12:13], v[28:29] op_sel_hi:[0,1,1]\n"
           "s_waitcnt lgkmcnt(2)\n"
           "v_pk_fma_f32 v[26:27], s[30:31], v[14:15], v[26:27] op_sel:[1,0,0]\nv_pk_fma_f32 v[28:29], s[30:31], v[16:17], v[28:29] op_sel:[1,0,0]\n"
           "s_waitcnt lgkmcnt(1)\n"
           "v_pk_fma_f32 v[26:27], s[30:31], v[18:19], v[26:27] op_sel_hi:[0,1,1]\nv_pk_fma_f32 v[28:29], s[30:31], v[20:21], v[28:29] op_sel_hi:[0,1,1]\n"
           "s_waitcnt lgkmcnt(0)\n"
           "v_pk_fma_f32 v[26:27], s[30:31], v[22:23], v[26:27] op_sel:[1,0,0]\nv_pk_fma_f32 v[28:29], s[30:31], v[24:25], v[28:29] op_sel:[1,0,0]\n",
           V32, "v44", "v45", "v46", "v47")

// m) v_mov_b32 with a DPP row_newbcast source
DEF_KERNEL(k_mov_dpp32,
           "v_mov_b32_dpp v10, v3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v11, v3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v12, v3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v13, v3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v14, v3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v15, v3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v16, v3 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v17, v3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v18, v3 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v19, v3 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v20, v3 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v21, v3 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v22, v3 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v23, v3 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v24, v3 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b32_dpp v25, v3 row_newbcast:15 row_mask:0xf bank_mask:0xf\n",
           V16)

// n) 64-bit DPP move (DP-rate DPP only supports row_newbcast): broadcasts a register PAIR
DEF_KERNEL(k_mov_dpp64,
           "v_mov_b64_dpp v[10:11], v[2:3] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[12:13], v[2:3] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[14:15], v[2:3] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[16:17], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[18:19], v[2:3] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[20:21], v[2:3] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[22:23], v[2:3] row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[24:25], v[2:3] row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[26:27], v[2:3] row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[28:29], v[2:3] row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[30:31], v[2:3] row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[32:33], v[2:3] row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[34:35], v[2:3] row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[36:37], v[2:3] row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[38:39], v[2:3] row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
           "v_mov_b64_dpp v[42:43], v[2:3] row_newbcast:15 row_mask:0xf bank_mask:0xf\n",
           V32, "v41")

// o) SpMM step as shipped: per nonzero v_mov_b64_dpp (offset, value) + v_add_u32 + ds_read_b128 + 4 v_fmac_f32
#define NZP4(U, PAIR_LO, PAIR_HI, ADDR, B0, B3) \
  "v_mov_b64_dpp v[" PAIR_LO ":" PAIR_HI "], v[6:7] row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_add_u32 " ADDR ", v" PAIR_LO ", v40\n" \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n"
#define FM4(A, B0, B1, B2, B3) \
  "v_fmac_f32 v26, " A ", " B0 "\nv_fmac_f32 v27, " A ", " B1 "\nv_fmac_f32 v28, " A ", " B2 "\nv_fmac_f32 v29, " A ", " B3 "\n"
DEF_KERNEL(k_step_pair_v4,
           NZP4("0", "30", "31", "v44", "10", "13") NZP4("1", "32", "33", "v45", "14", "17")
           NZP4("2", "34", "35", "v46", "18", "21") NZP4("3", "36", "37", "v47", "22", "25")
           "s_waitcnt lgkmcnt(3)\n" FM4("v31", "v10", "v11", "v12", "v13")
           "s_waitcnt lgkmcnt(2)\n" FM4("v33", "v14", "v15", "v16", "v17")
           "s_waitcnt lgkmcnt(1)\n" FM4("v35", "v18", "v19", "v20", "v21")
           "s_waitcnt lgkmcnt(0)\n" FM4("v37", "v22", "v23", "v24", "v25"),
           V32, "v44", "v45", "v46", "v47")

// o2) as o) but every nonzero of a step accumulates into its OWN four registers
//     (no FMA depends on the previous nonzero's FMA): does the dependent chain cost?
#define FM4I(A, ACC0, ACC1, ACC2, ACC3, B0, B1, B2, B3) \
  "v_fmac_f32 " ACC0 ", " A ", " B0 "\nv_fmac_f32 " ACC1 ", " A ", " B1 "\nv_fmac_f32 " ACC2 ", " A ", " B2 "\nv_fmac_f32 " ACC3 ", " A ", " B3 "\n"
DEF_KERNEL(k_step_pair_v4_indep,
           NZP4("0", "30", "31", "v44", "10", "13") NZP4("1", "32", "33", "v45", "14", "17")
           NZP4("2", "34", "35", "v46", "18", "21") NZP4("3", "36", "37", "v47", "22", "25")
           "s_waitcnt lgkmcnt(3)\n" FM4I("v31", "v26", "v27", "v28", "v29", "v10", "v11", "v12", "v13")
           "s_waitcnt lgkmcnt(2)\n" FM4I("v33", "v48", "v49", "v50", "v51", "v14", "v15", "v16", "v17")
           "s_waitcnt lgkmcnt(1)\n" FM4I("v35", "v52", "v53", "v54", "v55", "v18", "v19", "v20", "v21")
           "s_waitcnt lgkmcnt(0)\n" FM4I("v37", "v56", "v57", "v58", "v59", "v22", "v23", "v24", "v25"),
           V32, "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59")

// o3) the VALU half of o) alone (no LDS reads) and o4) the LDS half alone (address
//     forming + reads, no FMAs): how much of the 9.5 ns is imperfect overlap?
#define NZP4_NOLDS(U, PAIR_LO, PAIR_HI, ADDR) \
  "v_mov_b64_dpp v[" PAIR_LO ":" PAIR_HI "], v[6:7] row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_add_u32 " ADDR ", v" PAIR_LO ", v40\n"
#define DPPQ(U, PAIR_LO, PAIR_HI) \
  "v_mov_b64_dpp v[" PAIR_LO ":" PAIR_HI "], v[6:7] row_newbcast:" U " row_mask:0xf bank_mask:0xf\n"
// (the four broadcasts first, then the four adds, as the compiler orders the shipped loop)
DEF_KERNEL(k_step_valu_only,
           DPPQ("0", "30", "31") DPPQ("1", "32", "33") DPPQ("2", "34", "35") DPPQ("3", "36", "37")
           "v_add_u32 v44, v30, v40\nv_add_u32 v45, v32, v40\nv_add_u32 v46, v34, v40\nv_add_u32 v47, v36, v40\n"
           FM4("v31", "v10", "v11", "v12", "v13") FM4("v33", "v14", "v15", "v16", "v17")
           FM4("v35", "v18", "v19", "v20", "v21") FM4("v37", "v22", "v23", "v24", "v25"),
           V32, "v44", "v45", "v46", "v47")
DEF_KERNEL(k_step_lds_only,
           NZP4("0", "30", "31", "v44", "10", "13") NZP4("1", "32", "33", "v45", "14", "17")
           NZP4("2", "34", "35", "v46", "18", "21") NZP4("3", "36", "37", "v47", "22", "25")
           "s_waitcnt lgkmcnt(0)\n",
           V32, "v44", "v45", "v46", "v47")

// p) the same with 2 columns per lane (128-column tiles): ds_read_b64 + 2 v_fmac_f32 per nonzero
#define NZP2(U, PAIR_LO, PAIR_HI, ADDR, B0, B1) \
  "v_mov_b64_dpp v[" PAIR_LO ":" PAIR_HI "], v[6:7] row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_add_u32 " ADDR ", v" PAIR_LO ", v40\n" \
  "ds_read_b64 v[" B0 ":" B1 "], " ADDR "\n"
#define FM2(A, B0, B1) "v_fmac_f32 v26, " A ", " B0 "\nv_fmac_f32 v27, " A ", " B1 "\n"
DEF_KERNEL(k_step_pair_v2,
           NZP2("0", "30", "31", "v44", "10", "11") NZP2("1", "32", "33", "v45", "14", "15")
           NZP2("2", "34", "35", "v46", "18", "19") NZP2("3", "36", "37", "v47", "22", "23")
           "s_waitcnt lgkmcnt(3)\n" FM2("v31", "v10", "v11")
           "s_waitcnt lgkmcnt(2)\n" FM2("v33", "v14", "v15")
           "s_waitcnt lgkmcnt(1)\n" FM2("v35", "v18", "v19")
           "s_waitcnt lgkmcnt(0)\n" FM2("v37", "v22", "v23"),
           V32, "v44", "v45", "v46", "v47")

// o5) o3 with the 64-bit DPP broadcasts replaced by 32-bit ones (two per nonzero),
//     o6) by plain 64-bit moves: what do the DPP operations cost INSIDE the mix?
#define DPP2(U, PAIR_LO, PAIR_HI) \
  "v_mov_b32_dpp v" PAIR_LO ", v6 row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_mov_b32_dpp v" PAIR_HI ", v7 row_newbcast:" U " row_mask:0xf bank_mask:0xf\n"
DEF_KERNEL(k_step_valu_dpp32,
           DPP2("0", "30", "31") DPP2("1", "32", "33") DPP2("2", "34", "35") DPP2("3", "36", "37")
           "v_add_u32 v44, v30, v40\nv_add_u32 v45, v32, v40\nv_add_u32 v46, v34, v40\nv_add_u32 v47, v36, v40\n"
           FM4("v31", "v10", "v11", "v12", "v13") FM4("v33", "v14", "v15", "v16", "v17")
           FM4("v35", "v18", "v19", "v20", "v21") FM4("v37", "v22", "v23", "v24", "v25"),
           V32, "v44", "v45", "v46", "v47")
#define MOVQ(PAIR_LO, PAIR_HI) "v_mov_b64 v[" PAIR_LO ":" PAIR_HI "], v[6:7]\n"
DEF_KERNEL(k_step_valu_nodpp,
           MOVQ("30", "31") MOVQ("32", "33") MOVQ("34", "35") MOVQ("36", "37")
           "v_add_u32 v44, v30, v40\nv_add_u32 v45, v32, v40\nv_add_u32 v46, v34, v40\nv_add_u32 v47, v36, v40\n"
           FM4("v31", "v10", "v11", "v12", "v13") FM4("v33", "v14", "v15", "v16", "v17")
           FM4("v35", "v18", "v19", "v20", "v21") FM4("v37", "v22", "v23", "v24", "v25"),
           V32, "v44", "v45", "v46", "v47")
DEF_KERNEL(k_step_fma16_only,
           FM4("v31", "v10", "v11", "v12", "v13") FM4("v33", "v14", "v15", "v16", "v17")
           FM4("v35", "v18", "v19", "v20", "v21") FM4("v37", "v22", "v23", "v24", "v25"),
           V32)

// q) VGPR bank conflicts: the accumulator and the multiplicand in the SAME register bank
//    (index mod 4) against all three operands in different banks
DEF_KERNEL(k_fmac_bank_conflict,
           "v_fmac_f32 v10, v48, v26\n"
           "v_fmac_f32 v11, v49, v27\n"
           "v_fmac_f32 v12, v50, v28\n"
           "v_fmac_f32 v13, v51, v29\n"
           "v_fmac_f32 v14, v48, v30\n"
           "v_fmac_f32 v15, v49, v31\n"
           "v_fmac_f32 v16, v50, v32\n"
           "v_fmac_f32 v17, v51, v33\n"
           "v_fmac_f32 v18, v48, v34\n"
           "v_fmac_f32 v19, v49, v35\n"
           "v_fmac_f32 v20, v50, v36\n"
           "v_fmac_f32 v21, v51, v37\n"
           "v_fmac_f32 v22, v48, v38\n"
           "v_fmac_f32 v23, v49, v39\n"
           "v_fmac_f32 v24, v50, v40\n"
           "v_fmac_f32 v25, v51, v41\n",
           V16)
DEF_KERNEL(k_fmac_bank_free,
           "v_fmac_f32 v10, v49, v27\n"
           "v_fmac_f32 v11, v50, v28\n"
           "v_fmac_f32 v12, v51, v29\n"
           "v_fmac_f32 v13, v48, v30\n"
           "v_fmac_f32 v14, v49, v31\n"
           "v_fmac_f32 v15, v50, v32\n"
           "v_fmac_f32 v16, v51, v33\n"
           "v_fmac_f32 v17, v48, v34\n"
           "v_fmac_f32 v18, v49, v35\n"
           "v_fmac_f32 v19, v50, v36\n"
           "v_fmac_f32 v20, v51, v37\n"
           "v_fmac_f32 v21, v48, v38\n"
           "v_fmac_f32 v22, v49, v39\n"
           "v_fmac_f32 v23, v50, v40\n"
           "v_fmac_f32 v24, v51, v41\n"
           "v_fmac_f32 v25, v48, v42\n",
           V16)

// r) 8 columns per lane (512-column tiles): per nonzero ONE broadcast + add, TWO ds_read_b128
//    (second at +1 KiB) and 8 v_fmac_f32: does halving the broadcast work per FMA pay?
#define NZP8(U, PAIR_LO, PAIR_HI, ADDR, B0, B3, B4, B7) \
  "v_mov_b64_dpp v[" PAIR_LO ":" PAIR_HI "], v[6:7] row_newbcast:" U " row_mask:0xf bank_mask:0xf\n" \
  "v_add_u32 " ADDR ", v" PAIR_LO ", v40\n" \
  "ds_read_b128 v[" B0 ":" B3 "], " ADDR "\n" \
  "ds_read_b128 v[" B4 ":" B7 "], " ADDR " offset:1024\n"
#define FM8(A, B0, B1, B2, B3, B4, B5, B6, B7) \
  "v_fmac_f32 v26, " A ", " B0 "\nv_fmac_f32 v27, " A ", " B1 "\nv_fmac_f32 v28, " A ", " B2 "\nv_fmac_f32 v29, " A ", " B3 "\n" \
  "v_fmac_f32 v64, " A ", " B4 "\nv_fmac_f32 v65, " A ", " B5 "\nv_fmac_f32 v66, " A ", " B6 "\nv_fmac_f32 v67, " A ", " B7 "\n"
#define V8X "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67"
DEF_KERNEL(k_step_pair_v8,
           NZP8("0", "30", "31", "v44", "10", "13", "14", "17") NZP8("1", "32", "33", "v45", "18", "21", "22", "25")
           NZP8("2", "34", "35", "v46", "48", "51", "52", "55") NZP8("3", "36", "37", "v47", "56", "59", "60", "63")
           "s_waitcnt lgkmcnt(6)\n" FM8("v31", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17")
           "s_waitcnt lgkmcnt(4)\n" FM8("v33", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25")
           "s_waitcnt lgkmcnt(2)\n" FM8("v35", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")
           "s_waitcnt lgkmcnt(0)\n" FM8("v37", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"),
           V32, V8X)
// r2) the same with two nonzeros per wait group (half the B registers in flight)
DEF_KERNEL(k_step_pair_v8x2,
           NZP8("0", "30", "31", "v44", "10", "13", "14", "17") NZP8("1", "32", "33", "v45", "18", "21", "22", "25")
           "s_waitcnt lgkmcnt(2)\n" FM8("v31", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17")
           "s_waitcnt lgkmcnt(0)\n" FM8("v33", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25")
           NZP8("2", "34", "35", "v46", "48", "51", "52", "55") NZP8("3", "36", "37", "v47", "56", "59", "60", "63")
           "s_waitcnt lgkmcnt(2)\n" FM8("v35", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")
           "s_waitcnt lgkmcnt(0)\n" FM8("v37", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"),
           V32, V8X)
// r3) software pipelined: the reads of group g+1 are issued before the FMAs of group g
DEF_KERNEL(k_step_pair_v8_pipe,
           NZP8("2", "34", "35", "v46", "48", "51", "52", "55") NZP8("3", "36", "37", "v47", "56", "59", "60", "63")
           "s_waitcnt lgkmcnt(6)\n" FM8("v31", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17")
           "s_waitcnt lgkmcnt(4)\n" FM8("v33", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25")
           NZP8("0", "30", "31", "v44", "10", "13", "14", "17") NZP8("1", "32", "33", "v45", "18", "21", "22", "25")
           "s_waitcnt lgkmcnt(6)\n" FM8("v35", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")
           "s_waitcnt lgkmcnt(4)\n" FM8("v37", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"),
           V32, V8X)
// r4) 4-column step software pipelined the same way (for comparison with o))
DEF_KERNEL(k_step_pair_v4_pipe,
           NZP4("2", "34", "35", "v46", "18", "21") NZP4("3", "36", "37", "v47", "22", "25")
           "s_waitcnt lgkmcnt(3)\n" FM4("v31", "v10", "v11", "v12", "v13")
           "s_waitcnt lgkmcnt(2)\n" FM4("v33", "v14", "v15", "v16", "v17")
           NZP4("0", "30", "31", "v44", "10", "13") NZP4("1", "32", "33", "v45", "14", "17")
           "s_waitcnt lgkmcnt(3)\n" FM4("v35", "v18", "v19", "v20", "v21")
           "s_waitcnt lgkmcnt(2)\n" FM4("v37", "v22", "v23", "v24", "v25"),
           V32, "v44", "v45", "v46", "v47")

typedef void (*kern_t)(unsigned long long*, int);

static const char* g_filter = nullptr;
static void run(const char* name, kern_t k, int per_block_insts) {
  if (g_filter && !strstr(name, g_filter)) return;
  const int iters = 5000;
  unsigned long long* d;
  CHECK(hipMalloc(&d, sizeof(unsigned long long) * 256 * 16));
  for (int waves : {4, 8, 16}) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 0, 0, d, 1000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(256 * waves);
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += double(v);
    avg /= h.size();
    // s_memtime ticks at 100 MHz on this part; report both tick- and wall-derived figures
    const double insts_per_simd = double(iters) * 4 * per_block_insts * (waves / 4.0);
    printf("%-18s waves/SIMD=%d  wall=%.3f ms  ns/inst/SIMD=%.3f  memtime_ticks/wave=%.0f\n", name,
           waves / 4, ms, ms * 1e6 / insts_per_simd, avg);
  }
  {  // 8 waves per SIMD: two 16-wave workgroups per CU (64 KiB of LDS each)
    const int waves = 16, blocks = 512;
    unsigned long long* d2;
    CHECK(hipMalloc(&d2, sizeof(unsigned long long) * blocks * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 0, 0, d2, 1000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 0, 0, d2, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = double(iters) * 4 * per_block_insts * 8;
    printf("%-18s waves/SIMD=8  wall=%.3f ms  ns/inst/SIMD=%.3f\n", name, ms, ms * 1e6 / insts_per_simd);
    CHECK(hipFree(d2));
  }
  CHECK(hipFree(d));
}

int main(int argc, char** argv) {
  if (argc > 1) g_filter = argv[1];  // run only the kernels whose name contains this
  run("fma_vgpr", k_fma_vgpr, 16);
  run("fma_sgpr", k_fma_sgpr, 16);
  run("pkfma_vgpr", k_pkfma_vgpr, 16);
  run("pkfma_sgpr", k_pkfma_sgpr, 16);
  run("readlane", k_readlane, 16);
  run("lshl_add", k_lshl_add, 16);
  run("ds_read_b128", k_ds_read_b128, 16);
  run("spmm_step(4nz)", k_spmm_step, 4);
  run("spmm_step_fma(4nz)", k_spmm_step_fma, 4);
  run("fmac_dpp", k_fmac_dpp, 16);
  run("fmac_vop2", k_fmac_vop2, 16);
  run("spmm_step_dpp(4nz)", k_spmm_step_dpp, 4);
  run("spmm_step_dpp1(4nz)", k_spmm_step_dpp1, 4);
  run("salu", k_salu, 16);
  run("salu+valu(16)", k_salu_valu, 16);
  run("spmm_step_scalar(4nz)", k_spmm_step_scalar, 4);
  run("step_pair_v4(4nz)", k_step_pair_v4, 4);
  run("step_pair_v4_indep", k_step_pair_v4_indep, 4);
  run("step_valu_only(4nz)", k_step_valu_only, 4);
  run("step_lds_only(4nz)", k_step_lds_only, 4);
  run("step_valu_dpp32(4nz)", k_step_valu_dpp32, 4);
  run("step_valu_nodpp(4nz)", k_step_valu_nodpp, 4);
  run("step_fma16_only(4nz)", k_step_fma16_only, 4);
  run("step_pair_v2(4nz)", k_step_pair_v2, 4);
  run("fmac_bank_conflict", k_fmac_bank_conflict, 16);
  run("fmac_bank_free", k_fmac_bank_free, 16);
  run("mov_dpp32", k_mov_dpp32, 16);
  run("mov_dpp64", k_mov_dpp64, 16);
  run("step_pair_v8(4nz)", k_step_pair_v8, 4);
  run("step_pair_v8x2(4nz)", k_step_pair_v8x2, 4);
  run("step_pair_v8_pipe", k_step_pair_v8_pipe, 4);
  run("step_pair_v4_pipe", k_step_pair_v4_pipe, 4);
  return 0;
}
