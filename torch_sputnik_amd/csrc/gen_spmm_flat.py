#!/usr/bin/env python3
"""Generates spmm_flat_body.inc: the main loop of the flat-stream SpMM kernel
(spmm_flat.hip) as one gfx950 inline-asm text.

Why generated assembly.  The loop keeps the accumulators of a wave's rows in
FIXED vector registers and picks the row of every nonzero with the VGPR index
mode (s_set_gpr_idx_on: destination and src2 of the FMAs become M0-relative).
That makes the instruction stream independent of the row a nonzero belongs to:
a wave walks ONE flat stream of entries per K chunk -- all of its rows' entries
back to back -- in a software pipeline (LDS reads of entry j+3 in flight while
entry j is multiplied), with no per-(row, chunk) bookkeeping at all.  hipcc has
no notion of an M0-relative register, hence the fixed register map below and a
generator instead of C++.

Structure (see spmm_flat.hip for the data the loop reads):

  window   16 consecutive stream entries, one per lane of every 16-lane row:
           (LDS byte offset of the B row, value) as a register pair plus the
           packed row bytes; three window register sets rotate (current /
           value gather in flight / plan load in flight), so the slot code
           exists three times, once per register set
  slot s   entry s of the current window: chunk-boundary test, DPP broadcast of
           the pair, address add, two ds_read_b128 into strip set s % 4, then
           the four v_pk_fma_f32 of the entry issued three slots earlier
  boundary a subroutine (s_swappc_b64): the wave has issued all entries of K
           chunk c -> LDS reads drained, next B tile landed, workgroup barrier,
           next tile's LDS-DMA copies issued
  vmcnt    window loads and B copies share the counter and their interleaving
           depends on the data, so the two waits pick their immediate from what
           was issued since (s_sb: B batches since the last window switch,
           s_nsw: window switches since the last boundary); both choices are
           conservative when clamped

Register map (everything is pinned; the kernel passes its inputs in exactly
these registers):

  v0        lane * 16 + LDS address of the tile       (input)
  v1        byte offset of the lane's 16 bytes inside a B row, for the copies (input)
  v2, v3    byte offsets of the lane's plan words inside the stream (input, advanced here)
  v[4:6] v[8:10] v[12:14]   the three windows: offset, value (first: value's byte offset), row bytes
  v[16:23]  four broadcast pairs (LDS address, value)
  v[32:63]  four strips of 8 floats
  v[64:127] accumulators, row r at v[64 + 8 r ...]   (outputs)
  s[36:37] stream  s[38:39] values  s[40:41] ends  s[42:43] dense
  s44 row pitch of B in bytes  s45 k - 1  s46 chunks  s47 first B row of the
  wave inside a chunk  s48 LDS address of the wave's first piece, buffer 0   (inputs)
  s50.. scratch (see names below)
"""
import sys

BK = 32          # rows of B per chunk
WIN_BYTES = 144  # 16 x (offset, value offset) + 16 row bytes
NSTRIP = 4       # software pipeline depth (strip sets)

V_LANE, V_STAGE, V_PLAN, V_ROWS = "v0", "v1", "v2", "v3"
WIN = [4, 8, 12]                    # first register of each window set
PAIR = [16, 18, 20, 22]
STRIP = [32, 40, 48, 56]
ACC = 64

S_STREAM, S_VALUES, S_ENDS, S_DENSE = "s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]"
S_PITCH, S_KMAX, S_NCHUNKS, S_ROW0, S_LDS0, S_DEBUG = "s44", "s45", "s46", "s47", "s48", "s49"
S_REM, S_C, S_ENDCUR, S_ENDNEXT, S_SB, S_NSW, S_ENDOFF = "s50", "s51", "s52", "s53", "s54", "s55", "s56"
S_RET, S_BND = "s[58:59]", "s[60:61]"
S_ROWS = [62, 63, 64, 65]
S_IDX = [66, 67, 68, 69]
S_T = ["s70", "s71", "s72", "s73", "s74", "s75"]

out = []


def emit(line):
    out.append(line)


def label(name):
    return f"L_{name}_%="


def fma_stage(ring, lgkm):
    p, s = PAIR[ring], STRIP[ring]
    emit(f"s_waitcnt lgkmcnt({lgkm})")
    emit(f"s_set_gpr_idx_on s{S_IDX[ring]}, 0xc")
    for q in range(4):
        emit(f"v_pk_fma_f32 v[{ACC + 2 * q}:{ACC + 2 * q + 1}], v[{p}:{p + 1}], "
             f"v[{s + 2 * q}:{s + 2 * q + 1}], v[{ACC + 2 * q}:{ACC + 2 * q + 1}] "
             f"op_sel:[1,0,0] op_sel_hi:[1,1,1]")
    emit("s_set_gpr_idx_off")


def issue_stage(win, slot):
    ring = slot % NSTRIP
    w, p, s = WIN[win], PAIR[ring], STRIP[ring]
    emit(f"v_mov_b64_dpp v[{p}:{p + 1}], v[{w}:{w + 1}] row_newbcast:{slot} row_mask:0xf bank_mask:0xf")
    emit(f"s_bfe_u32 s{S_IDX[ring]}, s{S_ROWS[slot // 4]}, {hex(8 * (slot % 4) | (8 << 16))}")
    emit(f"v_add_u32 v{p}, v{p}, {V_LANE}")
    emit(f"ds_read_b128 v[{s}:{s + 3}], v{p}")
    emit(f"ds_read_b128 v[{s + 4}:{s + 7}], v{p} offset:1024")


def stage_chunk(chunk_sgpr):
    """LDS-DMA copies of this wave's four pieces of chunk `chunk_sgpr` (< chunks)."""
    t_row, t_lds, t_r, t_lo, t_hi = S_T[1], S_T[2], S_T[3], S_T[4], S_T[5]
    emit(f"s_lshl_b32 {t_row}, {chunk_sgpr}, 5")
    emit(f"s_add_u32 {t_row}, {t_row}, {S_ROW0}")
    emit(f"s_and_b32 {t_lds}, {chunk_sgpr}, 1")
    emit(f"s_lshl_b32 {t_lds}, {t_lds}, 16")
    emit(f"s_add_u32 {t_lds}, {t_lds}, {S_LDS0}")
    for i in range(4):
        emit(f"s_add_u32 {t_r}, {t_row}, {8 * i}")
        emit(f"s_min_u32 {t_r}, {t_r}, {S_KMAX}")
        emit(f"s_mul_hi_u32 {t_hi}, {t_r}, {S_PITCH}")
        emit(f"s_mul_i32 {t_lo}, {t_r}, {S_PITCH}")
        emit(f"s_add_u32 {t_lo}, {t_lo}, s42")
        emit(f"s_addc_u32 {t_hi}, {t_hi}, s43")
        emit(f"s_add_u32 m0, {t_lds}, {hex(8 * i * 2048)}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 {V_STAGE}, s[{t_lo[1:]}:{t_hi[1:]}]")


def plan_load(win):
    w = WIN[win]
    emit(f"global_load_dwordx2 v[{w}:{w + 1}], {V_PLAN}, {S_STREAM}")
    emit(f"global_load_dword v{w + 2}, {V_ROWS}, {S_STREAM}")
    emit(f"v_add_u32 {V_PLAN}, {hex(WIN_BYTES)}, {V_PLAN}")
    emit(f"v_add_u32 {V_ROWS}, {hex(WIN_BYTES)}, {V_ROWS}")


def gather(win):
    w = WIN[win]
    emit(f"global_load_dword v{w + 1}, v{w + 1}, {S_VALUES}")


def switch_to(win, tag):
    """Window set `win` becomes current: its value gather and the plan load of
    the set after it must have landed; start that set's gather and the plan
    load into the set that was current until now."""
    nxt, old = (win + 1) % 3, (win + 2) % 3
    emit(f"s_cmp_lg_u32 {S_SB}, 0")
    emit(f"s_cbranch_scc1 {label('sw4_' + tag)}")
    emit("s_waitcnt vmcnt(0)")
    emit(f"s_branch {label('swd_' + tag)}")
    emit(f"{label('sw4_' + tag)}:")
    emit("s_waitcnt vmcnt(4)")
    emit(f"{label('swd_' + tag)}:")
    emit(f"s_mov_b32 {S_SB}, 0")
    emit(f"s_add_u32 {S_NSW}, {S_NSW}, 1")
    for i in range(4):
        emit(f"v_readlane_b32 s{S_ROWS[i]}, v{WIN[win] + 2}, {i}")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 3")        # timing experiment: no window loads
    emit(f"s_cbranch_scc1 {label('swskip_' + tag)}")
    gather(nxt)
    plan_load(old)
    emit(f"{label('swskip_' + tag)}:")


def generate():
    emit("; ---- prologue ----")
    for r in list(range(ACC, ACC + 64)) + list(range(STRIP[0], STRIP[0] + 32)) + list(range(PAIR[0], PAIR[0] + 8)):
        emit(f"v_mov_b32 v{r}, 0")
    for r in S_IDX:
        emit(f"s_mov_b32 s{r}, 0")
    emit(f"s_getpc_b64 {S_BND}")
    emit(f"{label('pc')}:")
    emit(f"s_add_u32 s60, s60, {label('boundary')}-{label('pc')}")
    emit("s_addc_u32 s61, s61, 0")
    plan_load(0)
    plan_load(1)
    emit("s_waitcnt vmcnt(2)")       # plan of window 0 landed
    gather(0)
    emit(f"s_mov_b32 {S_T[0]}, 0")
    stage_chunk(S_T[0])
    emit("s_waitcnt vmcnt(0)")
    emit("s_barrier")
    emit(f"s_mov_b32 {S_C}, 0")
    emit(f"s_mov_b32 {S_SB}, 0")
    emit(f"s_mov_b32 {S_NSW}, 0")
    emit(f"s_cmp_lt_u32 {S_NCHUNKS}, 2")
    emit(f"s_cbranch_scc1 {label('pro_nostage')}")
    emit(f"s_mov_b32 {S_T[0]}, 1")
    stage_chunk(S_T[0])
    emit(f"s_mov_b32 {S_SB}, 1")
    emit(f"{label('pro_nostage')}:")
    emit(f"s_load_dwordx2 s[52:53], {S_ENDS}, 0x0")
    emit(f"s_mov_b32 {S_ENDOFF}, 8")
    emit("s_waitcnt lgkmcnt(0)")
    emit(f"s_mov_b32 {S_REM}, {S_ENDCUR}")
    # window 0 becomes current (its gather and window 1's plan landed: the copies
    # of chunk 1, if any, are the newest operations)
    # (s_nsw stays 1: this switch's three loads are newer than the copies of chunk 1)
    switch_to(0, "pro")

    emit("; ---- main loop ----")
    for win in range(3):
        for slot in range(16):
            ring = slot % NSTRIP
            emit(f"{label(f'slot_{win}_{slot}')}:")
            emit(f"s_sub_u32 {S_REM}, {S_REM}, 1")
            emit(f"s_cbranch_scc1 {label(f'bnd_{win}_{slot}')}")
            issue_stage(win, slot)
            fma_stage((ring + 1) % NSTRIP, 2 * (NSTRIP - 1))
        switch_to((win + 1) % 3, f"w{win}")
        if win == 2:
            emit(f"s_branch {label('slot_0_0')}")

    emit("; ---- boundary stubs ----")
    for win in range(3):
        for slot in range(16):
            emit(f"{label(f'bnd_{win}_{slot}')}:")
            emit(f"s_swappc_b64 {S_RET}, {S_BND}")
            emit(f"s_cbranch_scc1 {label(f'drain_{slot % NSTRIP}')}")
            emit(f"s_branch {label(f'slot_{win}_{slot}')}")

    emit("; ---- boundary subroutine ----")
    emit(f"{label('boundary')}:")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 2")        # timing experiment: no drain of the LDS reads
    emit(f"s_cbranch_scc1 {label('bnodrain')}")
    emit("s_waitcnt lgkmcnt(0)")
    emit(f"{label('bnodrain')}:")
    for n in range(3):
        emit(f"s_cmp_eq_u32 {S_NSW}, {n}")
        emit(f"s_cbranch_scc1 {label(f'bv{n}')}")
    emit("s_waitcnt vmcnt(9)")
    emit(f"s_branch {label('bvd')}")
    for n in range(3):
        emit(f"{label(f'bv{n}')}:")
        emit(f"s_waitcnt vmcnt({3 * n})")
        if n < 2:
            emit(f"s_branch {label('bvd')}")
    emit(f"{label('bvd')}:")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 1")        # timing experiment: no rendezvous
    emit(f"s_cbranch_scc1 {label('bnobar')}")
    emit("s_barrier")
    emit(f"{label('bnobar')}:")
    emit(f"s_add_u32 {S_C}, {S_C}, 1")
    emit(f"s_cmp_eq_u32 {S_C}, {S_NCHUNKS}")
    emit(f"s_cbranch_scc1 {label('bret')}")
    emit(f"s_sub_u32 {S_REM}, {S_ENDNEXT}, {S_ENDCUR}")
    emit(f"s_mov_b32 {S_ENDCUR}, {S_ENDNEXT}")
    emit(f"s_load_dword {S_ENDNEXT}, {S_ENDS}, {S_ENDOFF}")
    emit(f"s_add_u32 {S_ENDOFF}, {S_ENDOFF}, 4")
    emit(f"s_mov_b32 {S_NSW}, 0")
    emit(f"s_add_u32 {S_T[0]}, {S_C}, 1")
    emit(f"s_cmp_ge_u32 {S_T[0]}, {S_NCHUNKS}")
    emit(f"s_cbranch_scc1 {label('bnostage')}")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 0")        # timing experiment: no copies of B
    emit(f"s_cbranch_scc1 {label('bnostage')}")
    stage_chunk(S_T[0])
    emit(f"s_add_u32 {S_SB}, {S_SB}, 1")
    emit(f"{label('bnostage')}:")
    emit(f"s_cmp_eq_u32 {S_C}, {S_NCHUNKS}")   # SCC = 0: not finished
    emit(f"{label('bret')}:")
    emit(f"s_setpc_b64 {S_RET}")

    emit("; ---- drain: the three entries still in the pipeline ----")
    for ring in range(NSTRIP):
        emit(f"{label(f'drain_{ring}')}:")
        for j in range(1, NSTRIP):
            fma_stage((ring + j) % NSTRIP, 2 * (NSTRIP - 1 - j))
        if ring < NSTRIP - 1:
            emit(f"s_branch {label('end')}")
    emit(f"{label('end')}:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")


def main():
    generate()
    path = sys.argv[1] if len(sys.argv) > 1 else None
    text = "// GENERATED by gen_spmm_flat.py -- do not edit.\n" + "".join(
        f'"{line}\\n"\n' for line in out)
    if path:
        with open(path, "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
