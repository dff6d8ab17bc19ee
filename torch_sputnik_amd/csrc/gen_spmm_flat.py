#!/usr/bin/env python3
"""Generates the main loop of the flat-stream SpMM kernel (spmm_flat.hip) as
gfx950 inline-asm text: spmm_flat_body_<variant>.inc.

Why generated assembly.  The loop keeps the accumulators of a wave's 8 rows in
FIXED vector registers and picks the row of every nonzero with the VGPR index
mode (s_set_gpr_idx_on: destination and src2 of the FMAs become M0-relative).
That makes the instruction stream independent of the row a nonzero belongs to:
a wave walks ONE flat stream of entries -- all of its rows' entries of a K
chunk back to back, sorted by (column, row) -- with no per-(row, chunk)
bookkeeping at all.  hipcc has no notion of an M0-relative register, hence the
fixed register map below and a generator instead of C++.

Two loops over the same stream (--mode):

  entry   every entry reads its B strip (two ds_read_b128) three entries ahead
          of its four v_pk_fma_f32, in a ring of four strip sets; no taken branch
          in the steady state.  One B dword from LDS per FMA: bound by the LDS
          (17 ns per entry and SIMD on random data, tools/gpridx_bench.hip).
  group   the entries of one column of B that several of the wave's rows hold
          follow each other in the stream (a "column group"), and the strip is
          read ONCE per group, three groups ahead: 0.57 reads per entry at density
          0.1, 0.25 at 0.5; per entry one 32-bit value broadcast and the FMAs
          (13.6 / 11.6 ns).  Which strip is current is a state of the program
          counter (four copies of the slot code), and a group start costs taken
          branches, tens of cycles each: the dense half of the benchmark's sweep
          gains (density 0.25 +10 %, 0.5 +17 %), the sparse half loses, so the
          dispatcher picks the loop by density (spmm_flat.hip).
          --layout: which path falls through.  With p = the share of entries that
          start a group, "straight" (a group start jumps out and back into the
          next copy) takes 2 p branches per entry, "diagonal" (the path "every
          entry starts a group" falls through; an entry that stays in its group
          jumps) 1 - p.

Common structure (see spmm_flat.hip for the data the loop reads):

  window   16 consecutive stream entries, one per lane of every 16-lane row: a
           packed word (byte 0: tile row of the entry's own column, byte 1: tile
           row of the column group three ahead, meaningful in the first entry of
           a group), the value, and the packed row bytes (bit 7: first entry of a
           column group, bits 0-6: accumulator index 8 * row).  "current" plus
           two prefetch stages (value gather in flight / plan load in flight)
           that are copied down at a window switch.
  boundary a subroutine (s_swappc_b64) reached when the chunk's entries / groups
           are used up: LDS reads drained, next B tile landed, workgroup barrier,
           the copies of the tile after it (all four pieces at once: spreading
           them over the chunk's entries was measured twice, 5-10 % slower).
  vmcnt    window loads and B copies share the counter and their interleaving
           depends on the data, so the two waits pick their immediate from what
           was issued since (s_sb: copy batches since the last window switch,
           s_nsw: window switches since the last boundary); both choices are
           conservative when clamped.

Register map (everything is pinned; the kernel passes its inputs in exactly
these registers):

  v0        lane * 16 + LDS address of the tile       (input)
  v1        byte offset of the lane's 16 bytes inside a B row, for the copies (input)
  v2, v3    byte offsets of the lane's plan words inside the stream (input, advanced here)
  v[4:5]    current window: LDS offset (own column / group ahead), value
  v[6:7] v8        next window: packed word, value (gather in flight), row bytes
  v[10:11] v12     the window after it: packed word, value's byte offset, row bytes (load in flight)
  v[16:23]  entry mode: four (LDS address, value) pairs; group mode: v[16:17] v[18:19]
            value broadcasts, v20 LDS address of a group
  v[32:63]  four strips of 8 floats
  v[64:127] accumulators, row r at v[64 + 8 r ...]   (outputs)
  s[36:37] stream  s[38:39] values  s[40:41] chunk info  s[42:43] dense
  s44 row pitch of B in bytes  s45 k - 1  s46 chunks  s47 first B row of the
  wave inside a chunk  s48 LDS address of the wave's first piece, buffer 0
  s49 timing-experiment bits   (inputs)
  s50.. scratch (see names below)
"""
import argparse

_ap = argparse.ArgumentParser()
_ap.add_argument("--mode", default="entry", choices=["entry", "group"])
_ap.add_argument("--layout", default="straight", choices=["straight", "diagonal"])
_ap.add_argument("out", nargs="?")
ARGS = _ap.parse_args()
GROUP = ARGS.mode == "group"

BK = 32            # rows of B per chunk: a stage is 64 KiB
BK_SHIFT = 5
ROW_BYTES = 2048   # bytes of a tile row (512 columns)
ROW_SHIFT = 11
WIN_BYTES = 144    # 16 x (packed word, value offset) + 16 row bytes
NSTRIP = 4         # strips in the ring (reads are issued NSTRIP - 1 entries / groups ahead)
NPIECES = 4        # 1 KiB pieces of a B tile that a wave copies (32 rows x 2 pieces / 16 waves)
ROWSTEP = 8        # B rows between a wave's pieces

V_LANE, V_STAGE, V_PLAN, V_ROWS = "v0", "v1", "v2", "v3"
CUR_OFF, CUR_VAL = 4, 5
NXT_OFF, NXT_VAL, NXT_ROWS = 6, 7, 8
NN_OFF, NN_VAL, NN_ROWS = 10, 11, 12
PAIR = [16, 18, 20, 22]     # entry mode
VALT = [16, 18]             # group mode
V_ADDR = 20                 # group mode
STRIP = [32, 40, 48, 56]
ACC = 64

S_STREAM, S_VALUES, S_CINFO, S_DENSE = "s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]"
S_PITCH, S_KMAX, S_NCHUNKS, S_ROW0, S_LDS0, S_DEBUG = "s44", "s45", "s46", "s47", "s48", "s49"
S_REM, S_C, S_CIOFF, S_MINE, S_SB, S_NSW = "s50", "s51", "s56", "s57", "s76", "s77"
S_CI = ["s52", "s53", "s54", "s55"]   # the NEXT chunk's info: groups, first group rows, unused, entries
S_RET, S_BND = "s[58:59]", "s[60:61]"
S_ROWS = [62, 63, 64, 65]
S_IDX = [66, 67, 68, 69]   # entry mode: ring of accumulator indices; group mode: the first one
S_COLS = "s78"             # group mode: the new chunk's first three group rows while their reads are issued
S_T = ["s70", "s71", "s72", "s73", "s74", "s75"]

out = []


def emit(line):
    out.append(line)


def label(name):
    return f".L_{name}_%="


def strip_reads(ring, addr):
    s = STRIP[ring]
    emit(f"ds_read_b128 v[{s}:{s + 3}], v{addr}")
    emit(f"ds_read_b128 v[{s + 4}:{s + 7}], v{addr} offset:1024")


def fmas(src_pair, op_sel, ring):
    s = STRIP[ring]
    for q in range(4):
        emit(f"v_pk_fma_f32 v[{ACC + 2 * q}:{ACC + 2 * q + 1}], v[{src_pair}:{src_pair + 1}], "
             f"v[{s + 2 * q}:{s + 2 * q + 1}], v[{ACC + 2 * q}:{ACC + 2 * q + 1}] {op_sel}")


# ---- entry mode -----------------------------------------------------------
def entry_fma_stage(ring, lgkm):
    """The FMAs of the entry whose pair and strip are in ring position `ring`
    (value = HIGH half of the pair)."""
    emit(f"s_waitcnt lgkmcnt({lgkm})")
    emit(f"s_set_gpr_idx_on s{S_IDX[ring]}, 0xc")
    fmas(PAIR[ring], "op_sel:[1,0,0] op_sel_hi:[1,1,1]", ring)
    emit("s_set_gpr_idx_off")


def entry_slot(slot):
    ring = slot % NSTRIP
    p = PAIR[ring]
    emit(f"{label(f'E_{slot}')}:")
    emit(f"s_sub_u32 {S_REM}, {S_REM}, 1")
    emit(f"s_cbranch_scc1 {label(f'bnd_{slot}')}")
    emit(f"v_mov_b64_dpp v[{p}:{p + 1}], v[{CUR_OFF}:{CUR_VAL}] row_newbcast:{slot} row_mask:0xf bank_mask:0xf")
    emit(f"s_bfe_u32 s{S_IDX[ring]}, s{S_ROWS[slot // 4]}, {hex(8 * (slot % 4) | (7 << 16))}")
    emit(f"v_add_u32 v{p}, v{p}, {V_LANE}")
    strip_reads(ring, p)
    entry_fma_stage((ring + 1) % NSTRIP, 2 * (NSTRIP - 1))


# ---- group mode -----------------------------------------------------------
def group_entry_body(ring, slot):
    t = VALT[slot % 2]
    emit(f"v_mov_b32_dpp v{t}, v{CUR_VAL} row_newbcast:{slot} row_mask:0xf bank_mask:0xf")
    emit(f"s_bfe_u32 s{S_IDX[0]}, s{S_ROWS[slot // 4]}, {hex(8 * (slot % 4) | (7 << 16))}")
    emit(f"s_set_gpr_idx_on s{S_IDX[0]}, 0xc")
    fmas(t, "op_sel:[0,0,0] op_sel_hi:[0,1,1]", ring)
    emit("s_set_gpr_idx_off")


def group_start(ring, slot):
    """First entry of a column group while strip `ring` is current: the chunk's
    groups may be used up (-> boundary); else the reads of the group NSTRIP - 1
    ahead go into strip `ring`, whose group is finished, and strip ring + 1,
    waited for here, becomes current."""
    emit(f"{label(f'N_{ring}_{slot}')}:")
    emit(f"s_sub_u32 {S_REM}, {S_REM}, 1")
    emit(f"s_cbranch_scc1 {label(f'bnd_{slot}')}")
    emit(f"v_mov_b32_dpp v{V_ADDR}, v{CUR_OFF} row_newbcast:{slot} row_mask:0xf bank_mask:0xf")
    emit(f"v_add_u32 v{V_ADDR}, v{V_ADDR}, {V_LANE}")
    strip_reads(ring, V_ADDR)
    emit(f"s_waitcnt lgkmcnt({2 * (NSTRIP - 1)})")


def first_groups():
    """Reads of the first three column groups of the chunk whose tile rows (a
    byte each) are in S_COLS, into strips 0..2."""
    for i in range(NSTRIP - 1):
        emit(f"s_bfe_u32 {S_T[0]}, {S_COLS}, {hex(8 * i | (8 << 16))}")
        emit(f"s_lshl_b32 {S_T[0]}, {S_T[0]}, {ROW_SHIFT}")
        emit(f"v_add_u32 v{V_ADDR}, {S_T[0]}, {V_LANE}")
        strip_reads(i, V_ADDR)


# ---- common ---------------------------------------------------------------
def stage_chunk(chunk_sgpr):
    """LDS-DMA copies of this wave's pieces of chunk `chunk_sgpr` (< chunks):
    B rows chunk * 32 + row0 + 8 i, clamped to the last row of B in the last
    (possibly partial) chunk -- no nonzero refers to the rows past it."""
    t_row, t_lds, t_r, t_lo, t_hi = S_T[1], S_T[2], S_T[3], S_T[4], S_T[5]
    emit(f"s_lshl_b32 {t_row}, {chunk_sgpr}, {BK_SHIFT}")
    emit(f"s_add_u32 {t_row}, {t_row}, {S_ROW0}")
    emit(f"s_and_b32 {t_lds}, {chunk_sgpr}, 1")
    emit(f"s_lshl_b32 {t_lds}, {t_lds}, 16")
    emit(f"s_add_u32 {t_lds}, {t_lds}, {S_LDS0}")
    for i in range(NPIECES):
        emit(f"s_add_u32 {t_r}, {t_row}, {ROWSTEP * i}")
        emit(f"s_min_u32 {t_r}, {t_r}, {S_KMAX}")
        emit(f"s_mul_hi_u32 {t_hi}, {t_r}, {S_PITCH}")
        emit(f"s_mul_i32 {t_lo}, {t_r}, {S_PITCH}")
        emit(f"s_add_u32 {t_lo}, {t_lo}, s42")
        emit(f"s_addc_u32 {t_hi}, {t_hi}, s43")
        emit(f"s_add_u32 m0, {t_lds}, {hex(ROWSTEP * i * ROW_BYTES)}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 {V_STAGE}, s[{t_lo[1:]}:{t_hi[1:]}]")


def plan_load(off, rows):
    emit(f"global_load_dwordx2 v[{off}:{off + 1}], {V_PLAN}, {S_STREAM}")
    emit(f"global_load_dword v{rows}, {V_ROWS}, {S_STREAM}")
    emit(f"v_add_u32 {V_PLAN}, {hex(WIN_BYTES)}, {V_PLAN}")
    emit(f"v_add_u32 {V_ROWS}, {hex(WIN_BYTES)}, {V_ROWS}")


def gather(val):
    emit(f"global_load_dword v{val}, v{val}, {S_VALUES}")


def switch_window(tag):
    """The next window becomes current: its value gather and the plan load of the
    window after it must have landed (the newest operations are the copies of a B
    tile, if a boundary came since the last switch)."""
    emit(f"s_cmp_lg_u32 {S_SB}, 0")
    emit(f"s_cbranch_scc1 {label('swc_' + tag)}")
    emit("s_waitcnt vmcnt(0)")
    emit(f"s_branch {label('swd_' + tag)}")
    emit(f"{label('swc_' + tag)}:")
    emit(f"s_waitcnt vmcnt({NPIECES})")
    emit(f"{label('swd_' + tag)}:")
    emit(f"s_mov_b32 {S_SB}, 0")
    emit(f"s_add_u32 {S_NSW}, {S_NSW}, 1")
    for i in range(4):
        emit(f"v_readlane_b32 s{S_ROWS[i]}, v{NXT_ROWS}, {i}")
    # LDS offset of the column the loop reads for an entry: its own (entry mode,
    # byte 0 of the packed word) or the group three ahead (group mode, byte 1)
    emit(f"v_bfe_u32 v{CUR_OFF}, v{NXT_OFF}, {8 if GROUP else 0}, 8")
    emit(f"v_mov_b32 v{CUR_VAL}, v{NXT_VAL}")
    emit(f"v_lshlrev_b32 v{CUR_OFF}, {ROW_SHIFT}, v{CUR_OFF}")
    emit(f"v_mov_b64 v[{NXT_OFF}:{NXT_VAL}], v[{NN_OFF}:{NN_VAL}]")
    emit(f"v_mov_b32 v{NXT_ROWS}, v{NN_ROWS}")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 3")        # timing experiment: no window loads
    emit(f"s_cbranch_scc1 {label('swskip_' + tag)}")
    gather(NXT_VAL)
    plan_load(NN_OFF, NN_ROWS)
    emit(f"{label('swskip_' + tag)}:")
    emit("s_nop 1")                            # VALU write of the window -> DPP read: 2 wait states


def take_chunk_info():
    """The chunk that starts: its count into S_REM (entries or groups), its first
    group rows into S_COLS."""
    emit(f"s_mov_b32 {S_REM}, {S_CI[0] if GROUP else S_CI[3]}")
    if GROUP:
        emit(f"s_mov_b32 {S_COLS}, {S_CI[1]}")


def generate():
    emit("; ---- prologue ----")
    for r in list(range(ACC, ACC + 64)) + list(range(STRIP[0], STRIP[0] + 32)) + list(range(16, 24)):
        emit(f"v_mov_b32 v{r}, 0")
    for r in S_IDX:
        emit(f"s_mov_b32 s{r}, 0")
    emit(f"s_getpc_b64 {S_BND}")
    emit(f"{label('pc')}:")
    emit(f"s_add_u32 s60, s60, {label('boundary')}-{label('pc')}")
    emit("s_addc_u32 s61, s61, 0")
    # windows: the plan of window 0 into "next", of window 1 into the stage behind it
    plan_load(NXT_OFF, NXT_ROWS)
    plan_load(NN_OFF, NN_ROWS)
    emit("s_waitcnt vmcnt(2)")
    gather(NXT_VAL)
    emit(f"s_mov_b32 {S_T[0]}, 0")
    stage_chunk(S_T[0])
    emit(f"s_load_dwordx4 s[52:55], {S_CINFO}, 0x0")   # chunk 0
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_barrier")
    take_chunk_info()
    emit(f"s_load_dwordx4 s[52:55], {S_CINFO}, 0x10")  # chunk 1 (waited for at the first boundary)
    emit(f"s_mov_b32 {S_CIOFF}, 32")
    emit(f"s_mov_b32 {S_C}, 0")
    emit(f"s_mov_b32 {S_NSW}, 0")
    emit(f"s_mov_b32 {S_SB}, 0")
    emit(f"s_mov_b32 {S_MINE}, 0")
    if GROUP:
        first_groups()
    emit(f"s_cmp_lt_u32 {S_NCHUNKS}, 2")
    emit(f"s_cbranch_scc1 {label('pro_nostage')}")
    emit(f"s_mov_b32 {S_T[0]}, 1")
    stage_chunk(S_T[0])
    emit(f"s_mov_b32 {S_SB}, 1")
    emit(f"s_mov_b32 {S_MINE}, 1")
    emit(f"{label('pro_nostage')}:")
    # window 0 becomes current (its gather and window 1's plan landed before the
    # barrier; the copies of chunk 1, if any, are the newest operations)
    switch_window("pro")

    emit("; ---- main loop ----")
    if not GROUP:
        for slot in range(16):
            entry_slot(slot)
        switch_window("w")
        emit(f"s_branch {label('E_0')}")
    elif ARGS.layout == "straight":
        emit(f"s_branch {label(f'E_{NSTRIP - 1}_0')}")
        for ring in range(NSTRIP):
            for slot in range(16):
                emit(f"{label(f'E_{ring}_{slot}')}:")
                emit(f"s_bitcmp1_b32 s{S_ROWS[slot // 4]}, {8 * (slot % 4) + 7}")
                emit(f"s_cbranch_scc1 {label(f'N_{ring}_{slot}')}")
                emit(f"{label(f'B_{ring}_{slot}')}:")
                group_entry_body(ring, slot)
            switch_window(f"r{ring}")
            emit(f"s_branch {label(f'E_{ring}_0')}")
        emit("; ---- first entry of a column group ----")
        for ring in range(NSTRIP):
            for slot in range(16):
                group_start(ring, slot)
                emit(f"s_branch {label(f'B_{(ring + 1) % NSTRIP}_{slot}')}")
    else:
        emit(f"s_branch {label(f'E_{NSTRIP - 1}_0')}")
        for diag in range(NSTRIP):
            for slot in range(16):
                ring = (diag + slot) % NSTRIP
                emit(f"{label(f'E_{ring}_{slot}')}:")
                emit(f"s_bitcmp1_b32 s{S_ROWS[slot // 4]}, {8 * (slot % 4) + 7}")
                emit(f"s_cbranch_scc0 {label(f'B_{ring}_{slot}')}")
                group_start(ring, slot)
                emit(f"{label(f'B_{(ring + 1) % NSTRIP}_{slot}')}:")
                group_entry_body((ring + 1) % NSTRIP, slot)
            ring = (diag + 16) % NSTRIP       # current behind slot 15
            switch_window(f"r{diag}")
            emit(f"s_branch {label(f'E_{ring}_0')}")

    emit("; ---- boundary stubs ----")
    for slot in range(16):
        emit(f"{label(f'bnd_{slot}')}:")
        emit(f"s_swappc_b64 {S_RET}, {S_BND}")
        if GROUP:
            emit(f"s_cbranch_scc1 {label('end')}")
            # strips 0..2 hold the new chunk's first groups: "strip 3 was current"
            emit(f"s_branch {label(f'N_{NSTRIP - 1}_{slot}')}")
        else:
            emit(f"s_cbranch_scc1 {label(f'drain_{slot % NSTRIP}')}")
            emit(f"s_branch {label(f'E_{slot}')}")

    emit("; ---- boundary subroutine ----")
    emit(f"{label('boundary')}:")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 2")        # timing experiment: no drain of the LDS reads
    emit(f"s_cbranch_scc1 {label('bnodrain')}")
    emit("s_waitcnt lgkmcnt(0)")
    emit(f"{label('bnodrain')}:")
    emit(f"s_cmp_eq_u32 {S_MINE}, 0")          # no copies of the next chunk pending
    emit(f"s_cbranch_scc1 {label('bvd')}")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 5")        # timing experiment: do not wait for the copies
    emit(f"s_cbranch_scc1 {label('bvd')}")
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
    take_chunk_info()
    emit(f"s_load_dwordx4 s[52:55], {S_CINFO}, {S_CIOFF}")   # the chunk after this one
    emit(f"s_add_u32 {S_CIOFF}, {S_CIOFF}, 16")
    if GROUP:
        first_groups()
    emit(f"s_mov_b32 {S_NSW}, 0")
    emit(f"s_mov_b32 {S_MINE}, 0")
    emit(f"s_add_u32 {S_T[0]}, {S_C}, 1")
    emit(f"s_cmp_ge_u32 {S_T[0]}, {S_NCHUNKS}")
    emit(f"s_cbranch_scc1 {label('bnostage')}")
    emit(f"s_bitcmp1_b32 {S_DEBUG}, 0")        # timing experiment: no copies of B
    emit(f"s_cbranch_scc1 {label('bnostage')}")
    stage_chunk(S_T[0])
    emit(f"s_add_u32 {S_SB}, {S_SB}, 1")
    emit(f"s_mov_b32 {S_MINE}, 1")
    emit(f"{label('bnostage')}:")
    emit(f"s_cmp_eq_u32 {S_C}, {S_NCHUNKS}")   # SCC = 0: not finished
    emit(f"{label('bret')}:")
    emit(f"s_setpc_b64 {S_RET}")

    if not GROUP:
        emit("; ---- drain: the three entries still in the pipeline ----")
        for ring in range(NSTRIP):
            emit(f"{label(f'drain_{ring}')}:")
            for j in range(1, NSTRIP):
                entry_fma_stage((ring + j) % NSTRIP, 2 * (NSTRIP - 1 - j))
            if ring < NSTRIP - 1:
                emit(f"s_branch {label('end')}")
    emit(f"{label('end')}:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")


def main():
    generate()
    text = "// GENERATED by gen_spmm_flat.py -- do not edit.\n" + "".join(
        f'"{line}\\n"\n' for line in out)
    if ARGS.out:
        with open(ARGS.out, "w") as f:
            f.write(text)
    else:
        print(text, end="")


if __name__ == "__main__":
    main()
