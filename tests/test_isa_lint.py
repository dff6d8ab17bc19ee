"""CPU: the post-build ISA lint of the hand-counted kernels (tools/isa_lint.py)
passes on the built library, and the checker itself catches the hazards it is
there for (synthetic instruction streams)."""
import importlib.util
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lint():
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(REPO, "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_built_library_passes_the_lint():
    assert _lint().lint(verbose=False) == 0


def _loop(body):
    """header; body...; s_barrier; back edge -- offsets 4 bytes apart."""
    instrs = [("s_mov_b32", "s0, 0", None)] + body + [("s_barrier", "", None),
                                                      ("s_cbranch_scc1", "65000", 4)]
    return [(4 * i, mn, ops, t) for i, (mn, ops, t) in enumerate(instrs)]


def test_lint_catches_a_register_reused_under_an_inflight_load():
    lint = _lint()
    ok = _loop([("global_load_dword", "v5, v1, s[2:3]", None),
                ("s_waitcnt", "vmcnt(0)", None),
                ("v_add_u32_e32", "v6, v5, v5", None)])
    assert lint.lint_kernel("k", ok) == []
    # the compiler placed an address computation in v5 ahead of the wait
    bad = _loop([("global_load_dword", "v5, v1, s[2:3]", None),
                 ("v_lshl_add_u64", "v[5:6], v[8:9], 2, s[4:5]", None),
                 ("s_waitcnt", "vmcnt(0)", None)])
    assert any("still in flight" in e for e in lint.lint_kernel("k", bad))
    # a use that the counted wait does not cover: one load too many stays in flight
    late = _loop([("global_load_dword", "v5, v1, s[2:3]", None),
                  ("global_load_dword", "v7, v1, s[2:3]", None),
                  ("s_waitcnt", "vmcnt(2)", None),
                  ("v_mov_b32_e32", "v9, v5", None),
                  ("s_waitcnt", "vmcnt(0)", None)])
    assert any("still in flight" in e for e in lint.lint_kernel("k", late))


def test_lint_catches_scratch_and_miscounted_iterations():
    lint = _lint()
    spill = _loop([("scratch_store_dword", "off, v3, s0", None)])
    assert any("scratch" in e for e in lint.lint_kernel("k", spill))
    # spmm_tiled_kernel<512, 16, 8, 32>: S + 2*RPW = 4 + 16 operations per iteration
    name = ("_ZN11sputnik_hip12_GLOBAL__N_117spmm_tiled_kernelINS0_10TileConfigILi512ELi16ELi8ELi32"
            "EEELb1EEEvi")
    body = [("global_load_lds_dwordx4", "v80, s[4:5]", None)] * 4
    body += [("global_load_dword", f"v{10 + i}, v1, s[2:3]", None) for i in range(16)]
    body += [("s_waitcnt", "vmcnt(10)", None), ("s_waitcnt", "vmcnt(6)", None),
             ("s_waitcnt", "vmcnt(16)", None)]
    assert lint.lint_kernel(name, _loop(body)) == []
    extra = body + [("global_load_dword", "v40, v1, s[2:3]", None)]
    assert any("per iteration" in e for e in lint.lint_kernel(name, _loop(extra)))


def test_lint_checks_the_index_mode_regions_of_the_flat_kernel():
    lint = _lint()
    name = "_ZN11sputnik_hip12_GLOBAL__N_116spmm_flat_kernelILi0EEEvi"

    def stream(body):
        return [(4 * i, mn, ops, t) for i, (mn, ops, t) in enumerate(body)]

    fma = ("v_pk_fma_f32", "v[64:65], v[16:17], v[32:33], v[64:65] op_sel:[1,0,0]", None)
    good = stream([("s_set_gpr_idx_on", "s66, gpr_idx(SRC2,DST)", None), fma, fma,
                   ("s_set_gpr_idx_off", "", None), ("s_endpgm", "", None)])
    assert lint.lint_kernel(name, good) == []
    # any other vector instruction inside a region would have its destination shifted by M0
    bad = stream([("s_set_gpr_idx_on", "s66, gpr_idx(SRC2,DST)", None), fma,
                  ("v_add_u32_e32", "v16, v16, v0", None),
                  ("s_set_gpr_idx_off", "", None), ("s_endpgm", "", None)])
    assert any("inside an index-mode region" in e for e in lint.lint_kernel(name, bad))
    # the index mode overwrites M0: not between an M0 write and its LDS-DMA copy
    m0 = stream([("s_add_u32", "m0, s72, 0x4000", None),
                 ("s_set_gpr_idx_on", "s66, gpr_idx(SRC2,DST)", None), fma,
                 ("s_set_gpr_idx_off", "", None),
                 ("global_load_lds_dwordx4", "v1, s[74:75]", None), ("s_endpgm", "", None)])
    assert any("between an M0 write" in e for e in lint.lint_kernel(name, m0))
    # VALU write -> DPP read needs two wait states
    dpp = stream([("v_mov_b32_e32", "v5, v7", None),
                  ("v_mov_b32_dpp", "v16, v5 row_newbcast:0 row_mask:0xf bank_mask:0xf", None),
                  ("s_set_gpr_idx_on", "s66, gpr_idx(SRC2,DST)", None), fma,
                  ("s_set_gpr_idx_off", "", None), ("s_endpgm", "", None)])
    assert any("DPP reads" in e for e in lint.lint_kernel(name, dpp))
