#!/usr/bin/env python3
"""Post-build ISA lint of libsputnik_hip.so for the kernels that count their
outstanding vector-memory operations by hand (`s_waitcnt vmcnt(N)` with the
loads issued from inline asm, which the compiler takes as complete where they
are issued: spmm_tiled.hip, spmm_tiled64.hip, sddmm_tiled.hip, attention.hip).

A scratch check (csrc/Makefile) sees spills; it does not see the compiler
adding a vector-memory operation inside a counted loop, nor its scheduler
giving the register of a load that is still in flight to something else (that
happened once: DESIGN.md section 3.1).  This script disassembles the gfx950
code objects of the library and checks, per kernel:

  1. no scratch_* / flat_* instruction anywhere;
  2. in the main loop (the innermost natural loop of the control-flow graph that
     holds the last s_barrier), every path of two consecutive iterations is
     walked (so that loads issued at the end of an iteration are followed into
     the next one) with a FIFO model of `vmcnt` (every vector-memory operation
     enters, `s_waitcnt vmcnt(N)` retires all but the N youngest): no
     instruction may read or write a VGPR that an un-retired load is going to
     write;
  3. spmm_tiled_kernel<BN, WAVES, RPW, BK>: the vector-memory operations on the
     main path of one iteration number exactly S + 2 * RPW (S = BK * (BN / 256)
     / WAVES LDS-DMA copies + one window = two loads per row), and the counted
     waits are exactly {2(D-1), 2(D-1) + S, 2 RPW} with D = 4 (eight columns per
     lane) or 8 -- the constants derived in the comment above
     spmm_tiled_body_dpp.

Exit status 1 on any violation.  Run by torch_sputnik_amd/build.py after the
link step and by tests/test_isa_lint.py (CPU: needs only llvm-objdump).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "torch_sputnik_amd", "lib", "libsputnik_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store",
        "buffer_atomic", "scratch_", "flat_")
INSTR = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):.*?(?:<[^>+]+\+0x([0-9a-fA-F]+)>)?\s*$")
FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def disassemble(lib):
    """-> {kernel symbol: [(offset, mnemonic, operands, branch target offset or None)]}"""
    tmp = tempfile.mkdtemp(prefix="isa_lint_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernels = {}
        for name in sorted(os.listdir(tmp)):
            if not name.endswith("gfx950"):
                continue
            text = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, name)], check=True,
                                  capture_output=True, text=True).stdout
            current, base = None, 0
            for line in text.splitlines():
                f = FUNC.match(line)
                if f:
                    base, current = int(f.group(1), 16), f.group(2)
                    kernels[current] = []
                    continue
                m = INSTR.match(line)
                if m and current is not None:
                    target = int(m.group(4), 16) if m.group(4) and m.group(1).startswith(
                        ("s_cbranch", "s_branch")) else None
                    kernels[current].append((int(m.group(3), 16) - base, m.group(1), m.group(2), target))
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def vregs(operands):
    regs = set()
    for m in VREG.finditer(operands):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def is_vmem(mnemonic):
    return mnemonic.startswith(VMEM)


def load_dest(mnemonic, operands):
    """VGPRs a vector-memory LOAD writes (empty for stores and LDS-DMA copies)."""
    if "_load_" not in mnemonic or "_lds_" in mnemonic:
        return set()
    first = operands.split(",")[0]
    return vregs(first)


class Cfg:
    """Basic blocks of one kernel, their successors, dominators and the natural
    loop around the kernel's last s_barrier."""

    def __init__(self, instrs):
        self.instrs = instrs
        by_off = {off: i for i, (off, *_rest) in enumerate(instrs)}
        leaders = {0}
        for i, (_off, mn, _ops, target) in enumerate(instrs):
            if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                if i + 1 < len(instrs):
                    leaders.add(i + 1)
                if target in by_off:
                    leaders.add(by_off[target])
        starts = sorted(leaders)
        self.blocks = [(a, (starts[j + 1] - 1) if j + 1 < len(starts) else len(instrs) - 1)
                       for j, a in enumerate(starts)]
        block_of = {}
        for b, (a, z) in enumerate(self.blocks):
            for i in range(a, z + 1):
                block_of[i] = b
        self.block_of = block_of
        self.succ = []
        for b, (a, z) in enumerate(self.blocks):
            _off, mn, _ops, target = instrs[z]
            out = []
            if not mn.startswith(("s_branch", "s_endpgm", "s_setpc")) and z + 1 < len(instrs):
                out.append(block_of[z + 1])
            if mn.startswith(("s_cbranch", "s_branch")) and target in by_off:
                out.append(block_of[by_off[target]])
            self.succ.append(out)
        self.pred = [[] for _ in self.blocks]
        for b, out in enumerate(self.succ):
            for t in out:
                self.pred[t].append(b)

    def dominators(self):
        n = len(self.blocks)
        full = set(range(n))
        dom = [full.copy() for _ in range(n)]
        dom[0] = {0}
        changed = True
        while changed:
            changed = False
            for b in range(1, n):
                preds = [dom[p] for p in self.pred[b]]
                new = (set.intersection(*preds) if preds else set()) | {b}
                if new != dom[b]:
                    dom[b], changed = new, True
        return dom

    def main_loop(self):
        """(header block, set of body blocks) of the innermost natural loop that
        holds the kernel's last s_barrier; None if there is none."""
        barriers = [i for i, ins in enumerate(self.instrs) if ins[1] == "s_barrier"]
        target_block = self.block_of[barriers[-1]] if barriers else None
        dom = self.dominators()
        best = None
        loops = {}
        for b, out in enumerate(self.succ):
            for h in out:
                if h in dom[b]:   # back edge b -> h
                    body = loops.setdefault(h, {h})
                    stack = [b]
                    while stack:
                        x = stack.pop()
                        if x not in body:
                            body.add(x)
                            stack.extend(self.pred[x])
        for h, body in loops.items():
            if target_block in body and (best is None or len(body) < len(best[1])):
                best = (h, body)
        if best is not None:
            return best
        # kernels whose only barrier publishes a stationary tile before the loop
        # (sddmm_tiled.hip): the largest loop that holds a counted wait
        def counted(body):
            for b in body:
                a, z = self.blocks[b]
                for _o, mn, ops, _t in self.instrs[a:z + 1]:
                    w = re.search(r"vmcnt\((\d+)\)", ops) if mn == "s_waitcnt" else None
                    if w and int(w.group(1)) > 0:
                        return True
            return False
        for h, body in loops.items():
            if counted(body) and (best is None or len(body) > len(best[1])):
                best = (h, body)
        return best


def lint_flat_kernel(instrs):
    """spmm_flat_kernel (generated loop, gen_spmm_flat.py): the VGPR index mode
    makes destination and src2 of EVERY vector ALU instruction M0-relative, and
    it shares M0 with the LDS-DMA copies.  Checked on the linear instruction
    stream (the loop calls its boundary through s_swappc_b64, which no CFG follows):
      * between s_set_gpr_idx_on and s_set_gpr_idx_off: only v_pk_fma_f32 into the
        accumulators v[64:127]; no branch, no branch target, every region closed;
      * an M0 write for an LDS-DMA copy is followed by that copy before the next
        s_set_gpr_idx_on (which overwrites M0), with an instruction in between;
      * a DPP move does not read a register that one of the two instructions in
        front of it wrote (VALU write -> DPP read: 2 wait states)."""
    errors = []
    targets = {t for _o, mn, _ops, t in instrs if t is not None}
    in_mode, m0_pending, m0_gap = False, False, 0
    recent = []   # (mnemonic, destination registers) of the last two instructions
    for off, mn, ops, _t in instrs:
        first = ops.split(",")[0] if ops else ""
        dest = vregs(first) if mn.startswith(("v_", "ds_read", "global_load", "buffer_load")) and \
            not mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) else set()
        if mn == "s_set_gpr_idx_on":
            if in_mode:
                errors.append(f"+0x{off:x}: s_set_gpr_idx_on inside an open index-mode region")
            if m0_pending:
                errors.append(f"+0x{off:x}: index mode switched on between an M0 write and its LDS-DMA copy")
            in_mode = True
        elif mn == "s_set_gpr_idx_off":
            in_mode = False
        elif in_mode:
            if off in targets:
                errors.append(f"+0x{off:x}: branch target inside an index-mode region")
            if mn.startswith("v_pk_fma_f32"):
                if not dest or min(dest) < 64 or max(dest) > 127:
                    errors.append(f"+0x{off:x} {mn} {ops}: index-mode FMA outside the accumulators")
            elif mn != "s_set_gpr_idx_idx":
                errors.append(f"+0x{off:x} {mn}: instruction inside an index-mode region")
        if "m0" in first and mn.startswith("s_"):
            m0_pending, m0_gap = True, 0
        elif mn == "global_load_lds_dwordx4":
            if m0_pending and m0_gap < 1:
                errors.append(f"+0x{off:x}: LDS-DMA copy directly behind its M0 write (1 wait state)")
            m0_pending = False
        elif m0_pending:
            m0_gap += 1
        if "_dpp" in mn:
            src = vregs(",".join(ops.split(",")[1:2]))
            for pmn, pdest in recent:
                if pmn.startswith("v_") and src & pdest:
                    errors.append(f"+0x{off:x} {mn} {ops}: DPP reads v{sorted(src & pdest)} written by "
                                  f"{pmn} within two instructions")
        recent = (recent + [(mn, dest)])[-2:]
    if in_mode:
        errors.append("index-mode region left open at the end of the kernel")
    if not any(mn == "s_set_gpr_idx_on" for _o, mn, _ops, _t in instrs):
        errors.append("no index-mode region found (wrong kernel?)")
    return errors


def lint_kernel(name, instrs):
    errors = []
    for off, mn, ops, _t in instrs:
        if mn.startswith(("scratch_", "flat_")):
            errors.append(f"{mn} at +0x{off:x}: scratch / flat access in a hand-counted kernel")
    if "spmm_flat_kernel" in name:
        return errors + lint_flat_kernel(instrs)
    if "spmm_panel" in name:
        # its one hand-written wait is the vmcnt(0) behind the panel copy, which
        # drains everything; all other loads are the compiler's, with its waits
        return errors
    if "sddmm_flat_kernel" in name:
        # its hand-written waits are vmcnt(0) drains of the copying wave, which issues
        # nothing but copies; the compute waves issue no vector-memory load at all (their
        # descriptors come from LDS) and nothing waits for their stores
        for off, mn, ops, _t in instrs:
            if mn == "s_waitcnt" and re.search(r"vmcnt\(([1-9]\d*)\)", ops):
                errors.append(f"+0x{off:x}: counted vmcnt wait in sddmm_flat_kernel (only drains expected)")
        return errors
    if "sddmm_stationary_kernel" in name or "sddmm_quad_kernel" in name:
        # their hand-written waits are full drains (wait_vm<0> where a row block's first
        # requests are awaited); every counted wait is the compiler's own.  Since round 5 a
        # workgroup walks several row blocks in a loop: nothing hand-counted is carried
        # around it (the drain at the top of a block also retires the block before's
        # stores), and the FIFO model below would only explore the compiler's schedule.
        return errors
    cfg = Cfg(instrs)
    loop = cfg.main_loop()
    if loop is None:
        if "spmm_tiled" in name:
            errors.append("no loop with a workgroup barrier or a counted wait found")
        return errors   # (a fully unrolled kernel: nothing is carried around a loop)
    header, body = loop

    def block_instrs(b):
        a, z = cfg.blocks[b]
        return instrs[a:z + 1]

    # side paths: blocks of the loop that issue loads and drain everything themselves
    side = set()
    for b in body:
        ins = block_instrs(b)
        if (any(mn == "s_waitcnt" and re.search(r"vmcnt\(0\)", ops) for _o, mn, ops, _t in ins) and
                any(is_vmem(mn) for _o, mn, _ops, _t in ins)):
            side.add(b)

    # 2. FIFO model of vmcnt along every path of two iterations (states are
    #    memoised per block: all main paths issue the same operations)
    seen = set()
    reported = set()
    lengths_at_header = set()
    stack = [(header, (), 0)]
    while stack:
        b, state, lap = stack.pop()
        key = (b, state, lap)
        if key in seen:
            continue
        seen.add(key)
        if len(seen) > 50000:
            errors.append("path explosion in the vmcnt model (kernel structure changed?)")
            break
        outstanding = list(state)
        for off, mn, ops, _t in block_instrs(b):
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ops)
                if m:
                    keep = int(m.group(1))
                    outstanding = outstanding[len(outstanding) - keep:] if keep else []
                continue
            touched = vregs(ops)
            own = set(load_dest(mn, ops)) if is_vmem(mn) else set()
            for j, dest in outstanding:
                # (a load into a register that an older load also writes is not the
                # hazard looked for: returns are in order, the younger one wins)
                hit = (touched & set(dest)) - own
                if hit and (off, j) not in reported:
                    reported.add((off, j))
                    errors.append(f"+0x{off:x} {mn} {ops}: touches v{sorted(hit)} while the load at "
                                  f"+0x{j:x} that writes it is still in flight")
            if is_vmem(mn):
                outstanding.append((off, tuple(sorted(load_dest(mn, ops)))))
        if len(outstanding) > 63:
            errors.append("more than 63 vector-memory operations outstanding: vmcnt is a 6-bit counter")
        for t in cfg.succ[b]:
            if t not in body:
                continue
            next_lap = lap + 1 if t == header else lap
            if t == header:
                lengths_at_header.add(len(outstanding))
            if next_lap < 2:
                stack.append((t, tuple(outstanding), next_lap))

    # 3. the counted constants of spmm_tiled_kernel
    m = re.search(r"spmm_tiled_kernelINS\d+_\d*\w*TileConfigILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEELb([01])E", name)
    if m:
        bn, waves, rpw, bk = (int(m.group(g)) for g in range(1, 5))
        stage = bk * (bn // 256) // waves
        depth = 4 if bn // 64 == 8 else 8
        # vector-memory operations issued on the way from the header to every block
        # (side blocks avoided): a fixpoint over sets of counts; the inner loops of
        # the body issue none, so the sets converge
        vm_in_block = {b: sum(1 for _o, mn, _ops, _t in block_instrs(b) if is_vmem(mn)) for b in body}
        reach = {b: set() for b in body}
        reach[header] = {0}
        work = [header]
        counts = set()
        while work:
            b = work.pop()
            if b in side:
                continue
            out = {c + vm_in_block[b] for c in reach[b]}
            if len(out) > 64:
                errors.append("the number of vector-memory operations per iteration is unbounded")
                break
            for t in cfg.succ[b]:
                if t == header:
                    counts |= out
                elif t in body and not out <= reach[t]:
                    reach[t] |= out
                    work.append(t)
        if counts != {stage + 2 * rpw}:
            errors.append(f"vector-memory operations per iteration on the main paths: {sorted(counts)}, "
                          f"the waits assume S + 2*RPW = {stage} + {2 * rpw}")
        waits = set()
        for b in body - side:
            for _o, mn, ops, _t in block_instrs(b):
                if mn == "s_waitcnt":
                    w = re.search(r"vmcnt\((\d+)\)", ops)
                    if w:
                        waits.add(int(w.group(1)))
        expected = {2 * (depth - 1) + stage, 2 * rpw}
        if rpw > depth:
            expected.add(2 * (depth - 1))
        if waits != expected:
            errors.append(f"counted waits {sorted(waits)} differ from the derived {sorted(expected)}")
    return errors


def lint(lib=LIB, verbose=True):
    kernels = disassemble(lib)
    counted = {k: v for k, v in kernels.items()
               if any(mn == "global_load_lds_dwordx4" for _o, mn, _ops, _t in v)}
    failures = 0
    for name, instrs in sorted(counted.items()):
        errors = lint_kernel(name, instrs)
        short = re.sub(r"^_ZN11sputnik_hip\d*_?\w*?(\d+)", "", name)[:100]
        if errors:
            failures += 1
            print(f"[isa_lint] FAIL {name}")
            for e in errors[:20]:
                print("           ", e)
        elif verbose:
            print(f"[isa_lint] ok   {len(instrs):6d} instructions  {short}")
    if not counted:
        print("[isa_lint] FAIL: no kernel with LDS-DMA copies found (wrong library?)")
        return 1
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(lint(sys.argv[1] if len(sys.argv) > 1 else LIB))
