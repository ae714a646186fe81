#!/usr/bin/env python3
"""Developer / test helper: pull the gfx950 code objects out of the built library (.hip_fatbin -> offload bundles ->
code objects) and disassemble them with llvm-objdump.  Used by scripts/kernel_resources.py-style probes and by
tests/test_asm_hazards.py (a CPU-side check of the hand-counted wait states inside the opaque asm statements)."""
import os
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(REPO, "dex-nerf_amd", "lib", "libdexnerf_hip.so")


def extract_code_objects(lib, workdir):
    """Write every gfx950 code object of `lib` into `workdir`; returns their paths."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    offs = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for n, o in enumerate(offs):
        bf = os.path.join(workdir, f"b{n}")
        with open(bf, "wb") as f:
            f.write(blob[o:(offs[n + 1] if n + 1 < len(offs) else len(blob))])
        co = bf + ".co"
        subprocess.call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                         f"--input={bf}", f"--output={co}"], stderr=subprocess.DEVNULL)
        if os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def disassemble(co):
    """{demangled kernel name: [instruction text lines]} of one code object."""
    txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)
    funcs, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:\s*$", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is None:
            continue
        ins = line.split("//")[0].strip()
        if ins:
            funcs[cur].append(ins)
    names = list(funcs)
    if names:
        dem = subprocess.check_output(["c++filt"] + names, text=True).splitlines()
        funcs = {d: funcs[n] for n, d in zip(names, dem)}
    return funcs


def kernel_isa_sha16(want, lib=None):
    """sha256 (first 16 hex digits) of the instruction stream of the one kernel of `lib` whose demangled name contains `want`: the
    identity of a kernel BINARY - unchanged by comments, refactors and edits to other kernels, changed by anything that moves an
    instruction.  bench.py ties a committed PMC record to the library it runs with this.  None when the tools or the kernel are missing."""
    import hashlib
    import tempfile
    lib = lib or DEFAULT_LIB
    try:
        with tempfile.TemporaryDirectory() as d:
            for co in extract_code_objects(lib, d):
                table = subprocess.check_output([f"{LLVM}/llvm-objdump", "-t", co], text=True, stderr=subprocess.DEVNULL)
                syms = subprocess.run(["c++filt"], input=table, text=True, capture_output=True, check=True).stdout
                if want not in syms.replace("dn::", ""):
                    continue
                hits = [ins for name, ins in disassemble(co).items() if want in name.replace("dn::", "")]
                if len(hits) == 1:
                    return hashlib.sha256("\n".join(hits[0]).encode()).hexdigest()[:16]
    except (OSError, subprocess.CalledProcessError):
        return None
    return None


if __name__ == "__main__":
    import tempfile
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else DEFAULT_LIB
    pat = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    with tempfile.TemporaryDirectory() as d:
        for co in extract_code_objects(lib, d):
            for name, ins in disassemble(co).items():
                if pat in name:
                    print(f"==== {name} ({len(ins)} instructions)")
                    if pat:
                        print("\n".join(ins))


# ---- static checks on a disassembled kernel (linear order; conservative across branches) ---------------------------
_REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]")


def regs_of(operand_text):
    """Set of ('v'|'s', index) named in an operand string."""
    out = set()
    for m in _REG.finditer(operand_text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_ins(line):
    parts = line.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def lds_read_violations(ins):
    """Instructions that touch the destination registers of an LDS read while that read may still be in flight.

    Model: LDS operations return in order; `s_waitcnt lgkmcnt(N)` leaves at most the N youngest outstanding (a wait without
    an lgkmcnt field leaves them all).  A ds_read's destination is 'pending' from its issue until a wait retires it; any
    other instruction naming one of those registers in between would read stale data or have its result overwritten.
    Scalar memory loads also count in lgkmcnt and return out of order, so they are tracked as entries without registers
    (they can only make a counted wait retire FEWER of the LDS reads than assumed: a wait with SMEM in flight is treated as
    retiring nothing unless it is lgkmcnt(0)).
    The walk is in layout order: a conditional branch is walked through (its fall-through path), and after an UNCONDITIONAL one
    (s_branch, s_endpgm, s_setpc_b64) nothing carries over - the next instruction is only reached by jumps, not from the reads
    above it (the run-time-shape kernels place out-of-line blocks there)."""
    pending = []   # [(dest regs or None for SMEM, text, index)]
    bad = []
    for i, line in enumerate(ins):
        op, ops = split_ins(line)
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = []
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                if any(p[0] is None for p in pending) and n > 0:
                    continue
                while len(pending) > n:
                    pending.pop(0)
            continue
        is_read = op.startswith("ds_read") or op.startswith("ds_load")
        # a younger LDS read may name the same destination (returns are in order: the younger value wins, e.g. a FIFO slot
        # refilled without having been consumed); every other mention - source or destination - is a violation
        touched = regs_of(" ".join(ops[1:] if is_read else ops))
        for dest, text, j in pending:
            if dest and touched & dest:
                bad.append((j, text, i, line))
        if is_read:
            pending.append((regs_of(ops[0]), line, i))
        elif op.startswith("ds_"):
            pending.append((set(), line, i))          # LDS stores / atomics occupy a counter slot, no destination
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            pending.append((None, line, i))
    return bad


def _wait_states(line):
    op, ops = split_ins(line)
    if op == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def sgpr_base_vmem_violations(ins, need=5):
    """VALU writes an SGPR (v_readfirstlane_b32, v_readlane_b32) -> a VMEM instruction reads that SGPR as its
    base: the gfx9 / CDNA hazard table asks for 5 wait states in between, and the hazard recognizer cannot see into the
    opaque asm statements that hold our SGPR-base global_store_dwordx4 / global_load_lds_dwordx4.  Returns
    [(writer index, writer, vmem index, vmem, wait states seen)]."""
    bad = []
    for i, line in enumerate(ins):
        op, ops = split_ins(line)
        if not (op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_")):
            continue
        sregs = {r for o in ops for r in regs_of(o) if r[0] == "s"}
        if not sregs:
            continue
        states, j = 0, i - 1
        while j >= 0 and states < need:
            wop, wops = split_ins(ins[j])
            if wop in ("v_readfirstlane_b32", "v_readlane_b32") and regs_of(wops[0]) & sregs:   # (v_readlane: SGPR-spill restores)
                bad.append((j, ins[j], i, line, states))
                break
            if wop.startswith("s_cbranch") or wop == "s_barrier" or wop == "s_endpgm":
                break   # another path / a barrier: not a straight-line distance any more
            states += _wait_states(ins[j])
            j -= 1
    return bad


def mfma_result_read_violations(ins, need=7):
    """A vector instruction that is not an MFMA reads a register an MFMA wrote fewer than `need` wait states earlier.

    The explicit-schedule stages (mlp_stage48.h run_stage48x) issue their MFMAs from asm statements, so the compiler's hazard
    recognizer pads nothing behind them; the conversions / picks that read finished accumulators are placed by construction.
    gfx950: the result of a 4-pass MFMA (v_mfma_f32_16x16x32_*) is due NumPasses + 3 = 7 wait states (quad-cycles) after its
    issue.  Counted here in time, as the hardware does: an s_nop N is N + 1, an intervening MFMA 4 (the matrix pipe takes a 4-pass
    MFMA every 16 cycles), anything else 1.  Another MFMA may name the register at once (accumulator forwarding / its own rule)."""
    bad = []
    for i, line in enumerate(ins):
        op, ops = split_ins(line)
        if not op.startswith("v_mfma"):
            continue
        dst = regs_of(ops[0])
        states, j = 0, i + 1
        while j < len(ins) and states < need:
            nop_, nops = split_ins(ins[j])
            if nop_ in ("s_branch", "s_endpgm", "s_setpc_b64"):
                break
            if nop_.startswith("v_mfma"):
                states += 4
            elif nop_ == "s_nop":
                states += int(nops[0], 0) + 1
            else:
                vector = nop_.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_"))
                has_dst = nop_.startswith(("v_", "ds_read", "ds_load")) or ("_load_" in nop_ and "_lds_" not in nop_)
                if vector and regs_of(" ".join(nops[1:] if has_dst else nops)) & dst:   # (sources only: a write-after-write is another rule)
                    bad.append((i, line, j, ins[j], states))
                    break
                states += 1
            j += 1
    return bad


def m0_lds_dma_violations(ins):
    """SALU writes M0 -> the LDS-DMA (global_load_lds_*) that takes its LDS base from M0: 1 wait state."""
    bad = []
    for i, line in enumerate(ins):
        op, _ = split_ins(line)
        if not op.startswith("global_load_lds"):
            continue
        if i >= 1 and re.match(r"s_mov_b32\s+m0\b", ins[i - 1]):
            bad.append((i - 1, ins[i - 1], i, line))
    return bad


def wide_store_data_violations(ins):
    """A VMEM store of more than 8 bytes still reads its data VGPRs when the next instruction issues: a VALU write to
    them in the next wait state corrupts the stored dwords (seen in round 1: hipcc reused the piece registers at once)."""
    bad = []
    for i, line in enumerate(ins):
        op, ops = split_ins(line)
        if op not in ("global_store_dwordx4", "global_store_dwordx3", "buffer_store_dwordx4", "buffer_store_dwordx3"):
            continue
        data = regs_of(ops[1])
        if i + 1 < len(ins):
            nop, nops = split_ins(ins[i + 1])
            if nop.startswith("v_") and nops and regs_of(nops[0]) & data:
                bad.append((i, line, i + 1, ins[i + 1]))
    return bad
