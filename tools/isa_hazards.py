#!/usr/bin/env python3
"""Scan the shipped gfx950 code objects for a hazard the compiler cannot see: a VMEM instruction issued from INLINE ASM
whose SGPR address pair was written by a VALU instruction (v_readfirstlane_b32 / v_readlane_b32 / a VOP3 with an SGPR
destination) fewer than 5 wait states earlier.  The hardware does not interlock that case (CDNA ISA guide, "manually
inserted wait states": VALU writes SGPR -> VMEM reads that SGPR: 5), LLVM's hazard recogniser inserts the s_nops for the
instructions it emits itself but not for the text of an asm statement, and the load then uses the STALE register pair.

    python tools/isa_hazards.py [megacrn_amd/libmegacrn_hip.so]      exit code 1 when a hazard is found

Found in round 4 with tools/kbench/prop1_test (prop_small.h's streamed adjacency fragments: one k-step of wrong fragments,
3e-2 .. 6e-2 relative error, in exactly the kernel variants whose ISA had such a pair); tests/test_host_cpu.py runs this scan
on every build.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJCOPY, OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objcopy", "/opt/rocm/lib/llvm/bin/llvm-objdump"
NEED = 5      # wait states between the VALU write of an SGPR and a VMEM instruction that reads it
VMEM = re.compile(r"^(global_(?:load|store|atomic)\w*|buffer_(?:load|store|atomic)\w*|scratch_(?:load|store)\w*)\s+(.*)$")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


CARRY_OUT = ("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")      # VOP3 forms whose SECOND operand is an SGPR destination (carry / flag)


def valu_sgpr_writes(ins):
    """SGPRs written by a VALU instruction: ANY s-register among its destination operands - the first operand always
    (v_readfirstlane / v_readlane / v_cmp* with an SGPR-pair destination / v_writelane is a VGPR write and has none), the second
    operand too for the carry-out and flag forms (*_co_*, v_mad_u64_u32, v_mad_i64_i32, v_div_scale)."""
    m = re.match(r"^(v_\w+)\s+(.*)$", ins)
    if not m:
        return set()
    op, args = m.group(1), m.group(2).split(",")
    out = sregs(args[0]) if args else set()
    if len(args) > 1 and ("_co_" in op or op.startswith(CARRY_OUT)):
        out |= sregs(args[1])
    if op.startswith("v_cmpx"):
        return set()                      # writes EXEC only
    return out


def _wait_states(ins):
    m = re.match(r"^s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def _vmem_saddr(ins):
    """SGPRs of the scalar address operand of a VMEM instruction (empty for the "off" / VGPR-address forms)."""
    vm = VMEM.match(ins)
    if not vm:
        return None
    addr = set()
    for p in [q.strip() for q in vm.group(2).split(",")][1:]:
        p0 = p.split()[0] if p else ""
        if re.fullmatch(r"s\[\d+:\d+\]", p0):
            addr |= sregs(p0)
    return addr


def _check(ins, addr, history):
    """history: (instruction, wait states) most recent LAST.  -> (writer, distance) of a hazard or None"""
    dist, watch = 0, set(addr)
    for prev, ws in reversed(history):
        if valu_sgpr_writes(prev) & watch:
            return (prev, dist) if dist < NEED else None
        # a SALU instruction that (re)writes a watched register produced the value the load reads: SALU -> VMEM needs no wait
        # states, and whatever wrote the register before it no longer matters
        ms = re.match(r"^s_\w+\s+([^,]+)", prev)
        if ms:
            watch -= sregs(ms.group(1))
            if not watch:
                return None
        dist += ws
        if dist >= NEED:
            return None
    return None


BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\d+)")


def scan_disassembly(dis):
    """-> list of (function, vmem instruction, writer instruction, wait states between them).
    Two passes per function: the linear instruction order, and every BRANCH EDGE - the instructions before a branch followed by
    the instructions at its target (loop back-edges: the streamed-fragment issue blocks sit at the head of unrolled loops)."""
    hits = []
    funcs, cur = [], None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            m2 = re.match(r"^\s+(\S.*?)\s*//", line)
            if m2 and cur is not None:
                cur[1].append((None, m2.group(1).strip()))
            continue
        if cur is None:
            cur = ("?", [])
            funcs.append(cur)
        cur[1].append((int(m.group(2), 16), m.group(1).strip()))
    for func, body in funcs:
        index = {a: i for i, (a, _) in enumerate(body) if a is not None}
        # pass 1: linear order
        for i, (_, ins) in enumerate(body):
            addr = _vmem_saddr(ins)
            if addr:
                h = _check(ins, addr, [(p, _wait_states(p)) for _, p in body[max(0, i - 12):i]])
                if h:
                    hits.append((func, ins, h[0], h[1]))
        # pass 2: branch edges (taken): history = what precedes the branch (+ the branch), continuation = the target's first instructions
        for i, (a, ins) in enumerate(body):
            mb = BRANCH.match(ins)
            if not mb or a is None:
                continue
            imm = int(mb.group(2))
            if imm >= 0x8000:
                imm -= 0x10000
            t = index.get(a + 4 + 4 * imm)
            if t is None:
                continue
            hist = [(p, _wait_states(p)) for _, p in body[max(0, i - 12):i + 1]]
            dist = 0
            for j in range(t, min(t + NEED, len(body))):
                tin = body[j][1]
                addr = _vmem_saddr(tin)
                if addr:
                    h = _check(tin, addr, hist)
                    if h and (func, tin, h[0], h[1]) not in hits:
                        hits.append((func, tin, h[0], h[1]))
                hist = hist + [(tin, _wait_states(tin))]
                dist += _wait_states(tin)
                if dist >= NEED:
                    break
    return hits


def code_objects(so, tmp):
    fat = os.path.join(tmp, "fatbin.bin")
    subprocess.run([OBJCOPY, "--dump-section", f".hip_fatbin={fat}", so, os.path.join(tmp, "copy.so")], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    for n, i in enumerate(starts):
        end = starts[n + 1] if n + 1 < len(starts) else len(data)
        img = os.path.join(tmp, f"co{n}.elf")
        open(img, "wb").write(data[i:end])
        yield img


def scan_library(so):
    tmp = tempfile.mkdtemp()
    try:
        hits, nvmem = [], 0
        for img in code_objects(so, tmp):
            dis = subprocess.run([OBJDUMP, "-d", img], capture_output=True, text=True, check=True).stdout
            nvmem += len(re.findall(r"\bglobal_load", dis))
            hits += scan_disassembly(dis)
        return hits, nvmem
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "megacrn_amd", "libmegacrn_hip.so")
    hits, nvmem = scan_library(so)
    for f, ins, prev, dist in hits:
        print(f"HAZARD {f}: `{prev}` -> {dist} wait state(s) -> `{ins}`")
    print(f"{so}: {nvmem} global loads scanned, {len(hits)} VALU-SGPR -> VMEM hazards")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
