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


def valu_sgpr_writes(ins):
    """SGPRs written by a VALU instruction (first operand(s) of v_readfirstlane / v_readlane / compares and carry-outs
    that name an SGPR destination)."""
    m = re.match(r"^(v_\w+)\s+(.*)$", ins)
    if not m:
        return set()
    op, args = m.group(1), m.group(2).split(",")
    if op.startswith(("v_readfirstlane", "v_readlane")):
        return sregs(args[0])
    if op.startswith("v_cmp") or op.startswith("v_div_scale") or "_co_" in op:
        # VOP3 forms: an SGPR pair among the leading destination operands
        dst = args[0] if op.startswith("v_cmp") else (args[1] if len(args) > 1 else "")
        return sregs(dst)
    return set()


def scan_disassembly(dis):
    """-> list of (function, vmem instruction, writer instruction, wait states between them)"""
    hits, func, window = [], "?", []      # window: (instruction text, wait states it provides), most recent last
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            func, window = m.group(1), []
            continue
        m = re.match(r"^\s+(\S.*?)\s*//", line)
        if not m:
            continue
        ins = m.group(1).strip()
        vm = VMEM.match(ins)
        if vm:
            # the SGPR address is the trailing s[a:b] operand (saddr form); "off" forms carry no SGPR
            ops = vm.group(2)
            tail = [p.strip() for p in ops.split(",")]
            addr = set()
            for p in tail[1:]:
                p0 = p.split()[0] if p else ""
                if re.fullmatch(r"s\[\d+:\d+\]", p0):
                    addr |= sregs(p0)
            if addr:
                dist, watch = 0, set(addr)
                for prev, ws in reversed(window):
                    w = valu_sgpr_writes(prev)
                    if w & watch:
                        if dist < NEED:
                            hits.append((func, ins, prev, dist))
                        break
                    # a SALU instruction that (re)writes a watched register produced the value the load reads: SALU -> VMEM
                    # needs no wait states, and whatever wrote the register before it no longer matters
                    ms = re.match(r"^s_\w+\s+([^,]+)", prev)
                    if ms:
                        watch -= sregs(ms.group(1))
                        if not watch:
                            break
                    dist += ws
                    if dist >= NEED:
                        break
        ws = 1
        m2 = re.match(r"^s_nop\s+(\d+)", ins)
        if m2:
            ws = int(m2.group(1)) + 1
        window.append((ins, ws))
        if len(window) > 12:
            window.pop(0)
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
