#!/usr/bin/env python3
"""Compact view of a kernel's ISA: the order of memory ops, waits, branches, barriers and MFMAs (runs compressed).
Usage: hipcc ... --cuda-device-only -S -o engine.s engine.hip ; python tools/isa_summary.py engine.s <mangled-substring> [max]"""
import sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 200
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
KEEP = ("global_load", "global_store", "global_atomic", "s_waitcnt", "s_cbranch", "s_and_saveexec", "s_or_b64 exec", "v_mfma", "ds_write", "ds_read",
        "ds_store", "ds_load", "s_barrier", "buffer_", "scratch_", "s_memrealtime", "s_branch")
out = []
for l in lines[start:end]:
    t = l.strip()
    if t.startswith(KEEP):
        op = t.split(" ")[0]
        out.append(t.split(";")[0].strip() if op == "s_waitcnt" else op)
comp, prev, cnt = [], None, 0
for t in out + [None]:
    if t == prev:
        cnt += 1
    else:
        if prev:
            comp.append(f"{prev} x{cnt}" if cnt > 1 else prev)
        prev, cnt = t, 1
print(f"{lines[start].split(':')[0]}: {end - start} lines")
print("\n".join(comp[:mx]))
