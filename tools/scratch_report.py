#!/usr/bin/env python3
"""Per-kernel private-segment (scratch = register spill) size of every kernel in the shipped library, read from the code objects' metadata.
Round 5 lost a tenth of the bf16-mode step to an edit that pushed the 256x256 ping-pong GEMM tiles past 256 VGPRs (556 B/lane of
scratch): nothing failed, the step only got slower (profiles/r5/experiments.md section 10).  tests/test_host_cpu.py holds the guard.
usage: tools/scratch_report.py [library.so]      -> lines "bytes  kernel" for every kernel that spills"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
EM_AMDGPU = 224


def code_objects(so):
    """the gfx950 ELF images embedded in the host library (one per translation unit)"""
    data = open(so, "rb").read()
    pos = 0
    while True:
        i = data.find(b"\x7fELF\x02\x01\x01", pos)
        if i < 0:
            return
        pos = i + 4
        if struct.unpack_from("<H", data, i + 18)[0] != EM_AMDGPU:
            continue
        shoff, = struct.unpack_from("<Q", data, i + 40)
        shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
        yield data[i:i + shoff + shentsize * shnum]


def scratch_sizes(so):
    """{demangled kernel name: private_segment_fixed_size in bytes per lane}"""
    out = {}
    for img in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(img)
        try:
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            n = re.search(r"\.name:\s+(\S+)", blk)
            s = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if n and s:
                out[n.group(1)] = int(s.group(1))
    if not out:
        return out
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return {re.sub(r"^void ", "", d).replace("mcrn::", "").split("(")[0]: out[m] for m, d in zip(names, dem)}


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "megacrn_amd", "libmegacrn_hip.so")
    sizes = scratch_sizes(so)
    print(f"{len(sizes)} kernels, {sum(1 for v in sizes.values() if v)} with scratch")
    for k, v in sorted(sizes.items(), key=lambda kv: -kv[1]):
        if v:
            print(f"{v:6d}  {k}")
