#!/bin/bash
# Compile-only probe of the adjacency-stationary kernels: registers / scratch per kernel (seconds, no GPU needed).
# Usage: tools/probe_isa.sh [extra hipcc flags]   -> ISA in gpurun_out/probe/probe.s
set -e
cd "$(dirname "$0")/../megacrn_amd/csrc"
mkdir -p ../../gpurun_out/probe
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc --cuda-device-only -S "$@" -I. -o ../../gpurun_out/probe/probe.s ../../tools/probe_prop.hip
grep -E "^; (NumVgprs|ScratchSize|Occupancy)|^_ZN4mcrn.*:" ../../gpurun_out/probe/probe.s | paste - - - - | sed -E 's/: ; @[^\t]*//' | cut -c1-160
