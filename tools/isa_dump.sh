#!/bin/bash
# tools/isa_dump.sh [extra -D flags]: gfx950 assembly of wfpt_kernels.hip into build/wfpt_kernels.s (device side only), with the same flags the
# library is built with. tools/isa_count.py then tallies one kernel's instructions per class and per loop.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  --offload-device-only -S -Iinclude -Iwavefront_path_tracer_amd/csrc "$@" -o build/wfpt_kernels.s wavefront_path_tracer_amd/csrc/wfpt_kernels.hip
grep -c . build/wfpt_kernels.s
