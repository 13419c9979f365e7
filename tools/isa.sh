#!/bin/bash
# Device assembly of the render-loop kernels with the build's own flags, and the
# register / LDS use of every kernel:  bash tools/isa.sh [out.s]
OUT="${1:-/tmp/rt_context.s}"
cd "$(dirname "$0")/../raytracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize \
  -I../../include --cuda-device-only -S -o "$OUT" device/rt_context.hip 2>/dev/null
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|name):" "$OUT" | paste - - - - - | sed 's/  */ /g'
