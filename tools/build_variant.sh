#!/bin/bash
# Builds the device library of a source tree into ab/<name>/ for A/B timing
# (RT_MI355_LIB_DIR=ab/<name>):  bash tools/build_variant.sh <name> [git rev | "." ] [extra hipcc flags...]
# "." (default) = the working tree; a git revision is exported to a scratch directory first.
set -e
NAME="$1"; REV="${2:-.}"; shift; shift || true
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/ab/$NAME"; mkdir -p "$OUT"
SRC="$ROOT"
if [ "$REV" != "." ]; then
  SRC="$(mktemp -d)"
  git -C "$ROOT" archive "$REV" raytracer_amd/csrc include | tar -x -C "$SRC"
fi
cd "$SRC/raytracer_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize \
  -I"$SRC/include" "$@" -shared -o "$OUT/librt_mi355.so" device/rt_context.hip device/bvh_build.hip device/rt_group.hip
cp "$ROOT/raytracer_amd/lib/librt_host.so" "$OUT/"
[ "$REV" != "." ] && rm -rf "$SRC"
echo "built $OUT"
