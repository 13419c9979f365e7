#!/bin/bash
# Interleaved A/B of two builds of the libraries on the headline workload:
#   bash tools/ab_bench.sh <dir of build A> [bench args...]   (B = raytracer_amd/lib)
# Prints Mrays/s, ms per step, traversal launch ms for A, B, A, B.
A="$1"; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  for lib in "$A" ""; do
    RT_MI355_LIB_DIR="$lib" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${lib:-B}'[-1], d['value'], d['ms_per_step'], r['avg_launch_ms'])" || exit 1
  done
done
