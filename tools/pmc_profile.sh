#!/bin/bash
# Hardware-counter passes over a short bench run, one rocprofv3 invocation per
# pass (counters of one block must fit that block's slots).  Usage, on the GPU
# box:  bash tools/pmc_profile.sh <out-dir> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="$1"; shift
ARGS=("$@")
mkdir -p "$OUT"
pass() {
  name="$1"; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py "${ARGS[@]}" --no-cpu-baseline > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -3 "$OUT/$name.err"; return 1; }
  echo "pass $name ok"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU &&
pass sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM &&
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES &&
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum &&
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
pass write WRITE_SIZE &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
