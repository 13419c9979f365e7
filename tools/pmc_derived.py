#!/usr/bin/env python3
"""Derived metrics per kernel from the counter passes of tools/pmc_profile.sh
(the gfx94x formulas; rocprofv3 on this image has no gfx950 derived-metric
tables).  Counters are sums over the eight XCDs, each with 32 CUs x 4 SIMDs.

  python tools/pmc_derived.py <pmc-dir>  >  profiles/<round>_derived_metrics.md

  VALU busy          = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE * 128)      share of SIMD cycles issuing vector ALU work
  lanes per VALU op  = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU            of 64 (execution mask)
  waves per SIMD     = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE * 128)           occupancy (SQ_WAVE_CYCLES counts quad-cycles)
  L2 hit rate        = TCC_HIT / (TCC_HIT + TCC_MISS)
  HBM read / written = FETCH_SIZE, WRITE_SIZE (KiB; see tools/pmc_traffic.py for the calibration of FETCH_SIZE)
Kernels run one at a time under counter collection, so a kernel that normally
shares the GPU with another launch is seen alone, on the wave slots it is given.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1

print("| kernel | launches | VALU busy | lanes per VALU op (of 64) | waves per SIMD | L2 hit rate | HBM read per launch (FETCH_SIZE) | HBM written per launch | vector instr. per launch | LDS bank-conflict share of LDS cycles |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0)):
    t = tot[k]
    if not k.startswith(("rtd::", "void rtd::", "rtb::")):
        continue
    n = calls[k].get("GRBM_GUI_ACTIVE", 0) or 1
    simd_cycles = t["GRBM_GUI_ACTIVE"] * 128.0
    busy = 4.0 * t["SQ_ACTIVE_INST_VALU"] / simd_cycles if simd_cycles else 0.0
    lanes = t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"] if t["SQ_ACTIVE_INST_VALU"] else 0.0
    occ = 4.0 * t["SQ_WAVE_CYCLES"] / simd_cycles if simd_cycles else 0.0
    hit = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]) if t["TCC_HIT_sum"] + t["TCC_MISS_sum"] else 0.0
    # share of the LDS array's busy cycles that conflicts added (both count LDS-array cycles; dividing by
    # SQ_ACTIVE_INST_LDS, which counts quad-cycles of instruction issue, gave figures above 100 %)
    conflicts = t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"] if t.get("SQ_LDS_IDX_ACTIVE") else 0.0
    print(f"| `{k}` | {n} | {100 * busy:.0f} % | {lanes:.1f} | {occ:.1f} | {100 * hit:.0f} % | "
          f"{t['FETCH_SIZE'] * 1024 / n / 1e6:.1f} MB | {t['WRITE_SIZE'] * 1024 / n / 1e6:.1f} MB | "
          f"{t['SQ_INSTS_VALU'] / n / 1e6:.1f} M | {100 * conflicts:.0f} % |")
