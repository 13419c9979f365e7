#!/usr/bin/env python3
"""Turn the counter passes of tools/pmc_profile.sh into the record bench.py reads
(profiles/r03_counters.json): per workload, the traversal and shading kernels'
hardware counters per launch and per ray, stamped with a hash of the kernel sources
they were measured on, so that a record of other kernels is never quoted.

  python tools/pmc_record.py <pmc-dir> <out.json>

<pmc-dir> holds, per pass, rocprofv3's counter CSVs and the JSON line bench.py
printed in that pass (<pass>.json).  Every pass ran the same bench command, so the
rays per step are the same in all of them.

Units and corrections.  Counters are summed over the eight XCDs.  FETCH_SIZE and
WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a read request as 64 B
although it may carry 128 B (tools/fetch_calibrate.hip: 16-B-per-lane streams and
random 128-B records read at exactly half their bytes, random 64-B records at
1/1), so the true figure lies between FETCH_SIZE and twice it: `hbm_bytes` takes
the upper end (2 x FETCH_SIZE + WRITE_SIZE), `hbm_bytes_raw` the lower.
SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (x4 for cycles);
GRBM_GUI_ACTIVE is summed over the XCDs (32 CUs x 4 SIMDs each).
Counter passes run kernels one at a time: a traversal launch is seen alone on the
wave slots its chain gives it (half a compute unit), not beside the other chain's.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sums(root):
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0]
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k][row["Counter_Name"]] += 1
    return tot, calls


def kernel_record(t, c, rays):
    """Per-launch and per-ray figures of one kernel; `rays` = rays its launches processed in one pass."""
    n = max(c.values()) if c else 0
    if not n:
        return None
    simd_cycles = t["GRBM_GUI_ACTIVE"] * 128.0
    out = {"launches_per_pass": n, "rays_per_pass": rays}
    if t.get("SQ_INSTS_VALU"):
        out["valu_insts_per_launch"] = t["SQ_INSTS_VALU"] / n
        out["valu_insts_per_ray"] = t["SQ_INSTS_VALU"] / rays
    if t.get("SQ_INSTS_SALU") and t.get("SQ_INSTS_VALU"):
        out["salu_per_valu"] = round(t["SQ_INSTS_SALU"] / t["SQ_INSTS_VALU"], 4)
    if t.get("SQ_ACTIVE_INST_VALU"):
        out["lanes_per_valu_inst"] = round(t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"], 2)
        out["valu_busy"] = round(4.0 * t["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4) if simd_cycles else None
        out["valu_cycles_per_inst"] = round(4.0 * t["SQ_ACTIVE_INST_VALU"] / t["SQ_INSTS_VALU"], 3)
    if t.get("SQ_WAVE_CYCLES") and simd_cycles:
        out["waves_per_simd"] = round(4.0 * t["SQ_WAVE_CYCLES"] / simd_cycles, 2)
    # where a wave's time goes (quad-cycle counters; shares of the waves' resident time)
    if t.get("SQ_WAVE_CYCLES"):
        wc = t["SQ_WAVE_CYCLES"]
        for name, key in (("SQ_WAIT_ANY", "wait_any_share"), ("SQ_WAIT_INST_ANY", "wait_inst_any_share"),
                          ("SQ_WAIT_INST_LDS", "wait_inst_lds_share"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_share"),
                          ("SQ_ACTIVE_INST_VALU", "active_inst_valu_share"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds_share"),
                          ("SQ_ACTIVE_INST_VMEM", "active_inst_vmem_share"), ("SQ_ACTIVE_INST_SCA", "active_inst_scalar_share")):
            if t.get(name) is not None:
                out[key] = round(t[name] / wc, 4)
    if t.get("TCP_PENDING_STALL_CYCLES_sum") is not None and t.get("GRBM_GUI_ACTIVE"):
        out["tcp_pending_stall_share_of_cu_cycles"] = round(t["TCP_PENDING_STALL_CYCLES_sum"] / (t["GRBM_GUI_ACTIVE"] * 32.0), 4)
    if t.get("SQ_LDS_IDX_ACTIVE"):
        # SQ_LDS_BANK_CONFLICT = LDS-array cycles added by conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array cycles
        out["lds_bank_conflict_share"] = round(t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"], 4)
    if t.get("TCC_HIT_sum") is not None and t["TCC_HIT_sum"] + t.get("TCC_MISS_sum", 0) > 0:
        out["l2_hit_rate"] = round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        fetch, write = t["FETCH_SIZE"] * 1024.0, t["WRITE_SIZE"] * 1024.0
        out["fetch_size_bytes_per_launch"] = fetch / n
        out["write_size_bytes_per_launch"] = write / n
        out["hbm_bytes_per_launch"] = (2.0 * fetch + write) / n
        out["hbm_bytes_per_launch_raw"] = (fetch + write) / n
        out["hbm_bytes_per_ray"] = (2.0 * fetch + write) / rays
    return out


def main():
    from bench import kernel_source_hash

    pmc_dir, out_path = sys.argv[1:3]
    tot, calls = sums(pmc_dir)
    lines = []
    for path in sorted(glob.glob(os.path.join(pmc_dir, "*.json"))):
        with open(path) as f:
            text = f.read().strip()
        if text.startswith("{"):
            lines.append(json.loads(text.splitlines()[-1]))
    if not lines:
        raise SystemExit("no bench line found beside the counter passes")
    line = lines[0]
    # rays traced by the uncounted kernels in one pass: warm-up + timed steps + the repeat step, each a full frame
    rays_per_step = line["value"] * 1e6 * line["ms_per_step"] / 1e3
    steps = line["steps"] + line["warmup"] + (1 if line.get("image_repeats") is not None else 0)
    rays = rays_per_step * steps
    fast = "fast slab" in line["config"].get("arithmetic", "")
    traces = [k for k in tot if "k_trace<2, false" in k]
    trace = max(traces, key=lambda k: tot[k].get("SQ_INSTS_VALU", 0)) if traces else None
    shade = next((k for k in tot if "k_shade" in k), None)
    # the one-launch bounce loop, where the run went through it (bench.py --opt FUSED_LOOP=1, or the library's own choice)
    paths = [k for k in tot if "k_path<" in k]
    path = max(paths, key=lambda k: tot[k].get("SQ_INSTS_VALU", 0)) if paths else None
    args = line["bench_args"]
    key = f"{args['workload']}:{args['width']}x{args['height']}:b{args['bounces']}:d{args['detail']}:fast{int(fast)}"
    if args.get("fast_build"):
        key += f":leaf{args['fast_build']}"
    try:
        with open(out_path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        rec = {}
    h = kernel_source_hash()
    if rec.get("source_hash") != h:
        rec = {"source_hash": h, "workloads": {}}
    entry = rec["workloads"].get(key, {})
    entry.update({"bench_args": args, "rays_per_step": rays_per_step, "steps_per_pass": steps})
    if path and (not trace or tot[path].get("SQ_INSTS_VALU", 0) > tot[trace].get("SQ_INSTS_VALU", 0)):
        entry.update({"path_kernel": path, "path": kernel_record(tot[path], calls[path], rays)})
    else:
        entry.update({"trace_kernel": trace, "trace": kernel_record(tot[trace], calls[trace], rays),
                      "shade": kernel_record(tot[shade], calls[shade], rays)})
    rec["workloads"][key] = entry
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(json.dumps(rec["workloads"][key], indent=1)[:3000])


if __name__ == "__main__":
    main()
