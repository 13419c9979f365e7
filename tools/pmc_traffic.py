#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_profile.sh into the
per-launch memory-side traffic of the traversal kernel, as one JSON file that
bench.py reports under roofline.traffic.

  python tools/pmc_traffic.py <pmc-dir> <calibration-dir> <out.json> [bench args recorded in the file]

Units and corrections.  rocprofv3 reports both counters in KiB.  On gfx950
FETCH_SIZE is (read requests leaving L2) x 64 B although a request may carry
128 B; tools/fetch_calibrate.hip measures this on known byte counts:
a 16-B-per-lane stream and random 128-B records read at exactly 1/2 of their
bytes, random 64-B records at 1/1.  The kernel mixes both sizes, so the true
figure lies between FETCH_SIZE and 2 x FETCH_SIZE; `bytes_per_launch` takes the
upper end (FETCH_SIZE doubled), `bytes_per_launch_raw` the lower.  WRITE_SIZE
is exact for 16-B-per-lane streaming stores and counts 32 B for an isolated
8-B store.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


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


def main():
    pmc_dir, cal_dir, out_path = sys.argv[1:4]
    tot, calls = sums(pmc_dir)
    kernel = max((k for k in tot if "k_trace<2, false" in k), key=lambda k: tot[k]["FETCH_SIZE"])
    n = calls[kernel]["FETCH_SIZE"]
    fetch = tot[kernel]["FETCH_SIZE"] * 1024.0 / n
    write = tot[kernel]["WRITE_SIZE"] * 1024.0 / calls[kernel]["WRITE_SIZE"]
    shade = next(k for k in tot if "k_shade" in k)
    ns = calls[shade]["FETCH_SIZE"]

    cal, _ = sums(cal_dir)
    known = {"k_stream16": 2 << 30, "void k_gather<4>": 1 << 30, "void k_gather<8>": 2 << 30}
    known_w = {"k_store16": 2 << 30, "k_scatter8": 1 << 27}
    calibration = {k: {"bytes_read": v, "FETCH_SIZE_bytes": cal[k]["FETCH_SIZE"] * 1024.0,
                       "ratio": round(cal[k]["FETCH_SIZE"] * 1024.0 / v, 4)} for k, v in known.items() if k in cal}
    calibration.update({k: {"bytes_written": v, "WRITE_SIZE_bytes": cal[k]["WRITE_SIZE"] * 1024.0,
                            "ratio": round(cal[k]["WRITE_SIZE"] * 1024.0 / v, 4)}
                        for k, v in known_w.items() if k in cal})
    out = {
        "kernel": kernel,
        "bench_args": sys.argv[4:],
        "launches": n,
        "fetch_size_bytes_per_launch": round(fetch),
        "write_size_bytes_per_launch": round(write),
        "bytes_per_launch_raw": round(fetch + write),
        "bytes_per_launch": round(2.0 * fetch + write),
        "k_shade": {"launches": ns,
                    "fetch_size_bytes_per_launch": round(tot[shade]["FETCH_SIZE"] * 1024.0 / ns),
                    "write_size_bytes_per_launch": round(tot[shade]["WRITE_SIZE"] * 1024.0 / calls[shade]["WRITE_SIZE"])},
        "calibration": calibration,
    }
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
