#!/usr/bin/env python3
"""Summarise the counter CSVs written by tools/pmc_profile.sh: per kernel, the
sum of every counter over its dispatches and the dispatch count."""
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
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print(f"  {c:36s} {tot[k][c]:18.0f}  over {calls[k][c]} dispatches")
