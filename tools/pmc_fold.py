#!/usr/bin/env python3
"""Mean counter values per (kernel, grid) from rocprofv3 counter_collection CSVs: python tools/pmc_fold.py DIR [substr]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
sub = sys.argv[2] if len(sys.argv) > 2 else "ssdhip"
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    print(key)
    for c, v in sorted(acc[key].items()):
        print("    %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
