#!/usr/bin/env python3
"""Fold rocprofv3 counter-collection CSVs (one --pmc pass each) into per-kernel HBM traffic for the ssdhip kernels.

    python tools/pmc_summary.py OUT.json PASS_DIR [PASS_DIR ...]

Units / corrections (MI355X_MICROARCH.md, HBM + rocprofv3 sections): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read (128-B requests tallied
at 64 B), so the read side is doubled: hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.  WRITE_SIZE is
uncalibrated (taken as is).  Values are per launch (mean over the launches seen)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"ssdhip::(\w+)", name)
    if not m:
        return None
    s = m.group(1)
    t = re.search(r"ssdhip::\w+<([^>]*)>", name)
    if t:
        args = [a.strip() for a in t.group(1).split(",")]
        # K4 is two launches per decode since round 5 (the 512-thread kernel + the redo launch of the FULL 256-thread one): keep them apart
        s += "<%s>" % (", ".join(args) if s == "nms_kernel" else args[0])
    return s


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row.get("Kernel_Name", ""))
                if k is None:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, ctr in acc.items():
        mean = {c: sum(v) / len(v) for c, v in ctr.items()}
        e = {"launches": max(len(v) for v in ctr.values()), "counters_mean": mean}
        if "FETCH_SIZE" in mean or "WRITE_SIZE" in mean:
            rd = 2.0 * mean.get("FETCH_SIZE", 0.0) * 1024.0
            wr = mean.get("WRITE_SIZE", 0.0) * 1024.0
            e.update({"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                      "note": "read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB (uncalibrated)"})
        res[k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
