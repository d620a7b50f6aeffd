"""One training step out of a rocprofv3 kernel trace: the kernels between the last two launches of MARKER (default: the encoder's
first kernel), grouped by kernel name with call counts and total microseconds, plus the step's span.
Usage: python tools/train_timeline.py TRACE.csv OUT.json [MARKER]"""
import csv
import json
import sys
from collections import OrderedDict


def main(trace, out, marker="iou_kernel"):
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) < 2:
        raise SystemExit("no complete step in the trace")
    a, b = starts[-2], starts[-1]
    span = (rows[b][0] - rows[a][0]) / 1e3
    agg = OrderedDict()
    for s, e, k in rows[a:b]:
        name = k.replace("void ", "")[:110]
        c = agg.setdefault(name, [0, 0.0])
        c[0] += 1
        c[1] += (e - s) / 1e3
    busy = sum(v[1] for v in agg.values())
    table = sorted(({"kernel": k, "calls": v[0], "total_us": round(v[1], 1), "share": round(v[1] / busy, 4)} for k, v in agg.items()),
                   key=lambda r: -r["total_us"])
    json.dump({"step_us": round(span, 1), "kernel_busy_us": round(busy, 1), "launches": b - a, "kernels": table}, open(out, "w"), indent=0)
    print("step %.1f us, kernels busy %.1f us, %d launches" % (span, busy, b - a))
    for r in table[:25]:
        print("%9.1f us %5d x  %5.1f %%  %s" % (r["total_us"], r["calls"], 100 * r["share"], r["kernel"][:100]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
