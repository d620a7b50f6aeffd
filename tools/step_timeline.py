"""Kernel timeline of the LAST complete bench step in a rocprofv3 kernel trace (`*_kernel_trace.csv`): every kernel between
one preprocess_kernel and the next, with start offsets and durations in microseconds.
Usage: python tools/step_timeline.py TRACE.csv OUT.json"""
import csv
import json
import sys


def main(trace, out):
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[2]]
    if len(starts) < 2:
        raise SystemExit("no complete step in the trace")
    a, b = starts[-2], starts[-1]
    t0 = rows[a][0]
    step = [{"start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "kernel": k.replace("void ", "")[:64]}
            for s, e, k in rows[a:b]]
    span = (rows[b][0] - t0) / 1e3
    busy = sum(k["dur_us"] for k in step)
    json.dump({"step_us": round(span, 1), "kernel_busy_us": round(busy, 1), "kernels": step}, open(out, "w"), indent=0)
    print("step %.1f us, kernels busy %.1f us, %d launches" % (span, busy, len(step)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
