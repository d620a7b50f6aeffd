"""Kernel timeline of the last complete TIMED bench step in a rocprofv3 kernel trace (`*_kernel_trace.csv`): every kernel between
one preprocess_kernel and the next, with start offsets and durations in microseconds.  A timed step decodes from the head outputs
(scan_heads_kernel); the bench's later raw_predictions / per-kernel timing calls also start with a preprocess_kernel and are skipped.
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
    pairs = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if any("scan_heads_kernel" in r[2] for r in rows[a:b])
             and not any("scan_kernel" in r[2] for r in rows[a:b])]
    a, b = pairs[-1] if pairs else (starts[-2], starts[-1])
    t0 = rows[a][0]
    step = [{"start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "kernel": k.replace("void ", "")[:64]}
            for s, e, k in rows[a:b]]
    span = (max(e for _, e, _ in rows[a:b]) - t0) / 1e3                # first kernel's start to the last kernel's end
    busy = sum(k["dur_us"] for k in step)
    json.dump({"step_us": round(span, 1), "kernel_busy_us": round(busy, 1), "kernels": step}, open(out, "w"), indent=0)
    print("step %.1f us, kernels busy %.1f us, %d launches" % (span, busy, len(step)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
