"""Kernel timelines of bench steps in a rocprofv3 kernel trace (`*_kernel_trace.csv`): every kernel between one preprocess kernel and
the next, with start offsets, durations and the hardware queue in microseconds.

bench.py issues, in this order: eager warm-up steps, graph warm-ups, the TIMED steps (graph replays unless --graph 0), ten eager steps for
`decode_ms_in_step`, the per-kernel decode timing (the first `scan_kernel` of the trace), then the same step on TAMED heads (three
warm-ups, the timed replays, ten eager steps).  A step decodes from the head outputs (`scan_heads_kernel`); the per-kernel timing calls
and `raw_predictions` do not and are skipped.  Three timelines are written (the old tool took "the last step that contains
scan_heads_kernel", which since the tamed-heads leg is a tamed step -- its NMS row was not the headline's):

  headline_timed   the last TIMED step of the headline workload (a graph replay: two queues overlap)
  headline_eager   the last eager step of the headline workload (one queue: per-kernel durations without overlap)
  tamed_eager      the last eager step on tamed heads

Usage: python tools/step_timeline.py TRACE.csv OUT.json [TIMED_STEPS=10]"""
import csv
import json
import sys


def timeline(rows, a, b):
    t0 = rows[a][0]
    step = [{"start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "queue": q, "kernel": k.replace("void ", "")[:72]}
            for s, e, k, q in rows[a:b]]
    span = (max(e for _, e, _, _ in rows[a:b]) - t0) / 1e3             # first kernel's start to the last kernel's end
    busy = sum(k["dur_us"] for k in step)
    return {"step_us": round(span, 1), "kernel_busy_us": round(busy, 1), "launches": len(step),
            "queues": sorted(set(k["queue"] for k in step)), "kernels": step}


def main(trace, out, timed_steps=10):
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Queue_Id", 0) or 0)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "ssdhip::preprocess" in r[2]]          # preprocess_kernel / preprocess3_kernel (round 5)
    if len(starts) < 2:
        raise SystemExit("no complete step in the trace")
    steps = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if any("scan_heads_kernel" in r[2] for r in rows[a:b])
             and not any("scan_kernel(" in r[2] for r in rows[a:b])]
    first_scan = next((i for i, r in enumerate(rows) if "scan_kernel(" in r[2]), len(rows))
    head = [p for p in steps if p[1] <= first_scan]                    # the headline workload's steps
    tamed = [p for p in steps if p[0] > first_scan]
    res = {}
    if head:
        res["headline_eager"] = timeline(rows, *head[-1])             # the ten eager steps behind the timed region come last
        if len(head) > 10:
            res["headline_timed"] = timeline(rows, *head[-11])         # ... so the last timed step is the eleventh from the end
    if tamed:
        res["tamed_eager"] = timeline(rows, *tamed[-1])
    # compatibility with the readers of rounds 2-4: the top-level keys are the headline's eager step
    top = res.get("headline_eager") or timeline(rows, *steps[-1])
    out_obj = dict(top)
    out_obj.update(res)
    json.dump(out_obj, open(out, "w"), indent=0)
    for name, t in res.items():
        print("%-15s step %.1f us, kernels busy %.1f us, %d launches, queues %s" % (name, t["step_us"], t["kernel_busy_us"], t["launches"], t["queues"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10)
