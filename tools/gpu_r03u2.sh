#!/bin/bash
# Round-3 visit U2: kernel stats + timeline of one reference-precision forward (models/precise.py).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03ze
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x3 -o x3 -- python $R/tools/prof_x3.py > $OUT/prof_x3.log 2>&1
grep "x3 forward" $OUT/prof_x3.log
cp $(find $OUT/trace_x3 -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv
python - $(find $OUT/trace_x3 -name "*kernel_trace.csv" | head -1) $OUT/x3_timeline.json <<'P'
import csv, json, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "conv1_1_x3_kernel" in r[2]]
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
out = [{"start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "kernel": k.replace("void ", "")[:70]} for s, e, k in rows[a:b]]
json.dump({"span_us": round((rows[b][0] - t0) / 1e3, 1), "busy_us": round(sum(k["dur_us"] for k in out), 1), "kernels": out}, open(sys.argv[2], "w"), indent=0)
print("span", (rows[b][0] - t0) / 1e3, "busy", sum(k["dur_us"] for k in out), len(out))
P
find $OUT -name "*.csv" -size +5M -delete; find $OUT -name "*.db" -delete
