#!/bin/bash
# Round-3 visit V2: reference-precision path after (a) the first-layer kernel with LDS-staged inputs, (b) heads and extra layers on three
# streams, (c) conv2_1 on the slab kernel: tests, timeline, the leg.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03zg
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_precise_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_precise.txt 2>&1
tail -n 3 $OUT/pytest_precise.txt
python - <<'P'
import json, torch, bench_extra as bx
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
a = bx.fp32_forward_leg(dev, 32)
b = bx.fp32x3_forward_leg(dev, 32, a)
print(json.dumps({"fp32": {k: v for k, v in a.items() if k != "note"}, "fp32x3": {k: v for k, v in b.items() if k not in ("note", "dtype")}}))
json.dump({"conv_roofline_fp32": a, "conv_roofline_fp32x3": b}, open("gpurun_out/r03zg/fp32x3_leg.json", "w"), indent=1)
P
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x3 -o x3 -- python $R/tools/prof_x3.py > $OUT/prof_x3.log 2>&1
grep "x3 forward" $OUT/prof_x3.log
cp $(find $OUT/trace_x3 -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv
python - $(find $OUT/trace_x3 -name "*kernel_trace.csv" | head -1) $OUT/x3_timeline.json <<'P'
import csv, json, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
starts = [i for i, r in enumerate(rows) if "conv1_1_x3_kernel" in r[2]]
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
out = [{"start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1), "queue": q, "kernel": k.replace("void ", "")[:70]} for s, e, k, q in rows[a:b]]
json.dump({"span_us": round((rows[b][0] - t0) / 1e3, 1), "busy_us": round(sum(k["dur_us"] for k in out), 1), "kernels": out}, open(sys.argv[2], "w"), indent=0)
print("span", (rows[b][0] - t0) / 1e3, "sum of durations", sum(k["dur_us"] for k in out), len(out))
P
find $OUT -name "*.csv" -size +5M -delete; find $OUT -name "*.db" -delete
