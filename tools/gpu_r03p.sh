#!/bin/bash
# Round-3 visit P: the reference-precision path on the slab kernel: tests + the fp32x3 leg.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03x
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_precise_gpu.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.txt 2>&1
grep -E "float16 x 3|passed|failed|FAILED" $OUT/pytest.txt | tail -n 12 | cut -c1-220
python - <<'P'
import json, torch, bench_extra as bx
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
a = bx.fp32_forward_leg(dev, 32)
b = bx.fp32x3_forward_leg(dev, 32, a)
print(json.dumps({"conv_roofline_fp32": {k: v for k, v in a.items() if k != "note"}, "conv_roofline_fp32x3": {k: v for k, v in b.items() if k not in ("note", "dtype")}}))
json.dump({"conv_roofline_fp32": a, "conv_roofline_fp32x3": b}, open("gpurun_out/r03x/fp32x3_leg.json", "w"), indent=1)
P
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x3 -o x3 -- python $R/tools/prof_x3.py > $OUT/prof_x3.log 2>&1
cd $R
grep "x3 forward" $OUT/prof_x3.log
cp $(find $OUT/trace_x3 -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
python - <<'P'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03x/x3_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("   %-74s %5s avg %9.1f us  share %5.1f%%" % (r["Name"][:74], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
P
