#!/bin/bash
# Round-3 visit Q: early conv4_3 head on the side stream (mode 4) A/B; kernel breakdown of the reference-precision forward.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03w
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_end_to_end_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest.txt 2>&1
tail -n 3 $OUT/pytest.txt
for rep in 1 2; do
  for cfgs in 3:56 4:56 4:40 4:72; do
    mode=${cfgs%%:*}; w0=${cfgs#*:}
    SSDHIP_HEAD_OVERLAP=$mode SSDHIP_HEAD0_WGS=$w0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_m${mode}_w${w0}_$rep.json 2> $OUT/bench_err.log
    python - $OUT/bench_m${mode}_w${w0}_$rep.json $cfgs <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("mode:head0_wgs", sys.argv[2], d["value"], d["ms_per_step"], d["config"]["launch"][:20])
except Exception as e:
    print("failed", sys.argv[2], e)
P
  done
done
tail -n 3 $OUT/bench_err.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x3 -o x3 -- python $R/tools/prof_x3.py > $OUT/prof_x3.log 2>&1
cd $R
cat $OUT/prof_x3.log | grep "x3 forward"
cp $(find $OUT/trace_x3 -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
python - <<'P'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03w/x3_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("   %-74s %5s avg %9.1f us  share %5.1f%%" % (r["Name"][:74], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
P
