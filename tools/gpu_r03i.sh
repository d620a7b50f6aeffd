#!/bin/bash
# Round-3 visit G: the reference-precision path (tests), split-K extras inside the graphed step (A/B + timelines).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03p
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_precise_gpu.py tests/test_conv_gpu.py tests/test_decode_layer_gpu.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest_precise.txt 2>&1
grep -E "x3 max|float16 x 3|passed|failed|Error|error" $OUT/pytest_precise.txt | head -40
for rep in 1 2; do
  for pref in none splitk; do
    SSDHIP_PREFER=$pref timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_${pref}_$rep.json 2> $OUT/bench_err.log
    python - $OUT/bench_${pref}_$rep.json $pref <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"])
P
  done
done
cd /tmp
for pref in none splitk; do
  SSDHIP_PREFER=$pref timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$pref -o bench -- \
      python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra > $OUT/trace_$pref.json 2> $OUT/trace_err_$pref.log
  f=$(find $OUT/trace_$pref -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_timeline.py $f $OUT/step_timeline_$pref.json > /dev/null 2>&1
  cp $(find $OUT/trace_$pref -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$pref.csv 2>/dev/null
done
cd $R
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
python - <<'P'
import json
for pref in ("none","splitk"):
    d=json.load(open("gpurun_out/r03p/step_timeline_%s.json"%pref))
    print(pref, "step_us", d["step_us"], "busy", d["kernel_busy_us"])
    for k in d["kernels"]:
        print("   %8.1f %7.1f %s" % (k["start_us"], k["dur_us"], k["kernel"][:70]))
P
python - <<'P'
import json, torch, bench_extra as bx
dev = torch.device("cuda:0")
a = bx.fp32_forward_leg(dev, 32)
b = bx.fp32x3_forward_leg(dev, 32, a)
print(json.dumps({"conv_roofline_fp32": a, "conv_roofline_fp32x3": b}))
json.dump({"conv_roofline_fp32": a, "conv_roofline_fp32x3": b}, open("gpurun_out/r03p/fp32x3_leg.json", "w"), indent=1)
P
