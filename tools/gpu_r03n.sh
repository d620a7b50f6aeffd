#!/bin/bash
# Round-3 visit N: fused rowmax + match encoder: tests, timing, kernel stats (encoder + loss legs).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03t
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_encode_gpu.py tests/test_boxes_gpu.py tests/test_end_to_end_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest.txt 2>&1
tail -n 4 $OUT/pytest.txt
for lib in tools/libssdhip_r02.so ssd_keras_amd/libssdhip.so; do
  echo "== $lib"; SSDHIP_LIB=$R/$lib timeout 200 python tools/time_encoder.py 2>&1 | grep case
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_enc -o enc -- python $R/tools/time_encoder.py > $OUT/trace_enc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_loss -o loss -- python $R/tools/time_loss.py > $OUT/trace_loss.log 2>&1
cd $R
cp $(find $OUT/trace_enc -name "*kernel_stats.csv" | head -1) $OUT/encoder_kernel_stats.csv
cp $(find $OUT/trace_loss -name "*kernel_stats.csv" | head -1) $OUT/loss_kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
python - <<'P'
import csv
for f in ("gpurun_out/r03t/encoder_kernel_stats.csv","gpurun_out/r03t/loss_kernel_stats.csv"):
    print(f)
    for r in csv.DictReader(open(f)):
        n=r.get("Name","")
        if "ssdhip" in n: print("   %-70s %6s %9.2f" % (n[:70], r.get("Calls"), float(r.get("AverageNs",0))/1e3))
P
