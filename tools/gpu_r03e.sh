#!/bin/bash
# Round-3 visit E: loss select (8+12+12 bits, cheaper prologues) and the top-k tie path: tests + per-kernel durations.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_decode_gpu.py tests/test_decode_layer_gpu.py tests/test_decode_fullsize_gpu.py tests/test_end_to_end_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_loss_decode.txt 2>&1
tail -n 8 $OUT/pytest_loss_decode.txt
cd /tmp
for v in r02:tools/libssdhip_r02.so new:ssd_keras_amd/libssdhip.so; do
  tag=${v%%:*}; lib=${v#*:}
  SSDHIP_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loss_$tag -o loss -- python $R/tools/time_loss.py > $OUT/loss_$tag.log 2>&1
  f=$(find $OUT/loss_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; grep case $OUT/loss_$tag.log
  python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r.get("Name") or r.get("KernelName") or ""
    if "ssdhip" in n:
        print("%-60s calls %6s avg_us %8.2f" % (n[:60], r.get("Calls"), float(r.get("AverageNs") or r.get("Average") or 0)/1e3))
P
  cp "$f" $OUT/loss_${tag}_kernel_stats.csv
done
cd $R
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -size +5M -delete
for v in r02:tools/libssdhip_r02.so new:ssd_keras_amd/libssdhip.so; do
  tag=${v%%:*}; lib=${v#*:}
  echo "== decode $tag" | tee -a $OUT/time_decode_topk.log
  SSDHIP_LIB=$R/$lib S512=0 timeout 300 python tools/time_decode.py 2>&1 | grep case | tee -a $OUT/time_decode_topk.log | cut -c1-260
done
