#!/bin/bash
# Round-3 visit D: per-kernel durations of the loss (old library vs new, tile 128 vs 256), slab layers old vs new within one visit.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in r02:tools/libssdhip_r02.so:256 new:ssd_keras_amd/libssdhip.so:256 new128:tools/libssdhip_prof.so:128; do
  tag=${v%%:*}; rest=${v#*:}; lib=${rest%%:*}; ta=${rest#*:}
  SSDHIP_LIB=$R/$lib SSDHIP_LOSS_TA=$ta timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loss_$tag -o loss -- python $R/tools/time_loss.py > $OUT/loss_$tag.log 2>&1
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
# slab layers: old library vs new (modes 128 / 1152) in one visit
SSDHIP_LIB=tools/libssdhip_r02.so ABLATE_MODES=128 timeout 200 python tools/ablate_convh2.py 2>&1 | grep layer | cut -c1-160 | sed 's/^/r02 /'
SSDHIP_LIB=ssd_keras_amd/libssdhip.so ABLATE_MODES=128,1152 timeout 200 python tools/ablate_convh2.py $OUT/ablate_convh_new.json 2>&1 | grep layer | cut -c1-220 | sed 's/^/new /'
