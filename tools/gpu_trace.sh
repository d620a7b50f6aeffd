#!/bin/bash
# Kernel-trace timelines of the bench step under several launch configurations.  Usage (through gpurun): bash tools/gpu_trace.sh TAG
set -u
TAG=${1:-r02p}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "1 1" "1 2" "0 0" "1 0"; do
  set -- $cfg
  SSDHIP_HEAD_OVERLAP=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_g$1_o$2 -o b -- \
      python $R/bench.py --graph $1 --steps 10 --warmup 5 --no-cpu-baseline --no-extra > $OUT/trace_g$1_o$2.json 2> $OUT/trace_g$1_o$2.err
  f=$(find $OUT/trace_g$1_o$2 -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_timeline.py $f $OUT/timeline_g$1_o$2.json
done
find $OUT -name "*.csv" -size +5M -delete
find $OUT -name "*.db" -delete
