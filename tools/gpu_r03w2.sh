#!/bin/bash
# Round-3 visit W2: kernel stats of the training step (eager), 15 steps.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03zh
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -o train -- python $R/tools/prof_train.py > $OUT/prof_train.log 2>&1
tail -n 2 $OUT/prof_train.log
cp $(find $OUT/trace_train -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv
find $OUT -name "*.csv" -size +5M -delete; find $OUT -name "*.db" -delete
