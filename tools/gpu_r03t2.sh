#!/bin/bash
# Round-3 visit T2: re-measure two A/Bs DESIGN.md quotes from visits whose logs were not kept:
# the two-stream head schedule (SSDHIP_HEAD_OVERLAP 0 vs 3) and the producers' issue priority (SSDHIP_C64_PRIO 0 vs 1).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03zd
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for rep in 1 2 3; do
  for m in 0 3; do
    SSDHIP_HEAD_OVERLAP=$m timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_overlap${m}_$rep.json 2>> $OUT/err.log
    python - $OUT/bench_overlap${m}_$rep.json $m <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("HEAD_OVERLAP", sys.argv[2], "step_ms", d["ms_per_step"])
P
  done
done
for rep in 1 2 3; do
  for pr in 0 1; do
    SSDHIP_C64_PRIO=$pr timeout 300 python tools/ablate_c64.py > $OUT/c64_prio${pr}_$rep.json 2>> $OUT/err.log
    echo "C64_PRIO=$pr $(cat $OUT/c64_prio${pr}_$rep.json)"
  done
done
