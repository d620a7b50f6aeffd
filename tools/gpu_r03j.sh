#!/bin/bash
# Round-3 visit J: pool5 fast path, head CU split after the shorter chain, x3 bound diagnostics, full suite.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03q
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
grep -E "largest error|passed|failed|FAILED" $OUT/pytest_gpu.txt | tail -n 30
for rep in 1 2; do
  for wgs in 160 128 192 224; do
    SSDHIP_HEAD_WGS=$wgs timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_wgs${wgs}_$rep.json 2> $OUT/bench_err.log
    python - $OUT/bench_wgs${wgs}_$rep.json $wgs <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("head_wgs", sys.argv[2], d["value"], d["ms_per_step"])
P
  done
done
