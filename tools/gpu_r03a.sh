#!/bin/bash
# Round-3 visit A: where do the per-tile costs of the slab kernel and the Cin = 64 kernel go?
set -u
OUT=gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/ablate_convh2.py $OUT/ablate_convh_epilogue.json > $OUT/ablate_convh2.log 2>&1
cat $OUT/ablate_convh2.log
for v in "" _c64a1 _c64a2 _c64a4 _c64a8; do
  SSDHIP_LIB=tools/libssdhip_prof$v.so timeout 120 python tools/ablate_c64.py >> $OUT/ablate_c64.jsonl 2>> $OUT/ablate_c64.err
done
cat $OUT/ablate_c64.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick.json 2> $OUT/bench_err.log
head -c 1500 $OUT/bench_quick.json
