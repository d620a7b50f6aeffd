#!/bin/bash
# Round-3 visit S3: two loader waves in the Cin = 64 kernel (shipped) against one (tools/libssdhip_prof_nl1.so), same box, alternating.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03zb
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -k "c64 or conv1_block" > $OUT/pytest_conv.txt 2>&1
tail -n 3 $OUT/pytest_conv.txt
for rep in 1 2 3; do
  for l in prof_nl1 prof; do
    SSDHIP_LIB=$R/tools/libssdhip_$l.so timeout 300 python tools/ablate_c64.py > $OUT/c64_${l}_$rep.json 2>> $OUT/err.log
    echo "$(cat $OUT/c64_${l}_$rep.json)"
  done
done
for rep in 1 2; do
  for l in prof_nl1 prof; do
    SSDHIP_LIB=$R/tools/libssdhip_$l.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_${l}_$rep.json 2>> $OUT/err.log
    python - $OUT/bench_${l}_$rep.json $l <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "step", d["ms_per_step"], "conv fwd", d["conv_roofline"]["forward_ms"])
P
  done
done
