#!/bin/bash
# Round-3 visit S2: the Cin = 64 kernels with the filters in registers (WREG) against the LDS-resident form, same box, alternating.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03za
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -k "c64 or conv1_block or conv64 or model or backbone" > $OUT/pytest_conv.txt 2>&1
tail -n 3 $OUT/pytest_conv.txt
for rep in 1 2; do
  for w in 0 1; do
    SSDHIP_C64_WREG=$w timeout 300 python tools/ablate_c64.py > $OUT/c64_wreg${w}_$rep.json 2>> $OUT/err.log
    echo "WREG=$w $(cat $OUT/c64_wreg${w}_$rep.json)"
  done
done
for rep in 1 2; do
  for w in 0 1; do
    SSDHIP_C64_WREG=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_wreg${w}_$rep.json 2>> $OUT/err.log
    python - $OUT/bench_wreg${w}_$rep.json $w <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("WREG", sys.argv[2], "step", d["ms_per_step"], "conv fwd", d["conv_roofline"]["forward_ms"])
P
  done
done
