#!/bin/bash
# Round-3 visit F: split-K extra layers, loss / top-k follow-ups, the whole GPU suite, a bench line.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03m
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 200 python tools/time_extras.py $OUT/time_extras.json 2>&1 | grep layer
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -n 6 $OUT/pytest_gpu.txt
timeout 200 python tools/time_loss.py $OUT/time_loss.json 2>&1 | grep case
S512=0 timeout 300 python tools/time_decode.py 2>&1 | grep random_init | cut -c1-260
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick.json 2> $OUT/bench_err.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03m/bench_quick.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["conv_roofline"]["forward_ms"], d["roofline"]["kernel_ms"], d["roofline"]["decode_ms_in_step"])
print({k:v for k,v in d["conv_roofline"]["kernel_per_layer"].items() if "k1" in k or "s2" in k or "p0" in k})
P
SSDHIP_LIB=tools/libssdhip_r02.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick_r02lib.json 2> $OUT/bench_err_r02.log
head -c 300 $OUT/bench_quick_r02lib.json; echo
