#!/bin/bash
# Round-3 visit B: the slab kernel with the bias as one early dword load + buffer stores; tolerant post-epilogue waits (mode 1152) A/B.
set -u
OUT=gpurun_out/r03i
mkdir -p $OUT
export TMPDIR=/tmp
ABLATE_MODES=128,1152,384,640 timeout 400 python tools/ablate_convh2.py $OUT/ablate_convh_bias_stores.json > $OUT/ablate_convh2.log 2>&1
cat $OUT/ablate_convh2.log
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_layers_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_conv.txt 2>&1
tail -5 $OUT/pytest_conv.txt
SSDHIP_CONVH_MODE=1152 timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_conv_1152.txt 2>&1
tail -3 $OUT/pytest_conv_1152.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick.json 2> $OUT/bench_err.log
head -c 600 $OUT/bench_quick.json; echo
SSDHIP_CONVH_MODE=1152 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick_1152.json 2> $OUT/bench_err_1152.log
head -c 600 $OUT/bench_quick_1152.json; echo
