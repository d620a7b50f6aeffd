#!/bin/bash
# Round-3 visit C: slab epilogue diet + tolerant waits; the new loss select; nms launch-bounds sweep.
set -u
OUT=gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
ABLATE_MODES=128,1152,384 timeout 300 python tools/ablate_convh2.py $OUT/ablate_convh.json > $OUT/ablate_convh2.log 2>&1
cat $OUT/ablate_convh2.log
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_layers_gpu.py tests/test_loss_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_conv_loss.txt 2>&1
tail -n 8 $OUT/pytest_conv_loss.txt
timeout 200 python tools/time_loss.py $OUT/time_loss.json 2>&1 | tee $OUT/time_loss.log
for ta in 64 128 256; do
  SSDHIP_LIB=tools/libssdhip_prof.so SSDHIP_LOSS_TA=$ta timeout 200 python tools/time_loss.py 2>&1 | grep case | tee -a $OUT/time_loss_ta_sweep.log
done
for v in "" _nms4 _nms5; do
  echo "lib prof$v" | tee -a $OUT/time_decode_nms_waves.log
  SSDHIP_LIB=tools/libssdhip_prof$v.so S512=0 timeout 300 python tools/time_decode.py 2>&1 | grep case | tee -a $OUT/time_decode_nms_waves.log
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick.json 2> $OUT/bench_err.log
head -c 400 $OUT/bench_quick.json; echo
SSDHIP_CONVH_MODE=1152 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick_1152.json 2> $OUT/bench_err_1152.log
head -c 400 $OUT/bench_quick_1152.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick_b.json 2> $OUT/bench_err.log
head -c 400 $OUT/bench_quick_b.json; echo
