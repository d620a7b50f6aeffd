#!/bin/bash
# bench_extra.train_leg alone (tools/time_train_leg.py) with conv2_2 -> pool2 / conv3_3 -> pool3 as one pool-keep launch each, or as before
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in "SSDHIP_NO_HALO_POOL_KEEP=1" "SSDHIP_NO_HALO_POOL_KEEP=0"; do
  r=$(env $v timeout 400 python tools/time_train_leg.py 2>/dev/null | tail -1)
  echo "AB $v -> $r"
done
done
