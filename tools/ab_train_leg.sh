#!/bin/bash
# bench_extra.train_leg alone (tools/time_train_leg.py) under the round-6 switches, alternating on one box
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in "SSD_TRAIN_GRAPH=0 SSDHIP_NO_TAPS_BWD=1" "SSD_TRAIN_GRAPH=0" "SSD_TRAIN_GRAPH=1 SSD_TRAIN_GRAPH_OPT=0" "SSD_TRAIN_GRAPH=1"; do
  r=$(env $v timeout 400 python tools/time_train_leg.py 2>/dev/null | tail -1)
  echo "AB $v -> $r"
done
done
