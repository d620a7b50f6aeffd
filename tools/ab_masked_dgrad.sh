#!/bin/bash
# bench_extra.train_leg alone (tools/time_train_leg.py) with and without the ReLU links (the masked data gradients), alternating on one box
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in "SSDHIP_NO_MASKED_DGRAD=1" "SSDHIP_NO_MASKED_SUMS=1" "SSDHIP_NO_MASKED_SUMS=0"; do
  r=$(env $v timeout 400 python tools/time_train_leg.py 2>/dev/null | tail -1)
  echo "AB $v -> $r"
done
done
