#!/bin/bash
# SSD512 forward + decode (tools/prof_ssd512_forward.py) with the heads' slab path on / off, alternating on one box
cd "$(dirname "$0")/.."
for b in 16 8; do
for rep in 1 2 3; do
for v in "SSDHIP_NO_HALO_MIXED=1" "SSDHIP_NO_HALO_MIXED=0"; do
  r=$(env $v timeout 400 python tools/prof_ssd512_forward.py $b 2>/dev/null | grep "ssd512 batch")
  echo "AB $v -> $r"
done
done
done
