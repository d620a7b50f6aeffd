#!/bin/bash
# the headline step (bench.py quick line) with the chain kernel's filter ring at 8 (rounds 4-6) / 16 / 32 fragments, alternating on one box
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in "SSDHIP_CHAIN_RING=8" "SSDHIP_CHAIN_RING=16" "SSDHIP_CHAIN_RING=32"; do
  r=$(env $v timeout 400 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('forward_ms'))")
  echo "AB $v -> $r"
done
done
