#!/bin/bash
# bench.py's headline step (graph, head schedule 4) with conv6_2's head in the small launch (SPLIT=2, rounds 4-5) or in the capped
# launch beside the chain (SPLIT=3, round 6), alternating on one box: ms_per_step, decode ms in step
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in "SSDHIP_HEAD_SPLIT=2" "SSDHIP_HEAD_SPLIT=3" "SSDHIP_HEAD_SPLIT=3 SSDHIP_HEAD_WGS=224" "SSDHIP_HEAD_SPLIT=3 SSDHIP_HEAD_WGS=200"; do
  r=$(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('decode_ms_in_step'))")
  echo "AB $v -> $r"
done
done
