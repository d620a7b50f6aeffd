#!/bin/bash
# bench.py's headline step (graph) under the predictor-head schedules, twice each: ms_per_step
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in "SSDHIP_GRAPH_HEAD_OVERLAP=0" "SSDHIP_GRAPH_HEAD_OVERLAP=3" "SSDHIP_GRAPH_HEAD_OVERLAP=3 SSDHIP_HEAD_WGS=160" "SSDHIP_GRAPH_HEAD_OVERLAP=3 SSDHIP_HEAD_WGS=192" "SSDHIP_GRAPH_HEAD_OVERLAP=3 SSDHIP_HEAD_WGS=224" "SSDHIP_GRAPH_HEAD_OVERLAP=4" "SSDHIP_GRAPH_HEAD_OVERLAP=4 SSDHIP_HEAD_WGS=240" "SSDHIP_GRAPH_HEAD_OVERLAP=4 SSDHIP_HEAD_WGS=192" ${AB_EXTRA:+"$AB_EXTRA"}; do
  r=$(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('decode_ms_in_step'))")
  echo "AB $v -> $r"
done
done
