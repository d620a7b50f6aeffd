#!/bin/bash
# bench.py's headline with 3 (rounds 2-6) and 40 untimed graph replays in front of the timed region, alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for gw in 3 40 3 40 3 40; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --graph-warmup $gw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('graph-warmup $gw: value %.1f img/s, ms_per_step %.4f, tamed %.4f' % (d['value'], d['ms_per_step'], d['value_tamed_heads']['ms_per_step']))"
done
