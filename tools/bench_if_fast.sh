#!/bin/bash
# one quick headline run; the full bench line + timeline only if the box is not one of the pool's slow ones (ms_per_step below $1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; TAG=${2:-r04zz5}; LIM=${1:-2.25}
bash tools/gpu.sh $TAG benchq
ms=$(python -c "import json;print(json.load(open('gpurun_out/$TAG/benchq.json'))['ms_per_step'])")
echo "quick run: $ms ms per step (limit $LIM)"
if python -c "import sys;sys.exit(0 if float('$ms')<float('$LIM') else 1)"; then bash tools/gpu.sh $TAG bench timeline; else echo "slow box: skipped the full line"; fi
