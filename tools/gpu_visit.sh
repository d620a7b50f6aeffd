#!/bin/bash
# One GPU-box visit (round 2+): python tools/gpu_visit.sh TAG [tests|notests] [bench|nobench] [extra command ...]
# Everything lands under gpurun_out/TAG.
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-tests}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.txt 2>&1
  tail -25 $OUT/pytest_gpu.txt
fi
if [ "${3:-bench}" = "bench" ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench_err.log
  tail -c 6000 $OUT/bench.json
  tail -5 $OUT/bench_err.log
fi
shift 3 2>/dev/null || true
if [ $# -gt 0 ]; then
  bash -c "$*" > $OUT/extra.log 2>&1
  tail -40 $OUT/extra.log
fi
