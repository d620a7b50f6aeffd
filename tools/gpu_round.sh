#!/bin/bash
# One GPU-box visit: parity tests, the bench line, a kernel-trace profile of the same command, PMC passes.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh TAG [skip_tests]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-}" != "skip_tests" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench_err.log
tail -c 3000 $OUT/bench.json
# kernel trace of the same command (MIOpen's find results are cached by the run above)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/trace_bench.json 2> $GRAFT_REPO_ROOT/$OUT/trace_err.log )
# PMC passes (own runs, --kernel-trace only)
if [ "${3:-}" = "skip_pmc" ]; then exit 0; fi
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && REPS=5 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o dec -- \
      python $GRAFT_REPO_ROOT/tools/pmc_decode.py > $GRAFT_REPO_ROOT/$OUT/pmc_$C.log 2>&1 )
done
python tools/pmc_summary.py $OUT/decode_pmc_traffic.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_summary.log 2>&1
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT $OUT/trace 2>/dev/null | head -40
