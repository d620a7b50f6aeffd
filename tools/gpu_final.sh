#!/bin/bash
# Round-end visit: the whole GPU suite, the bench line, a kernel-trace profile + one step's timeline, per-kernel MFMA-busy counters.
# Usage (through gpurun): bash tools/gpu_final.sh TAG
set -u
TAG=${1:-r02z}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench_err.log
tail -c 1500 $OUT/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra > $OUT/trace_bench.json 2> $OUT/trace_err.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f $OUT/step_timeline.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    --kernel-trace --output-format csv -d $OUT/pmc_mfma -o fwd -- python $R/bench.py --graph 0 --steps 3 --warmup 8 --no-cpu-baseline --no-extra > $OUT/pmc_mfma.log 2>&1
cd $R
python tools/pmc_fold.py $OUT/pmc_mfma ssdhip > $OUT/pmc_mfma_per_kernel.txt 2>&1
find $OUT -name "*.csv" -size +5M -delete
find $OUT -name "*.db" -delete
head -50 $OUT/pmc_mfma_per_kernel.txt
