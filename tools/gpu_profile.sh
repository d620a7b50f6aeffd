#!/bin/bash
# Profiling visit: rocprofv3 kernel-trace stats of the bench command, PMC passes of the decode path, in-kernel phase timers.
# Usage (through gpurun): bash tools/gpu_profile.sh TAG
set -u
TAG=${1:-r02p}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/trace_bench.json 2> $OUT/trace_err.log
for C in FETCH_SIZE WRITE_SIZE; do
  REPS=5 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o dec -- python $R/tools/pmc_decode.py > $OUT/pmc_$C.log 2>&1
done
REPS=5 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc_SQ -o dec -- python $R/tools/pmc_decode.py > $OUT/pmc_SQ.log 2>&1
REPS=5 timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM \
    --kernel-trace --output-format csv -d $OUT/pmc_SQ2 -o dec -- python $R/tools/pmc_decode.py > $OUT/pmc_SQ2.log 2>&1
cd $R
python tools/pmc_summary.py $OUT/decode_pmc_traffic.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_summary.log 2>&1
python tools/pmc_fold.py $OUT/pmc_SQ ssdhip > $OUT/pmc_SQ_summary.txt 2>&1
python tools/pmc_fold.py $OUT/pmc_SQ2 ssdhip >> $OUT/pmc_SQ_summary.txt 2>&1
timeout 600 python tools/phase_profile.py > $OUT/phase_profile.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
find $OUT -name "*.db" -delete
cat $OUT/pmc_SQ_summary.txt | head -80
cat $OUT/phase_profile.txt | tail -20
ls $OUT/trace | head
