#!/bin/bash
# Slab-kernel visit: ablations (profiling library) + PMC passes.  Usage (through gpurun): bash tools/gpu_convh.sh TAG
set -u
TAG=${1:-r02n}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 200 python tools/ablate_convh.py $OUT/ablate_convh.json > $OUT/ablate.log 2>&1
tail -5 $OUT/ablate.log
cd /tmp
REPS=3 timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $OUT/pmc_SQ -o c -- python $R/tools/pmc_convh.py > $OUT/pmc_SQ.log 2>&1
REPS=3 timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM \
    --kernel-trace --output-format csv -d $OUT/pmc_SQ2 -o c -- python $R/tools/pmc_convh.py > $OUT/pmc_SQ2.log 2>&1
REPS=3 timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum \
    --kernel-trace --output-format csv -d $OUT/pmc_SQ3 -o c -- python $R/tools/pmc_convh.py > $OUT/pmc_SQ3.log 2>&1
cd $R
for d in pmc_SQ pmc_SQ2 pmc_SQ3; do python tools/pmc_fold.py $OUT/$d conv >> $OUT/pmc_convh_summary.txt 2>&1; done
find $OUT -name "*.csv" -size +5M -delete
find $OUT -name "*.db" -delete
cat $OUT/pmc_convh_summary.txt | head -120
