#!/bin/bash
# PMC passes over the convolution variants (own runs, --kernel-trace only).  Usage: bash tools/pmc_conv.sh TAG
set -u
TAG=${1:-conv}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export LAYERS=${LAYERS:-conv4_2,head1} VARIANTS=${VARIANTS:-4,5}
cd /tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o conv -- python $GRAFT_REPO_ROOT/tools/pmc_conv.py > $OUT/pass$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_fold.py $OUT conv > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +5M -delete
cat $OUT/summary.txt | head -120
