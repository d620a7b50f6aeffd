#!/bin/bash
# Build the profiling variant of the library (in-kernel phase timers, ablation modes) next to the product one.
#   tools/prof_build.sh                       -> tools/libssdhip_prof.so
#   tools/prof_build.sh NAME -DSSDHIP_X=1 ... -> tools/libssdhip_prof_NAME.so with the extra defines (objects under build/prof_NAME)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=${1:-}
[ $# -gt 0 ] && shift
SUF=${NAME:+_$NAME}
mkdir -p $R/build/prof$SUF
pids=()
for f in $R/ssd_keras_amd/csrc/*.hip; do
  X=""; [ "$(basename $f)" = "ssdhip_decode.hip" ] && X="-mllvm -disable-machine-licm"     # as ssd_keras_amd/build.py EXTRA_CFLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -DSSDHIP_PROFILE $X "$@" \
    -I $R/include -I $R/ssd_keras_amd/csrc -c $f -o $R/build/prof$SUF/$(basename $f .hip).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $R/tools/libssdhip_prof$SUF.so $R/build/prof$SUF/*.o
echo built $R/tools/libssdhip_prof$SUF.so
