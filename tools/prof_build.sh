#!/bin/bash
# Build the profiling variant of the library (in-kernel phase timers) next to the product one.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/prof
pids=()
for f in $R/ssd_keras_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -DSSDHIP_PROFILE \
    -I $R/include -I $R/ssd_keras_amd/csrc -c $f -o $R/build/prof/$(basename $f .hip).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $R/tools/libssdhip_prof.so $R/build/prof/*.o
echo built $R/tools/libssdhip_prof.so
