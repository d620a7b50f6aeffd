#!/bin/bash
# Build the profiling variant of the library (in-kernel phase timers) next to the product one.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DSSDHIP_PROFILE \
  -I $R/include -I $R/ssd_keras_amd/csrc -o $R/tools/libssdhip_prof.so $R/ssd_keras_amd/csrc/*.hip
echo built $R/tools/libssdhip_prof.so
