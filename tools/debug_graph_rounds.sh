#!/bin/bash
# the variants of tools/debug_graph_rounds.py, one process each
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/debug_graph_rounds.py 2>&1 | grep -E "ROUND|Error|error" | tail -12; }
run DBG_LR=1e-8 DBG_TRACE=1
run DBG_LR=1e-8 DBG_TRACE=1 DBG_FUSED=0
run DBG_LR=1e-8 DBG_EAGER_BETWEEN=0
run DBG_LR=1e-8 DBG_B=8
