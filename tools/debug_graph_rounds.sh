#!/bin/bash
# variants of the graph-replay reproducers, one process each
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/debug_loss_graph.py 2>&1 | grep -E "LOSSGRAPH|Error|error" | tail -3; }
run DBG_B=32
run DBG_B=2
run DBG_B=2 DBG_TOUCH=1
run2() { env "$@" timeout 300 python tools/debug_graph_rounds.py 2>&1 | grep -E "ROUNDS|Error|error" | tail -3; }
run2 DBG_LR=1e-8
run2 DBG_LR=1e-8 DBG_EAGER_BETWEEN=0
