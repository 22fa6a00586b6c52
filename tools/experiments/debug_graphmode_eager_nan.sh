#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do for c in a b c d; do CASE=$c timeout 300 python tools/experiments/debug_graphmode_eager_nan.py 2>&1 | grep "^case" | tr '\n' ' '; echo; done; done
echo "--- inspect"
CASE=a INSPECT=1 timeout 300 python tools/experiments/debug_graphmode_eager_nan.py 2>&1 | grep -v Warning | tail -40
echo "--- no poison, case a"
CASE=a POISON=0 timeout 300 python tools/experiments/debug_graphmode_eager_nan.py 2>&1 | grep "^case"
