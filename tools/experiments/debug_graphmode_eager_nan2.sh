#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2 3 4 5 6 7 8; do
  timeout 300 python tools/experiments/debug_graphmode_eager_nan2.py > /tmp/nan2_full.txt 2>&1
  grep -A40 "^RESULT" /tmp/nan2_full.txt > /tmp/nan2.txt
  if [ ! -s /tmp/nan2.txt ]; then echo "run $rep: no RESULT line"; tail -25 /tmp/nan2_full.txt; break; fi
  head -1 /tmp/nan2.txt
  if ! grep -q "^RESULT 0 " /tmp/nan2.txt; then cat /tmp/nan2.txt; break; fi
done
