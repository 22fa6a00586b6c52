#!/bin/bash
for dbg in 0 1 2 3 4 5 7; do
  echo "PWG_DBG=$dbg"; PWG_DBG=$dbg python tools/bench_conv.py 16 800 2>&1 | grep -E "res 128 k11 d1|res 128 k3 d1|res 32 k7 d1|res 256 k7 d1"
done
