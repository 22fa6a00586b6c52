#!/bin/bash
# Round 5: item-major vs x-window-major logical tile order of the convolution kernel (csrc/conv1d.hip: tile_of_workgroup).
# Per-shape A/B with bit-identity check, then the captured C3 / C5 training steps under both orders.
# Result: profiles/r05_tile_order_ab.txt
mkdir -p gpurun_out/r5to
O=gpurun_out/r5to
PWG_TILE_ORDER=0 timeout 200 python tools/bench_tile_order.py > $O/order0.txt 2> $O/order0.err
timeout 200 python tools/bench_tile_order.py > $O/auto.txt 2> $O/auto.err
python tools/bench_tile_order.py --compare $O/order0.txt $O/auto.txt > $O/compare.txt 2>&1
{
  for cfg in c3 c5; do
    for rep in 1 2; do [ $cfg = c5 ] && [ $rep = 2 ] && continue
      echo "== $cfg PWG_TILE_ORDER=0 (rep $rep)"; PWG_TILE_ORDER=0 timeout 200 python tools/train_replay.py $cfg 26 2>&1 | tail -1
      echo "== $cfg planner's order (rep $rep)"; timeout 200 python tools/train_replay.py $cfg 26 2>&1 | tail -1
    done
  done
} > $O/steps.txt 2>&1
cat $O/compare.txt $O/steps.txt
