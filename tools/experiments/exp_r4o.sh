mkdir -p gpurun_out/r4o
cd /root/repo
python -m pytest tests/test_weight_bank_gpu.py tests/test_ddp_graph_gpu.py tests/test_hifigan_train_gpu.py -x -q -s > gpurun_out/r4o/test.log 2>&1; echo "rc=$?" >> gpurun_out/r4o/test.log
grep -E "passed|failed|rc=|\[ddp\]|Error" gpurun_out/r4o/test.log | tail -8
echo "c3 bank tiled=0: $(PWG_BANK_TILED=0 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4o/timing.txt
echo "c3 bank tiled=1: $(python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4o/timing.txt
echo "c5 bank tiled=1: $(python tools/train_replay.py c5 16 2>&1 | tail -1)" >> gpurun_out/r4o/timing.txt
for D in 0 1; do
PWG_DDP_DIRECT=$D PWG_FORCE_DIST=1 python bench.py --no-extra-configs --no-cpu-baseline --no-latency --steps 3 --warmup 1 --train-steps 30 --train-warmup 6 > gpurun_out/r4o/bench_dist_direct$D.json 2> gpurun_out/r4o/bench_dist_direct$D.err
python - <<PY >> gpurun_out/r4o/timing.txt
import json
d=json.loads([l for l in open("gpurun_out/r4o/bench_dist_direct$D.json") if l.startswith("{")][-1])
print("RCCL world-of-one DDP_DIRECT=$D:", {k:(v.get("ms_per_step"), v.get("value")) for k,v in [("c3",d["train"]),("c5",d["configs"]["c5_train"])]})
PY
done
python bench.py --no-extra-configs --no-cpu-baseline --no-latency --steps 3 --warmup 1 --train-steps 30 --train-warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('no DDP:', d['summary'])" >> gpurun_out/r4o/timing.txt
cat gpurun_out/r4o/timing.txt
