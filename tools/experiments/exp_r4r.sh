mkdir -p gpurun_out/r4r
cd /root/repo
python -m pytest tests/test_ddp_graph_gpu.py tests/test_train_cli_gpu.py tests/test_wavenet_layer_gpu.py tests/test_pwg_melgan_gpu.py -x -q -s > gpurun_out/r4r/test.log 2>&1; echo "rc=$?" >> gpurun_out/r4r/test.log
grep -E "passed|failed|rc=|\[ddp\]|Error" gpurun_out/r4r/test.log | tail -8
for D in 0 1; do
PWG_DDP_DIRECT=$D PWG_FORCE_DIST=1 python bench.py --no-extra-configs --no-cpu-baseline --no-latency --steps 3 --warmup 1 --train-steps 40 --train-warmup 8 2> gpurun_out/r4r/bench_dist_direct$D.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('RCCL world-of-one DDP_DIRECT=$D:', {k:(round(v.get('ms_per_step'),2), round(v.get('value'),2)) for k,v in [('c3',d['train']),('c5',d['configs']['c5_train'])]})" >> gpurun_out/r4r/timing.txt
done
python bench.py --no-extra-configs --no-cpu-baseline --no-latency --steps 3 --warmup 1 --train-steps 40 --train-warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('no DDP:', {k:(round(v.get('ms_per_step'),2), round(v.get('value'),2)) for k,v in [('c3',d['train']),('c5',d['configs']['c5_train'])]})" >> gpurun_out/r4r/timing.txt
cat gpurun_out/r4r/timing.txt
