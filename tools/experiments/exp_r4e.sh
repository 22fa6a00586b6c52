mkdir -p gpurun_out/r4e
cd /root/repo
python -m pytest tests/test_weight_bank_gpu.py -x -q > gpurun_out/r4e/test_bank.log 2>&1; echo "rc=$?" >> gpurun_out/r4e/test_bank.log
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "PAIR_D=$1 WEIGHT_BANK=$2: $(PWG_PAIR_D=$1 PWG_WEIGHT_BANK=$2 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4e/timing.txt
done
echo "c5 PAIR_D=1 WEIGHT_BANK=1: $(python tools/train_replay.py c5 16 2>&1 | tail -1)" >> gpurun_out/r4e/timing.txt
echo "c2 bank=0: $(PWG_WEIGHT_BANK=0 python tools/train_replay.py c2 16 2>&1 | tail -1)" >> gpurun_out/r4e/timing.txt
echo "c2 bank=1: $(python tools/train_replay.py c2 16 2>&1 | tail -1)" >> gpurun_out/r4e/timing.txt
echo "c4 bank=0: $(PWG_WEIGHT_BANK=0 python tools/train_replay.py c4 16 2>&1 | tail -1)" >> gpurun_out/r4e/timing.txt
echo "c4 bank=1: $(python tools/train_replay.py c4 16 2>&1 | tail -1)" >> gpurun_out/r4e/timing.txt
cat gpurun_out/r4e/timing.txt
python -m pytest tests/test_train_full_shape_gpu.py tests/test_hifigan_train_gpu.py tests/test_pwg_mb_train_gpu.py tests/test_ddp_graph_gpu.py tests/test_more_families_train_gpu.py tests/test_train_cli_gpu.py -x -q > gpurun_out/r4e/test_train.log 2>&1; echo "rc=$?" >> gpurun_out/r4e/test_train.log
tail -5 gpurun_out/r4e/test_bank.log gpurun_out/r4e/test_train.log
