mkdir -p gpurun_out/r4j
cd /root/repo
export PWG_PAIR_D=0
python -m pytest tests/test_conv_ops_gpu.py tests/test_conv_fuzz_gpu.py tests/test_gconv_gpu.py tests/test_discriminator_gpu.py tests/test_pwg_melgan_gpu.py -x -q > gpurun_out/r4j/test_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r4j/test_conv.log
tail -n 15 gpurun_out/r4j/test_conv.log
for T in c4 c2 c3; do
echo "$T ROWS32=0 SMALL_CIN=0: $(PWG_ROWS32=0 PWG_SMALL_CIN=0 python tools/train_replay.py $T 16 2>&1 | tail -1)" >> gpurun_out/r4j/timing.txt
echo "$T ROWS32=1 SMALL_CIN=0: $(PWG_ROWS32=1 PWG_SMALL_CIN=0 python tools/train_replay.py $T 16 2>&1 | tail -1)" >> gpurun_out/r4j/timing.txt
echo "$T ROWS32=1 SMALL_CIN=1: $(python tools/train_replay.py $T 16 2>&1 | tail -1)" >> gpurun_out/r4j/timing.txt
done
cat gpurun_out/r4j/timing.txt
python -m pytest tests/test_train_full_shape_gpu.py tests/test_pwg_mb_train_gpu.py -x -q > gpurun_out/r4j/test_train.log 2>&1; echo "rc=$?" >> gpurun_out/r4j/test_train.log
tail -n 5 gpurun_out/r4j/test_train.log
