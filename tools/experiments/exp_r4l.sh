mkdir -p gpurun_out/r4l
cd /root/repo
timeout 600 python -m pytest tests/test_losses_gpu.py -x -q > gpurun_out/r4l/test_losses.log 2>&1; echo "rc=$?" >> gpurun_out/r4l/test_losses.log
tail -n 25 gpurun_out/r4l/test_losses.log
for T in c4 c2; do
echo "$T STFT_FFT=0: $(PWG_STFT_FFT=0 python tools/train_replay.py $T 16 2>&1 | tail -1)" >> gpurun_out/r4l/timing.txt
echo "$T STFT_FFT=1: $(python tools/train_replay.py $T 16 2>&1 | tail -1)" >> gpurun_out/r4l/timing.txt
done
cat gpurun_out/r4l/timing.txt
timeout 900 python -m pytest tests/test_train_full_shape_gpu.py tests/test_pwg_mb_train_gpu.py -x -q -k "c2 or c4 or pwg or mb" > gpurun_out/r4l/test_train.log 2>&1; echo "rc=$?" >> gpurun_out/r4l/test_train.log
tail -n 12 gpurun_out/r4l/test_train.log
