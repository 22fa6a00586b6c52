mkdir -p gpurun_out/r4n
cd /root/repo
python -m pytest tests/test_conv_ops_gpu.py tests/test_pwg_melgan_gpu.py tests/test_conv_fuzz_gpu.py -x -q > gpurun_out/r4n/test_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r4n/test_conv.log
tail -n 6 gpurun_out/r4n/test_conv.log
for T in c4 c2; do echo "$T: $(python tools/train_replay.py $T 16 2>&1 | tail -1)" >> gpurun_out/r4n/timing.txt; done
cat gpurun_out/r4n/timing.txt
python -m pytest tests/test_train_full_shape_gpu.py tests/test_pwg_mb_train_gpu.py -x -q -k "c2 or c4 or pwg or mb" 2>&1 | tail -n 3
for T in c4 c2; do PWG_PROF_SHAPES=1 python tools/profile_train_shapes.py $T 400 > gpurun_out/r4n/shapes_$T.txt 2>&1; done
grep -h "small_c\|conv1d_mfma_kernel\|pad1d" gpurun_out/r4n/shapes_c4.txt gpurun_out/r4n/shapes_c2.txt | cut -c1-150
