#!/bin/bash
# Round 6, call j: mode-4 weight gradient with 16-B X staging and 64-column chunks for 1 x 1 layers: tests, then per-shape
# A/B (same box) against dword staging / 32-column chunks.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv_ops_gpu.py tests/test_conv_fuzz_gpu.py tests/test_fold_batch_gpu.py tests/test_discriminator_gpu.py tests/test_train_full_shape_gpu.py tests/test_resstack_gpu.py tests/test_pwg_melgan_gpu.py tests/test_pqmf_upsample_gpu.py -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
PWG_WG_ROWS_X4=0 PWG_WG_K1_TT64=0 timeout 600 python tools/profile_train_shapes.py c3 400 > $O/shapes_c3_a.txt 2>&1
timeout 600 python tools/profile_train_shapes.py c3 400 > $O/shapes_c3_b.txt 2>&1
PWG_WG_ROWS_X4=0 PWG_WG_K1_TT64=0 timeout 600 python tools/profile_train_shapes.py c4 400 > $O/shapes_c4_a.txt 2>&1
timeout 600 python tools/profile_train_shapes.py c4 400 > $O/shapes_c4_b.txt 2>&1
python tools/time_by_category.py $O/shapes_c3_a.txt $O/shapes_c3_b.txt
grep "wgrad.*k1 " $O/shapes_c4_a.txt | head -5; grep "wgrad.*k1 " $O/shapes_c4_b.txt | head -5
for rep in 1 2 3; do for T in c3 c5 c4; do
PWG_WG_ROWS_X4=0 PWG_WG_K1_TT64=0 timeout 300 python tools/train_replay.py $T 60 2>/dev/null | tail -1 | sed "s/^/a /" >> $O/replay.txt
timeout 300 python tools/train_replay.py $T 60 2>/dev/null | tail -1 | sed "s/^/b /" >> $O/replay.txt
done; done; cat $O/replay.txt
