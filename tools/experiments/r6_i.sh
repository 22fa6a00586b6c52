#!/bin/bash
# Round 6, call i: (1) tests of the new helper kernels; stream / 16-B forms against the round-5 kernels bit for bit;
# (2) the NaN bisect of eager multi-stream branches.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06i; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_ops_gpu.py tests/test_pqmf_upsample_gpu.py tests/test_hifigan_gpu.py tests/test_pwg_melgan_gpu.py tests/test_pwg_dropout_gpu.py -q 2>&1 | tail -3
PWG_SMALL_COUT_STREAM=0 PWG_STRETCH_FAST=0 python tools/experiments/bitcmp_small_cout.py 2>/dev/null > $O/bits_old.txt
python tools/experiments/bitcmp_small_cout.py 2>/dev/null > $O/bits_new.txt
diff $O/bits_old.txt $O/bits_new.txt && echo "BIT-IDENTICAL: stream / 16-B kernels == round-5 kernels" | tee $O/bits_verdict.txt; cat $O/bits_new.txt
N=10 bash tools/experiments/r6_nan.sh
