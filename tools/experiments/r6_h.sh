#!/bin/bash
# Round 6, call h: HBM-bound helpers after the round-6 kernels (LDS-free small-cout stream, 16-B stretch conv) and the
# single-input-channel variants (non-temporal stores, workgroup count), each against the round-5 form on the same box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06h; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_ops_gpu.py tests/test_pqmf_upsample_gpu.py tests/test_hifigan_gpu.py tests/test_pwg_melgan_gpu.py tests/test_pwg_dropout_gpu.py -q 2>&1 | tail -3
PWG_SMALL_COUT_STREAM=0 PWG_STRETCH_FAST=0 timeout 300 python tools/bench_hbm_helpers.py > $O/helpers_old.txt 2>&1
timeout 300 python tools/bench_hbm_helpers.py > $O/helpers_new.txt 2>&1
PWG_SMALL_CIN_NT=1 timeout 300 python tools/bench_hbm_helpers.py > $O/helpers_nt.txt 2>&1
PWG_SMALL_CIN_WGS=512 timeout 300 python tools/bench_hbm_helpers.py > $O/helpers_wg512.txt 2>&1
PWG_SMALL_CIN_WGS=2048 timeout 300 python tools/bench_hbm_helpers.py > $O/helpers_wg2048.txt 2>&1
PWG_SMALL_CIN_WGS=256 PWG_SMALL_CIN_NT=1 timeout 300 python tools/bench_hbm_helpers.py > $O/helpers_wg256nt.txt 2>&1
for f in old new nt wg512 wg2048 wg256nt; do echo "== $f"; grep -v "^frac\|amdgpu.ids" $O/helpers_$f.txt | cut -c1-150; done
