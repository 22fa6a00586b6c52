#!/bin/bash
# Round 5: wave priority by phase in conv1d_mfma_dma_kernel (compile-time PWG_PRIO variants built by tools/build_variant.py:
# bit 1 = DMA issue of the next chunk raised, bit 2 = epilogue raised; bit 0 = matrix phase raised was measured first,
# -2 %).  Same box, alternating libraries.  Result: profiles/r05_setprio_ab.txt
mkdir -p gpurun_out/r5sp
O=gpurun_out/r5sp
P=$PWD/parallelwavegan_amd
for rep in 1 2; do
  for v in base prio2 prio4 prio6; do
    lib=$P/libpwgkernels_$v.so; [ $v = base ] && lib=$P/libpwgkernels.so
    echo "=== $v (rep $rep)"; PWG_KERNEL_LIB=$lib timeout 200 python tools/bench_conv.py 16 800 2>&1 | grep -v amdgpu.ids | tail -32
  done
done > $O/conv.txt 2>&1
grep -E "===|TOTAL" $O/conv.txt
for rep in 1 2; do
  for v in base prio2 prio4 prio6; do
    lib=$P/libpwgkernels_$v.so; [ $v = base ] && lib=$P/libpwgkernels.so
    echo "== c3 $v (rep $rep)"; PWG_KERNEL_LIB=$lib timeout 200 python tools/train_replay.py c3 26 2>&1 | tail -1
  done
done | tee $O/c3.txt
