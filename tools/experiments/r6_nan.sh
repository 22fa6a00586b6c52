#!/bin/bash
# Round 6: bisect the sporadic NaN of EAGER multi-stream branches (PWG_EAGER_BRANCH_STREAMS=1).
# N fresh processes of tools/experiments/debug_graphmode_eager_nan2.py per configuration; a process "fails" when any
# finite-check of its five C3 steps is false.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06nan; mkdir -p $O
cd $R
N=${N:-10}
run_cfg() {  # name, env...
  name=$1; shift
  fails=0; done_=0
  for rep in $(seq 1 $N); do
    env PWG_EAGER_BRANCH_STREAMS=1 "$@" timeout 300 python tools/experiments/debug_graphmode_eager_nan2.py > /tmp/nan_$name.txt 2>&1
    line=$(grep "^RESULT" /tmp/nan_$name.txt | head -1)
    if [ -z "$line" ]; then echo "$name run $rep: no RESULT"; tail -5 /tmp/nan_$name.txt; continue; fi
    done_=$((done_+1))
    if ! echo "$line" | grep -q "^RESULT 0 "; then fails=$((fails+1)); grep -A24 "^RESULT" /tmp/nan_$name.txt | head -26 > $O/fail_${name}_$rep.txt; fi
  done
  echo "$name: $fails failing of $done_ processes" | tee -a $O/summary.txt
}
for c in ${CFGS:-base hwq1 nocache side2}; do
  case $c in
    base) run_cfg base ;;
    control) run_cfg control NAN2_NO_INPUT_RECORD=1 ;;
    wgrad_control) run_cfg wgrad_control NAN2_NO_INPUT_RECORD=1 NAN2_WGRAD=1 NAN2_LAYER0=1 NAN2_NODETAIL=1 ;;
    nodetail) run_cfg nodetail NAN2_NODETAIL=1 ;;
    hwq1) run_cfg hwq1 GPU_MAX_HW_QUEUES=1 ;;
    hwq2) run_cfg hwq2 GPU_MAX_HW_QUEUES=2 ;;
    nocache) run_cfg nocache PYTORCH_NO_CUDA_MEMORY_CACHING=1 ;;
    side2) run_cfg side2 PWG_MAX_SIDE_STREAMS=2 ;;
    layer0) run_cfg layer0 NAN2_LAYER0=1 NAN2_NODETAIL=1 ;;
    wgrad) run_cfg wgrad NAN2_WGRAD=1 NAN2_LAYER0=1 NAN2_NODETAIL=1 ;;
    joinbwd) run_cfg joinbwd NAN2_JOIN_AFTER_BACKWARD=1 NAN2_LAYER0=1 NAN2_NODETAIL=1 ;;
  esac
done
cat $O/summary.txt
