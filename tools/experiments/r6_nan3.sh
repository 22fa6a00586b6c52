#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06nan3; mkdir -p $O
cd $R
for rep in $(seq 1 ${N:-40}); do
  PWG_EAGER_BRANCH_STREAMS=1 timeout 300 python tools/experiments/debug_eager_nan3.py > /tmp/nan3.txt 2>&1
  line=$(grep "^RESULT" /tmp/nan3.txt | head -1)
  echo "run $rep: $line"
  if [ -z "$line" ]; then tail -5 /tmp/nan3.txt; fi
  if [ -n "$line" ] && ! echo "$line" | grep -q "^RESULT 0 "; then cp /tmp/nan3.txt $O/fail_$rep.txt; grep -c "" $O/fail_$rep.txt; fails=$((fails+1)); if [ "${fails:-0}" -ge 2 ]; then break; fi; fi
done
ls $O
