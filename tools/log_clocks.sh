#!/bin/bash
# Sample the device's shader clock and package power once per second while a command runs (rocm-smi): what clock does the
# chip sustain under the workload?  (MI355X clocks to its power budget: MI355X_MICROARCH.md, "DVFS give-back".)
# usage: tools/log_clocks.sh OUT.txt -- command ...
OUT=$1; shift; shift
( while true; do
    echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Average Graphics Package Power|Current Socket Graphics Package Power' | sed 's/  */ /g' | tr '\n' '|')"
    sleep 1
  done ) > "$OUT" &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER 2>/dev/null
exit $RC
