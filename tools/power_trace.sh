#!/bin/bash
# usage: tools/power_trace.sh OUT.csv CMD...   -- samples socket power and sclk (rocm-smi) while CMD runs
out=$1; shift
( while true; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --csv 2>/dev/null | grep -v "^$" | tail -n 1
    sleep 0.05
  done ) > "$out" &
SPID=$!
"$@"
rc=$?
kill $SPID 2>/dev/null
wait $SPID 2>/dev/null
exit $rc
