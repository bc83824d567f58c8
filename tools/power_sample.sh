#!/bin/bash
# GPU box: samples rocm-smi power / clocks twice a second while a command runs:  tools/power_sample.sh out.txt -- <command...>
out=$1; shift; shift
( while true; do date +%s.%N; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)\|hotspot"; sleep 0.5; done ) > "$out" 2>&1 &
spid=$!
"$@"
rc=$?
kill $spid 2>/dev/null
exit $rc
