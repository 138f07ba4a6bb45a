#!/bin/bash
# sample clocks / power while a command runs: tools/smi_watch.sh <outfile> <cmd...>
OUT=$1; shift
( for i in $(seq 1 400); do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)|Performance Level|mclk|fclk" ; sleep 0.2; done ) > $OUT 2>&1 &
W=$!
"$@"
kill $W 2>/dev/null
