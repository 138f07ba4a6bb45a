#!/bin/bash
# A/B of the encoder hipGraph on one box: alternate graphed / eager timed regions (no events in either), 3 pairs.
mkdir -p gpurun_out
for i in 1 2 3; do
  for mode in graph none; do
    python bench.py --no-extras --cpu-images 0 --fast-steps 0 --steps 20 --warmup 3 --profile $mode --profile-steps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$mode', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d.get('graph',{}).get('encoder_body_replays_in_timed_region'))"
  done
done
