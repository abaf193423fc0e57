#!/bin/bash
# GPU box: throughput against the number of clones in flight.  usage: stream_sweep.sh ["b s" ...]
if [ $# -eq 0 ]; then set -- "1 1" "2 2" "4 4" "8 8" "12 12"; fi
for bs in "$@"; do
  set -- $bs
  timeout -k 10 120 python bench.py --cpu-seconds 0 --kernel-launches 4 --batch $1 --streams $2 --steps 15 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 streams $2: %.0f Mpix/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" || exit 1
done
