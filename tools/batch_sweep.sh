#!/bin/bash
# images per UNet batch x streams in the headline mode (one process each)
F="--steps 2 --warmup 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes --no-roofline"
for cfg in "8 2" "16 2" "16 3" "24 2" "12 2"; do
  set -- $cfg
  t0=$(date +%s)
  python3 bench.py $F --batch $1 --concurrent $2 > /tmp/bs.json 2> /tmp/bs.err
  t1=$(date +%s)
  python3 -c "
import json; d=json.loads(open('/tmp/bs.json').read().strip().splitlines()[-1]); print('batch $1 x streams $2:', d['value'], 'images/s', d['ms_per_step'], 'ms per step; process wall', $t1 - $t0, 's; peak torch memory GB', d['config'].get('peak_memory_gb'))" || tail -3 /tmp/bs.err
done
