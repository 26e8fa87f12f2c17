#!/bin/bash
# images per UNet batch x streams in the headline mode (one process each).  CFGS="24x2 24x3" tools/batch_sweep.sh
F="--steps 2 --warmup 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes --no-roofline"
for cfg in ${CFGS:-8x2 16x2 16x3 24x2 12x2}; do
  b=${cfg%x*}; c=${cfg#*x}
  t0=$(date +%s)
  python3 bench.py $F --batch $b --concurrent $c > /tmp/bs.json 2> /tmp/bs.err
  t1=$(date +%s)
  python3 -c "
import json; d=json.loads(open('/tmp/bs.json').read().strip().splitlines()[-1]); print('batch $b x streams $c:', d['value'], 'images/s', d['ms_per_step'], 'ms per step; process wall', $t1 - $t0, 's')" || tail -3 /tmp/bs.err
done
