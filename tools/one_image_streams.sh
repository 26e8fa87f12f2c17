F="--steps 3 --warmup 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes --no-roofline --batch 1"
for q in 4 8 16; do for c in 6 12; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py $F --concurrent $c 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one image per UNet call, GPU_MAX_HW_QUEUES=$q, $c streams:', d['value'], 'images/s')"
done; done
