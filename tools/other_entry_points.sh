#!/bin/bash
# the other schedules and entry points in the headline mode (profiles/r<round>_other_entry_points.txt): tools/bench_other.py + bench.py --start-step 35 / 15
python3 tools/bench_other.py 2>/dev/null
F="--steps 2 --warmup 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes --no-roofline"
for ss in 35 15; do
  python3 bench.py $F --start-step $ss 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('FreeFine_generation, GeoBench-%s schedule (N=50, start_step=$ss), split-bf16, %d edits per batch x %d streams (python bench.py --start-step $ss): %.4f images/s (%.2f ms per %d-image step)' % ('2D' if $ss == 35 else '3D', d['config']['images_per_unet_batch'], d['config']['concurrent_streams'], d['value'], d['ms_per_step'], d['config']['images_per_gpu_per_step']))"
done
