#!/bin/bash
# scratch A/B on one box: tools/ab_check.sh ENVVAR  (short bench lines with ENVVAR=1 / 0 alternating)
mkdir -p gpurun_out
V=$1
for v in 1 0 1 0; do
  env $V=$v timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-fast-modes --no-ref-layout 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$v', d['value'], d['ms_per_step'], 'hbm share', d['roofline']['hbm_bound_kernels']['share_of_timed_kernels'])" | tee -a gpurun_out/ab_$V.txt
done
