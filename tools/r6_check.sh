#!/bin/bash
mkdir -p gpurun_out
for c in 6 10; do
timeout 900 python bench.py --batch 1 --concurrent $c --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-fast-modes --no-ref-layout 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch 1 x $c streams:', d['value'], 'images/s', d['ms_per_step'], 'ms per step'); r=d['roofline']; print(r['kernel'], r['frac'], r['share_of_timed_kernels']); [print(k['kernel'], k.get('frac'), k['share_of_timed_kernels']) for k in r['next_kernels']]" | tee -a gpurun_out/r6_one_image_layout.txt
cp gpurun_out/bench_kernel_table.txt gpurun_out/r6_one_image_kernel_table_c$c.txt
done
