#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "pair_output_with_residual or kv64 or pair_producers" 2>&1 | tail -30 | cut -c1-300 | tee gpurun_out/r6_pairres_tests.txt
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-300 | tee gpurun_out/r6_unet_pipeline_tests.txt
for v in 1 0 1; do
  FFN_KV64=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-fast-modes --no-ref-layout 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FFN_KV64=$v', d['value'], d['ms_per_step'], d['roofline']['hbm_bound_kernels']['share_of_timed_kernels'])" | tee -a gpurun_out/r6_pairres_bench.txt
done
cp gpurun_out/bench_kernel_table.txt gpurun_out/r6_bench_kernel_table_pairres.txt
