#!/bin/bash
# round-6 evidence on one box: GPU test suite, default bench line, GEMM shape table, rocprofv3 stats, PMC of the dominant conv, the driver's flags, other entry points
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | cut -c1-300 > gpurun_out/r6_gputests_full.txt
python3 -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > gpurun_out/r6_smoke.txt 2>&1
python3 bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
cp gpurun_out/bench_kernel_table.txt gpurun_out/r6_bench_kernel_table.txt
timeout 900 python3 tools/bench_x3.py > gpurun_out/r6_x3_gemm_shapes.txt 2>&1
bash tools/profile_r6_stats.sh > gpurun_out/r6_prof_stats.log 2>&1
ONE_MODE=x3 ONE_B=72 bash tools/pmc.sh r6_x3_conv_64x64_320_72rows conv 64 320 320 > /dev/null
ONE_MODE=x3 ONE_B=48 bash tools/pmc.sh r6_x3_conv_64x64_320_48rows conv 64 320 320 > /dev/null
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_driver_flags.json 2> gpurun_out/r6_bench_driver_flags.err
bash tools/other_entry_points.sh > gpurun_out/r6_other_entry_points.txt 2>&1
ls -la gpurun_out | tail -20
