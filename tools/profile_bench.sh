#!/bin/bash
# usage (GPU box): tools/profile_bench.sh   -> gpurun_out/{bench.json, bench_kernel_table.txt, bench_kernel_stats.csv}
# the bench line of an un-profiled run, then `rocprofv3 --kernel-trace --stats` of the same command (1 timed step)
set -x
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
tail -1 $R/gpurun_out/bench.json | cut -c1-300
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err
f=$(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1)
cp "$f" $R/gpurun_out/bench_kernel_stats.csv
head -12 $R/gpurun_out/bench_kernel_stats.csv | cut -c1-200
# same, one stream only: every kernel runs alone, so the per-kernel averages are comparable with bench.py's event-timed roofline leg
rm -rf /tmp/prof_bench1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench1 -- python3 $R/bench.py --steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg > $R/gpurun_out/bench_prof_c1.json 2> $R/gpurun_out/bench_prof_c1.err
f=$(ls /tmp/prof_bench1/*/*kernel_stats.csv | head -1)
cp "$f" $R/gpurun_out/bench_kernel_stats_c1.csv
head -4 $R/gpurun_out/bench_kernel_stats_c1.csv | cut -c1-200
