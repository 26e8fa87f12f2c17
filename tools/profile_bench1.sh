#!/bin/bash
# usage (GPU box): tools/profile_bench1.sh <tag>  -> gpurun_out/<tag>_kernel_stats_1stream.csv : rocprofv3 --kernel-trace --stats of one
# timed step of bench.py on ONE stream (every kernel runs alone: averages comparable with bench.py's event-timed roofline leg)
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_bench1 /tmp/ffn_tune.pt
# an un-profiled run first writes the igemm tuning table, so that the profiled process launches no tuner candidates
python3 $R/bench.py --steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg --tune-file /tmp/ffn_tune.pt > /dev/null 2>&1
export FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune.pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench1 -- python3 $R/bench.py --steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg > $R/gpurun_out/${tag}_bench_prof_c1.json 2> $R/gpurun_out/${tag}_bench_prof_c1.err
f=$(ls /tmp/prof_bench1/*/*kernel_stats.csv | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats_1stream.csv
head -30 $R/gpurun_out/${tag}_kernel_stats_1stream.csv | cut -c1-220
