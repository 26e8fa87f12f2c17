#!/bin/bash
# usage (GPU box): tools/profile_r6_stats.sh  -> gpurun_out/r6_*: rocprofv3 --kernel-trace --stats of one timed bench step on ONE stream at the default layout
# (24 edits per UNet batch), every launch eager (rocprofv3 SIGSEGVs in graph mode from 16 edits per batch on; eager is like for like with the event-timed roofline leg)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r6e /tmp/ffn_tune_r6.pt
F="--steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes"
python3 $R/bench.py $F --tune-file /tmp/ffn_tune_r6.pt > /dev/null 2>&1
FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune_r6.pt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r6e -- python3 $R/bench.py $F --no-graph > $R/gpurun_out/r6_bench_profiled_1stream_eager.json 2> $R/gpurun_out/r6_bench_prof.err
cp "$(ls /tmp/prof_r6e/*/*kernel_stats.csv | head -1)" $R/gpurun_out/r6_bench_kernel_stats_1stream_eager.csv
cp $R/gpurun_out/bench_kernel_table.txt $R/gpurun_out/r6_bench_event_table_1stream_eager.txt
cd $R
