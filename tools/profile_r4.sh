#!/bin/bash
# usage (GPU box): tools/profile_r4.sh  -> gpurun_out/r4_*: the round-4 evidence set (the headline mode is split-bf16 now)
#   r4_bench_prof_c1.json, r4_kernel_stats_1stream.csv   rocprofv3 --kernel-trace --stats of one timed bench step on ONE stream (default mode / batch)
#   pmc_r4_*.txt                                          three separate --pmc passes each (tools/pmc.sh): the split-bf16 3x3 conv at the two row
#                                                         counts the default batch launches it with, the split-bf16 two-pass masked attention, the GEGLU projection
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r4 /tmp/ffn_tune_r4.pt
F="--steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes"
python3 $R/bench.py $F --tune-file /tmp/ffn_tune_r4.pt > /dev/null 2>&1
FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune_r4.pt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4 -- python3 $R/bench.py $F > $R/gpurun_out/r4_bench_prof_c1.json 2> $R/gpurun_out/r4_bench_prof_c1.err
cp "$(ls /tmp/prof_r4/*/*kernel_stats.csv | head -1)" $R/gpurun_out/r4_kernel_stats_1stream.csv
cp $R/gpurun_out/bench_kernel_table.txt $R/gpurun_out/r4_bench_event_table_1stream.txt
# the same with every launch EAGER (--no-graph): graph-replayed launches run back to back and are faster than event-bracketed eager ones (small kernels by
# up to 15 %), so only this run's rocprofv3 averages are like-for-like with the event-timed roofline leg of the same process
rm -rf /tmp/prof_r4e
cd /tmp
FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune_r4.pt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4e -- python3 $R/bench.py $F --no-graph > $R/gpurun_out/r4_bench_prof_c1_eager.json 2> $R/gpurun_out/r4_bench_prof_c1_eager.err
cp "$(ls /tmp/prof_r4e/*/*kernel_stats.csv | head -1)" $R/gpurun_out/r4_kernel_stats_1stream_eager.csv
cp $R/gpurun_out/bench_kernel_table.txt $R/gpurun_out/r4_bench_event_table_1stream_eager.txt
cd $R
cd $R
ONE_MODE=x3 ONE_B=16 bash tools/pmc.sh r4_x3_conv_64x64_320_16rows conv 64 320 320 > /dev/null
ONE_MODE=x3 ONE_B=24 bash tools/pmc.sh r4_x3_conv_64x64_320_24rows conv 64 320 320 > /dev/null
ONE_MODE=x3 ONE_B=24 bash tools/pmc.sh r4_x3_attn_S4096_24rows_2pass_masked attn 4096 320 5 2 > /dev/null
ONE_MODE=x3 bash tools/pmc.sh r4_x3_geglu_98304x2560x320 geglu 98304 320 2560 > /dev/null
ls -la gpurun_out | tail -12
