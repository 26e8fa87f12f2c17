#!/bin/bash
# usage (GPU box): tools/profile_r5_stats.sh  -> gpurun_out/r5_*: the rocprofv3 --stats half of tools/profile_r5.sh (no PMC passes) (headline mode split-bf16, split-bf16 GEMM core v2)
#   r5_bench_prof_c1.json, r5_kernel_stats_1stream.csv   rocprofv3 --kernel-trace --stats of one timed bench step on ONE stream (default mode / batch)
#   ..._eager                                             the same with every launch eager (like-for-like with the event-timed roofline leg)
#   pmc_r5_*.txt                                          three separate --pmc passes each (tools/pmc.sh): the split-bf16 3x3 conv at the shapes an 8-image
#                                                         batch launches (64x64 level: 16 / 24 rows; 32x32 level: 16 rows, two K slices), the two-pass
#                                                         masked attention, the GEGLU projection
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r5 /tmp/ffn_tune_r5.pt
F="--steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes"
python3 $R/bench.py $F --tune-file /tmp/ffn_tune_r5.pt > /dev/null 2>&1
FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune_r5.pt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5 -- python3 $R/bench.py $F > $R/gpurun_out/r5_bench_prof_c1.json 2> $R/gpurun_out/r5_bench_prof_c1.err
cp "$(ls /tmp/prof_r5/*/*kernel_stats.csv | head -1)" $R/gpurun_out/r5_kernel_stats_1stream.csv
cp $R/gpurun_out/bench_kernel_table.txt $R/gpurun_out/r5_bench_event_table_1stream.txt
rm -rf /tmp/prof_r5e
cd /tmp
FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune_r5.pt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5e -- python3 $R/bench.py $F --no-graph > $R/gpurun_out/r5_bench_prof_c1_eager.json 2> $R/gpurun_out/r5_bench_prof_c1_eager.err
cp "$(ls /tmp/prof_r5e/*/*kernel_stats.csv | head -1)" $R/gpurun_out/r5_kernel_stats_1stream_eager.csv
cp $R/gpurun_out/bench_kernel_table.txt $R/gpurun_out/r5_bench_event_table_1stream_eager.txt
cd $R
ls -la gpurun_out | tail -8
