#!/bin/bash
# usage (GPU box): tools/profile_r3_stats.sh -> the two rocprofv3 --kernel-trace --stats summaries of tools/profile_r3.sh only (no PMC passes)
cd $GRAFT_REPO_ROOT
bash tools/profile_bench1.sh r3 > gpurun_out/r3_profile_bench1.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_x3 /tmp/ffn_tune_x3.pt
python3 $R/bench.py --dtype bf16x3 --batch 8 --steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg --tune-file /tmp/ffn_tune_x3.pt > /dev/null 2>&1
FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune_x3.pt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -- python3 $R/bench.py --dtype bf16x3 --batch 8 --steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg > $R/gpurun_out/r3_x3_bench_prof_c1.json 2> $R/gpurun_out/r3_x3_bench_prof_c1.err
cp "$(ls /tmp/prof_x3/*/*kernel_stats.csv | head -1)" $R/gpurun_out/r3_x3_kernel_stats_1stream.csv
