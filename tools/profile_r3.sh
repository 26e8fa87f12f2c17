#!/bin/bash
# usage (GPU box): tools/profile_r3.sh  -> gpurun_out/r3_*: the round-3 evidence set
#   r3_kernel_stats_1stream.csv          rocprofv3 --kernel-trace --stats of one timed bench step on one stream (bf16 fast mode)
#   r3_x3_kernel_stats_1stream.csv       the same for the split-bf16 mode (--dtype bf16x3 --batch 8)
#   pmc_r3_*.txt                         three separate --pmc passes each: dominant conv, GEGLU projection, masked / unmasked attention
#                                        (attn_pp_kernel<true> / <false>), and the split-bf16 conv / attention
cd $GRAFT_REPO_ROOT
bash tools/profile_bench1.sh r3 > gpurun_out/r3_profile_bench1.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_x3 /tmp/ffn_tune_x3.pt
python3 $R/bench.py --dtype bf16x3 --batch 8 --steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg --tune-file /tmp/ffn_tune_x3.pt > /dev/null 2>&1
FFN_IGEMM_TUNE_FILE=/tmp/ffn_tune_x3.pt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -- python3 $R/bench.py --dtype bf16x3 --batch 8 --steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fp8-leg > $R/gpurun_out/r3_x3_bench_prof_c1.json 2> $R/gpurun_out/r3_x3_bench_prof_c1.err
cp "$(ls /tmp/prof_x3/*/*kernel_stats.csv | head -1)" $R/gpurun_out/r3_x3_kernel_stats_1stream.csv
cd $R
ONE_B=48 bash tools/pmc.sh r3_conv_64x64_320_48rows conv 64 320 320 > /dev/null
ONE_B=32 bash tools/pmc.sh r3_conv_64x64_320_32rows conv 64 320 320 > /dev/null
bash tools/pmc.sh r3_geglu_196608x2560x320 geglu 196608 320 2560 > /dev/null
ONE_B=16 bash tools/pmc.sh r3_attn_S4096_16rows_2pass_masked attn 4096 320 5 2 > /dev/null
ONE_B=16 bash tools/pmc.sh r3_attn_S4096_16rows_1pass attn 4096 320 5 1 > /dev/null
ONE_MODE=x3 ONE_B=24 bash tools/pmc.sh r3_x3_conv_64x64_320_24rows conv 64 320 320 > /dev/null
ONE_MODE=x3 ONE_B=8 bash tools/pmc.sh r3_x3_attn_S4096_8rows_2pass_masked attn 4096 320 5 2 > /dev/null
ls -la gpurun_out | tail -20
