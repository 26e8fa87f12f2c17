set -x
ONE_MODE=x3 ONE_B=72 bash tools/pmc.sh r5_x3_conv_64x64_320_72rows conv 64 320 320 > /dev/null
ONE_MODE=x3 ONE_B=48 bash tools/pmc.sh r5_x3_conv_64x64_320_48rows conv 64 320 320 > /dev/null
bash tools/profile_r5_stats.sh > gpurun_out/prof_stats.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_flags.json 2> gpurun_out/r5_bench_driver_flags.err
FFN_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 2 --steps 1 --warmup 1 --batch 8 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes > gpurun_out/r5_bench_gpus2_shared_device.json 2> gpurun_out/r5_bench_gpus2.err
ls -la gpurun_out | tail -20
