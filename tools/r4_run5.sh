mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py -m gpu -q -s -k "metric_schedules or fs_edit_n20" > gpurun_out/r4e_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4e_gputests.log
ONE_MODE=x3 ONE_B=24 bash tools/pmc.sh r4_x3_conv_32x32_640_24rows conv 32 640 640 > /dev/null
ONE_MODE=x3 ONE_B=16 bash tools/pmc.sh r4_x3_conv_32x32_640_16rows conv 32 640 640 > /dev/null
ONE_MODE=x3 ONE_B=24 bash tools/pmc.sh r4_x3_conv_16x16_1280_24rows conv 16 1280 1280 > /dev/null
SECONDS=0
python bench.py --steps 20 --warmup 5 > gpurun_out/r4e_bench_driver_flags.json 2> gpurun_out/r4e_bench_driver_flags.err
echo "driver-flag bench wall clock: $SECONDS s" > gpurun_out/r4e_bench_driver_flags.time
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4e_smoke.log 2>&1
tail -4 gpurun_out/r4e_gputests.log; cat gpurun_out/r4e_bench_driver_flags.time; cut -c1-200 gpurun_out/r4e_bench_driver_flags.json; tail -2 gpurun_out/r4e_smoke.log
