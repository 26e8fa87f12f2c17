mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r4g_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4g_gputests.log
python tools/bench_x3.py > gpurun_out/r4g_x3_shapes.txt 2>&1
python bench.py --steps 3 --warmup 1 > gpurun_out/r4g_bench_default.json 2> gpurun_out/r4g_bench_default.err
bash tools/profile_r4.sh > gpurun_out/r4_profile.log 2>&1
tail -3 gpurun_out/r4g_gputests.log; grep geglu gpurun_out/r4g_x3_shapes.txt; cut -c1-200 gpurun_out/r4g_bench_default.json
