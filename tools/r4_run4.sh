mkdir -p gpurun_out
python -m pytest tests/test_warp3d_gpu.py tests/test_pipeline_gpu.py -m gpu -q -s -k "warp or full_size_n50" --deselect "tests/test_pipeline_gpu.py::test_full_size_n50_schedules_vs_oracle_fixture[fs_edit_n20]" > gpurun_out/r4d_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4d_gputests.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r4d_bench_default.json 2> gpurun_out/r4d_bench_default.err
bash tools/profile_r4.sh > gpurun_out/r4_profile.log 2>&1
grep -n "full-size\|passed\|failed" gpurun_out/r4d_gputests.log | tail; cut -c1-300 gpurun_out/r4d_bench_default.json
